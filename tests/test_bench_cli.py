"""bench.py's command line, without a GPU: the help text must render (argparse %-formats every help string, so a bare
per-cent sign in one of them breaks `--help` for all) and must name the flags of the driver's contract."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_renders_and_names_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--regions", "--strong-reps"):
        assert flag in r.stdout
