"""The lookup tables behind the fp64 kernels' log / sincos / exp (montecarlocuda_amd/csrc/mc_tables_f64.inc):
the committed file is what the generator produces, and the host twin of the device arithmetic
(tools/check_f64_tables.c, same operations with C99 fma) stays within its error bounds against 80-bit libm,
including the edge inputs (u -> 1, u -> 0, the sqrt(1/2) seam of the mantissa reduction)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_tables_are_the_generated_ones(tmp_path):
    out = tmp_path / "tables.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_f64_tables.py"), str(out)], check=True, capture_output=True)
    assert out.read_text() == open(os.path.join(ROOT, "montecarlocuda_amd", "csrc", "mc_tables_f64.inc")).read()


def test_host_twin_of_the_table_math_meets_its_bounds(tmp_path):
    exe = tmp_path / "check_f64_tables"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tools", "check_f64_tables.c"), "-lm"],
                   check=True, capture_output=True)
    r = subprocess.run([str(exe), "3000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "normal" in r.stdout
