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


# ---- the ONE stdout line: at most 4096 bytes, whatever the run measured (round 4's had grown to 32 KB: BENCH_r04 parsed null) ----
import copy
import json

import pytest

sys.path.insert(0, ROOT)
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r04.json")))


def _check_line(line):
    text = json.dumps(line)
    assert len(text) <= 4096, len(text)
    assert "\n" not in text and json.loads(text) == line
    for k in CONTRACT:
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert line["detail"]
    return text


def test_compact_line_of_the_recorded_round_4_run_fits_4096_bytes():
    import bench
    d = _fixture()
    assert len(json.dumps(d)) > 30000            # the record that broke the driver's parser
    line = bench.compact_line(d, "bench_detail.json")
    _check_line(line)
    assert "dropped_for_size" not in line
    # the headline is carried in full precision, everything else at 6 significant digits
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"] and line["timed_region_s"] == d["timed_region_s"]
    assert line["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-5)
    assert line["roofline"]["issue_frac"] == pytest.approx(d["roofline"]["issue_frac"], rel=1e-5)
    assert line["roofline"]["issue_model"]["frac_effective"] <= 1.0 and line["roofline"]["traffic_stale"] is False
    assert line["fp64"]["value"] == pytest.approx(d["fp64"]["value"], rel=1e-5) and len(line["region_ms_per_step"]) == 5
    # strong_summary: per config [torch path, C library]: T(1), T(shard 0 of 8), device-side efficiency hot and cold
    ss = line["strong_summary"]
    rows = {r["config"]: r for r in d["strong"]["rows"]}
    assert set(ss) == set(rows)
    sh = {(r["config"], r["shard_of"]): r for r in d["strong"]["shard_rows"]}
    csh = {(r["config"], r["shard_of"]): r for r in d["c_multi"]["rows"] if "shard_of" in r}
    for c in ("C4", "C5", "C4_n32", "C5_n32"):
        assert ss[c]["t1_ms"][0] == pytest.approx(rows[c]["wall_ms_median"], rel=1e-5)
        assert ss[c]["t8_ms"][0] == pytest.approx(sh[(c, 8)]["wall_ms_median"], rel=1e-5)
        assert ss[c]["eff8"] == [pytest.approx(sh[(c, 8)]["device_side_efficiency"], rel=1e-5),
                                 pytest.approx(csh[(c, 8)]["device_side_efficiency"], rel=1e-5)]
        assert ss[c]["eff8_cold"] == [pytest.approx(sh[(c, 8)]["cold"]["device_side_efficiency"], rel=1e-5),
                                      pytest.approx(csh[(c, 8)]["cold"]["device_side_efficiency"], rel=1e-5)]
    assert "eff8_cold" not in ss["C4x10"]          # the 10x sizes are measured hot only


def test_compact_line_of_an_eight_rank_run_carries_the_roster_and_fits():
    import bench
    d = _fixture()
    d["n_gpus"] = 8
    for k in ("cpu_baseline", "cpu_all_cores", "c_multi"):
        d.pop(k)
    d["strong"]["shard_rows"] = []
    for r in d["strong"]["rows"]:
        r["paths_per_gpu"] = r["paths_total"] // 8
    d.update(world_size=8, backend="nccl", rccl_version="2.26.6",
             ranks=[{"rank": i, "device": i, "pci": "0000:%02x:00.0" % (5 + 16 * i), "host": "mi355x-node-0123456789abcdef"} for i in range(8)])
    line = bench.compact_line(d, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= 4096
    assert len(line["ranks"]) == 8 and line["world_size"] == 8 and line["backend"] == "nccl" and line["rccl_version"] == "2.26.6"
    assert {r["pci"] for r in line["ranks"]} == {r["pci"] for r in d["ranks"]}
    assert line["strong_summary"]["C4"]["paths_per_gpu"] == 125000000 and "wall_ms_median" in line["strong_summary"]["C5"]
    assert "cpu_baseline" not in line                      # rank 0 at N = 1 only


def test_compact_line_sheds_parts_rather_than_exceed_the_limit():
    import bench
    d = _fixture()
    # the C library on an 8-GPU node: rows for G = 2, 4, 8 of every config (what drivers/multiBench prints there)
    extra = []
    for r in [x for x in d["c_multi"]["rows"] if x.get("devices") == 1 and "shard_of" not in x and "config" in x]:
        for G in (2, 4, 8):
            extra.append(dict(copy.deepcopy(r), devices=G, strong_efficiency_vs_1=0.9123456, fanout_us=9.87654, wall_ms_median=r["wall_ms_median"] / G))
    d["c_multi"]["rows"] += extra
    d["devices_visible"] = 8
    d["device"] = {"index": 0, "pci": "0000:05:00.0", "name": "AMD Instinct MI355X", "compute_units": 256}
    line = bench.compact_line(d, "some/very/long/path/" * 5 + "bench_detail.json")
    _check_line(line)
    if "dropped_for_size" in line:
        assert set(line["dropped_for_size"]) <= {"cpu_all_cores", "region_ms_per_step", "strong_summary.c_devices", "strong_summary", "ranks", "fp64"}
    else:
        assert line["strong_summary"]["C4"]["c_devices"]["8"][1] == pytest.approx(0.912346)
    # and a record that cannot be made to fit is an error, not a long line
    d["config"]["workload"] = "x" * 5000
    with pytest.raises(RuntimeError):
        bench.compact_line(d, "bench_detail.json")
