"""bench.py itself, on the GPU box: the JSON contract at N=1, and the multi-rank bookkeeping rehearsed
with two ranks sharing the one GPU (gloo backend: triples reduced on the host; on an 8-GPU node the
same code runs with RCCL).  Short runs -- this checks plumbing, not speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 10.386270784322328


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_contract_single_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "4", "--cpu-seconds", "0.5",
                          "--fp64-steps", "5"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _json_line(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 4 and d["unit"] == "paths/s" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and "workload" in d["config"]
    assert d["regions"] == 5 and d["paths_priced"] == 5 * 40 * 10 ** 8 and d["value"] > 1e11
    # the reported step time is the median of the five regions
    assert len(d["region_ms_per_step"]) == 5 and d["ms_per_step"] == sorted(d["region_ms_per_step"])[2]
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and 0 < r["frac"] < 1
    assert r["achieved"] == pytest.approx(15.5 * 1e8 / (r["avg_kernel_us"] * 1e-6) / 1e12, rel=1e-9)
    # the per-launch duration is the kernel alone on the device (50 launches one at a time), never above the step
    # period by more than the launch gap; the in-region samples (every launch of a short run) are kept for the record
    assert r["kernel_samples"] == 50 and r["duration_basis"].startswith("exclusive")
    assert 40 < r["avg_kernel_us"] < 70 and r["in_region"]["kernel_samples"] >= 8
    assert r["in_region"]["avg_kernel_us"] >= r["avg_kernel_us"] * 0.95
    assert 0 < r["effective"]["frac"] < 1 and d["timed_region_s"] == pytest.approx(d["ms_per_step"] * 40e-3, rel=1e-9)
    assert "not measured in this run" in r["traffic_source"]
    if r["traffic_stale"]:     # device code, launch-shape rules or grid changed since the committed PMC passes: the model is withheld
        assert "issue_frac" not in r and "traffic_stale_note" in r
    elif "issue_frac" in r:    # a CEILING now: the cheapest measured cost of every opcode of the hot loop -- never above 1, on the step period either
        assert 0.5 < r["issue_frac"] <= 1.0 and r["issue_model"]["frac_effective"] <= 1.0
        assert r["issue_model"]["ceiling_us"] <= r["issue_model"]["typical_us"]
    else:
        assert "issue_model_withheld" in r
    assert r["grid_workgroups"] == 2048 and len(r["launch_stamp"]) == 16
    # counters collected from THIS build at THIS grid must not be called stale (round 4: the grid used to be read after the
    # fp64 side run and the strong rows, i.e. from another kernel's launch, and fresh counters were flagged)
    sys.path.insert(0, ROOT)
    import bench
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("vanilla_f32", {})
    if committed.get("launch_stamp") == bench.launch_stamp()["stamp"] and committed.get("grid_workgroups") == 2048:
        assert r["traffic_stale"] is False and r["pmc_grid_workgroups"] == 2048
        model = json.load(open(os.path.join(ROOT, "profiles", "issue_model.json"))).get("vanilla_f32", {})
        if model.get("launch_stamp") == committed["launch_stamp"] and (model.get("cross_check_ok") or model.get("rescaled_to_counters")):
            assert "issue_frac" in r
    c = d["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 1e6
    assert abs(d["fp64"]["price"] - BS) < 0.05
    # strong-scaling rows (C4, C5 and 10x) and the C library's own multi-GPU path are part of the N=1 line
    rows = {x["config"]: x for x in d["strong"]["rows"]}
    base = {"C4", "C4x10", "C5", "C5x10"}
    assert set(rows) == base | {c + "_n32" for c in base} and rows["C4"]["paths_priced"] == 10 ** 9 and rows["C5x10"]["paths_priced"] == 10 ** 8
    assert 9.70 < rows["C4"]["value"] < 9.74 and 0.1895 < rows["C5"]["value"] < 0.1905
    assert 9.70 < rows["C4_n32"]["value"] < 9.74 and 0.1895 < rows["C5_n32"]["value"] < 0.1905      # the reference's dp arithmetic: same prices
    assert rows["C4_n32"]["wall_ms_median"] < 0.8 * rows["C4"]["wall_ms_median"] and rows["C4_n32"]["normals"] == "f32"
    assert rows["C4x10"]["wall_ms_median"] == pytest.approx(10 * rows["C4"]["wall_ms_median"], rel=0.1)
    # every row is measured hot (pre-heat) with >= 10 timed calls; the base sizes cold as well, and cold is never faster by much
    assert all(x["reps"] >= 10 and x["preheat_ms"] == 300.0 for x in rows.values())
    assert all(("cold" in rows[c]) == (not c.startswith(("C4x10", "C5x10"))) for c in rows)
    assert rows["C5"]["cold"]["wall_ms_median"] > 0.9 * rows["C5"]["wall_ms_median"]
    # ... and, at N = 1, what one rank does at N = 2, 4, 8 (shard 0 of S): the device side of the scaling curve
    sh = {(x["config"], x["shard_of"]): x for x in d["strong"]["shard_rows"]}
    assert set(sh) == {(c, S) for c in rows for S in (2, 4, 8)} and sh[("C4", 8)]["paths"] == 125000000
    assert all(0.5 < x["device_side_efficiency"] < 1.1 for x in sh.values()) and sh[("C4x10", 8)]["device_side_efficiency"] > 0.9
    assert sh[("C5", 8)]["device_side_efficiency"] > 0.9 and "cold" in sh[("C5", 8)]      # hot / hot: 0.96-0.99 measured
    cm = d["c_multi"]
    assert any("shard_of" in x for x in cm["rows"])
    assert cm["rc"] == 0 and any(x.get("workload", "").startswith("C4 basket") and x["devices"] == 1 for x in cm["rows"])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_share_the_gpu(scaling):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", scaling,
           "--steps", "30", "--warmup", "3", "--regions", "2", "--cpu-seconds", "0", "--fp64-steps", "4", "--strong-reps", "2", "--strong-preheat-ms", "30"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out.stdout)
    total = 2 * 30 * 10 ** 8 * (2 if scaling == "weak" else 1)     # two regions of 30 steps
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["paths_priced"] == total
    assert d["config"]["paths_per_gpu_per_step"] == (10 ** 8 if scaling == "weak" else 5 * 10 ** 7)
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]      # every path counted exactly once across ranks
    assert "cpu_baseline" not in d and "c_multi" not in d
    rows = {x["config"]: x for x in d["strong"]["rows"]}               # one call sharded over the two ranks
    assert d["strong"]["n_gpus"] == 2 and rows["C4"]["paths_per_gpu"] == 5 * 10 ** 8 and rows["C4"]["paths_priced"] == 10 ** 9
    assert 9.70 < rows["C4"]["value"] < 9.74 and 0.1895 < rows["C5"]["value"] < 0.1905


def test_bench_rccl_plumbing_world_of_one():
    """The exact launch line the driver uses for N > 1 (torch.distributed.run + RCCL), with one rank:
    process-group creation on the device, bucketed asynchronous all-reduces of the triple rows, the
    device barrier and the MAX over ranks all run through RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "5",
           "--cpu-seconds", "0", "--fp64-steps", "4", "--strong-reps", "2", "--strong-preheat-ms", "30"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out.stdout)
    assert d["n_gpus"] == 1 and d["regions"] == 5 and d["paths_priced"] == 5 * 60 * 10 ** 8
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]
    assert d["value"] > 1e11
