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
    """The ONE `{` line of stdout: at most 4096 bytes (the driver keeps an 8 KB tail; round 4's 32 KB line came back parsed: null)."""
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    assert len(lines[0]) <= 4096, len(lines[0])
    return json.loads(lines[0])


def _detail(line, tmp):
    """The full record the line names (`detail`): every row; the compact line is stored in it as `line`."""
    assert line["detail"] == str(tmp)
    d = json.load(open(tmp))
    assert d["line"] == line
    return d


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_contract_single_gpu(tmp_path):
    tmp = tmp_path / "bench_detail.json"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "4", "--cpu-seconds", "0.5",
                          "--fp64-steps", "5", "--detail-file", str(tmp)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stderr) < 2000 and "{" not in out.stderr, out.stderr[-2000:]     # no record on stderr either
    line = _json_line(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "fp64", "configs", "strong_summary", "detail"):
        assert k in line, k
    assert "dropped_for_size" not in line
    assert line["n_gpus"] == 1 and line["steps"] == 40 and line["warmup"] == 4 and line["unit"] == "paths/s" and line["dtype"] == "f32"
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["scaling"] == "weak" and "workload" in line["config"]
    assert line["config"]["detail"] == "brief" and line["devices_visible"] >= 1 and len(line["device"]["pci"]) >= 12
    assert line["world_size"] == 1 and line["regions"] == 5 and line["paths_priced"] == 5 * 40 * 10 ** 8 and line["value"] > 1e11
    lr, lc = line["roofline"], line["cpu_baseline"]
    assert lr["bound"] == "valu" and lr["unit"] == "TFLOP/s" and lr["peak"] == 157.3 and 0 < lr["frac"] < 1 and "traffic" in lr
    assert lc["cores"] == 1 and lc["kind"] in ("reference", "port") and lc["value"] > 1e6 and lc["sample"]
    ss = line["strong_summary"]
    assert set(ss) == {"cols", "src", "C4", "C4x10", "C5", "C5x10", "C4_n32", "C5_n32"}
    assert ss["cols"] == ["t1_ms", "t_shard8_ms", "shard8_device_side_eff_1gpu", "shard8_device_side_eff_1gpu_cold"]
    for c in ("C4", "C5", "C4_n32", "C5_n32"):       # per column [bench.py's torch path, the C library]
        t1, t8, eff, cold = ss[c]
        assert len(t1) == 2 and all(v and v > 0 for v in t1 + t8), ss[c]
        assert all(0.5 < v < 1.1 for v in eff + cold), ss[c]
    assert ss["C5"][2][0] > 0.9 and ss["C4"][2][1] > 0.9          # hot / hot: 0.96-1.02 measured
    assert ss["C4"][0][1] == pytest.approx(ss["C4"][0][0], rel=0.05)      # the two harnesses agree
    # BASELINE configs[1..4], each with its fractions, price error and the reference's CPU path (0.5 s samples here)
    cf = line["configs"]
    assert list(cf) == ["C2", "C3", "C4", "C5"]
    for name, e in cf.items():
        assert e["paths_per_s"] > 1e8 and e["kernel_us"] > 0 and 0.05 < e["frac"] < 0.6 and e["cpu_baseline"]["cores"] == 1, (name, e)
        assert e["paths_per_s"] > 1e3 * e["cpu_baseline"]["value"] and 1000 < e["sclk_mhz"] <= 2450
    assert cf["C2"]["err"] < 2e-3 and cf["C3"]["err"] < 5e-3 and cf["C4"]["err"] < 5e-3 and cf["C5"]["err"] < 5e-4
    assert 40 < cf["C2"]["single_call"][0] < 200

    # ---- the full record ---------------------------------------------------------------------------------
    d = _detail(line, tmp)
    assert d["value"] == line["value"] and d["ms_per_step"] == line["ms_per_step"]
    # the reported step time is the median of the five regions
    assert len(d["region_ms_per_step"]) == 5 and d["ms_per_step"] == sorted(d["region_ms_per_step"])[2]
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]
    r = d["roofline"]
    assert r["achieved"] == pytest.approx(15.5 * 1e8 / (r["avg_kernel_us"] * 1e-6) / 1e12, rel=1e-9)
    assert lr["achieved"] == pytest.approx(r["achieved"], rel=1e-4) and lr["avg_kernel_us"] == pytest.approx(r["avg_kernel_us"], rel=1e-4)
    # the per-launch duration is the kernel alone on the device (50 launches one at a time), never above the step
    # period by more than the launch gap; the in-region samples (every launch of a short run) are kept for the record
    assert r["kernel_samples"] == 50 and r["duration_basis"].startswith("exclusive")
    assert 40 < r["avg_kernel_us"] < 70 and r["in_region"]["kernel_samples"] >= 8
    assert r["in_region"]["avg_kernel_us"] >= r["avg_kernel_us"] * 0.95
    assert 0 < r["effective"]["frac"] < 1 and d["timed_region_s"] == pytest.approx(d["ms_per_step"] * 40e-3, rel=1e-9)
    assert "not measured in this run" in r["traffic_source"]
    if r["traffic_stale"]:     # device code, launch-shape rules or grid changed since the committed PMC passes: the model is withheld
        assert "issue_frac" not in r and "traffic_stale_note" in r and "issue_frac" not in lr
    elif "issue_frac" in r:    # a CEILING now: the cheapest measured cost of every opcode of the hot loop -- never above 1, on the step period either
        assert 0.5 < r["issue_frac"] <= 1.0 and r["issue_model"]["frac_effective"] <= 1.0
        assert r["issue_model"]["ceiling_us"] <= r["issue_model"]["typical_us"]
        assert lr["issue_frac"] == pytest.approx(r["issue_frac"], rel=1e-4)
    else:
        assert "issue_model_withheld" in r
    assert r["grid_workgroups"] == 2048 and len(r["launch_stamp"]) == 16
    # the clock the card held while the headline kernel ran back to back (amdgpu hwmon), and the ceiling priced at it
    assert 1500 < r["sclk_mhz"] <= 2450 and lr["sclk_mhz"] == pytest.approx(r["sclk_mhz"], rel=1e-4)
    if "issue_frac" in r:
        assert r["issue_frac_at_measured_clock"] == pytest.approx(r["issue_frac"] * 2400.0 / r["sclk_mhz"], rel=1e-6) and r["issue_frac_at_measured_clock"] < 1.0
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
    assert abs(d["fp64"]["price"] - BS) < 0.05
    # strong-scaling rows and the C library's own multi-GPU path are in the full record (brief: no 10x sizes on fp32 normals)
    rows = {x["config"]: x for x in d["strong"]["rows"]}
    assert set(rows) == set(ss) - {"cols", "src"} and rows["C4"]["paths_priced"] == 10 ** 9 and rows["C5x10"]["paths_priced"] == 10 ** 8
    assert 9.70 < rows["C4"]["value"] < 9.74 and 0.1895 < rows["C5"]["value"] < 0.1905
    assert 9.70 < rows["C4_n32"]["value"] < 9.74 and 0.1895 < rows["C5_n32"]["value"] < 0.1905      # the reference's dp arithmetic: same prices
    assert rows["C4_n32"]["wall_ms_median"] < 0.8 * rows["C4"]["wall_ms_median"] and rows["C4_n32"]["normals"] == "f32"
    assert rows["C4x10"]["wall_ms_median"] == pytest.approx(10 * rows["C4"]["wall_ms_median"], rel=0.1)
    # every row is measured hot (pre-heat) with >= 10 timed calls; the base sizes cold as well, and cold is never faster by much
    assert all(x["reps"] >= 10 and x["preheat_ms"] == 300.0 for x in rows.values())
    assert all(("cold" in rows[c]) == (not c.startswith(("C4x10", "C5x10"))) for c in rows)
    assert rows["C5"]["cold"]["wall_ms_median"] > 0.9 * rows["C5"]["wall_ms_median"]
    # ... and, at N = 1, what one rank does at N = 8 (shard 0 of 8): the device side of the scaling curve
    sh = {(x["config"], x["shard_of"]): x for x in d["strong"]["shard_rows"]}
    assert set(sh) == {(c, 8) for c in rows} and sh[("C4", 8)]["paths"] == 125000000
    assert all(0.5 < x["device_side_efficiency"] < 1.1 for x in sh.values()) and sh[("C4x10", 8)]["device_side_efficiency"] > 0.9
    assert sh[("C5", 8)]["device_side_efficiency"] > 0.9 and "cold" in sh[("C5", 8)]      # hot / hot: 0.96-0.99 measured
    assert ss["C5"][1][0] == pytest.approx(sh[("C5", 8)]["wall_ms_median"], rel=1e-4)
    # the configs block of the full record: where every number of the line's `configs` comes from
    dc = d["configs"]
    assert dc["C4"]["paths_per_s"] == pytest.approx(10 ** 9 / (rows["C4"]["wall_ms_median"] * 1e-3), rel=1e-9) and dc["C4"]["price"] == rows["C4"]["value"]
    assert dc["C5"]["vs"].startswith("closed form") and dc["C3"]["steps"] == [40, 3] and dc["C3"]["vs_confidence_95"] < 2e-3
    assert dc["C3"]["cpu"]["object"].endswith("libref_f32_n4.so") and dc["C4"]["cpu"]["object"].endswith("libref_f32_n16.so") and dc["C4"]["cpu"]["cpu_dp"] > 1e5
    cm = d["c_multi"]
    assert cm["rc"] == 0 and "--brief" in cm["command"]
    crows = [x for x in cm["rows"] if "config" in x]
    assert {x["config"] for x in crows} == {"C4", "C5", "C4_n32", "C5_n32"} and {x["shard_of"] for x in crows if "shard_of" in x} == {8}
    assert sum("create_s" in x for x in cm["rows"]) == mc_devices_pow2(line["devices_visible"])      # one RCCL set-up per G, none for the shard rows


def mc_devices_pow2(visible):
    n, g = 0, 1
    while g <= visible:
        n, g = n + 1, g * 2
    return n


def test_bench_detail_full_adds_the_rows_brief_leaves_out(tmp_path):
    """--detail full: the 10x sizes on fp32 normals, shard 0 of 2 and of 4, the -O0 CPU build -- short reps here (plumbing)."""
    tmp = tmp_path / "d.json"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--regions", "2", "--cpu-seconds", "0.3",
                          "--fp64-steps", "0", "--strong-reps", "2", "--strong-preheat-ms", "20", "--c-multi-seconds", "0", "--detail", "full",
                          "--detail-file", str(tmp)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out.stdout)
    d = _detail(line, tmp)
    assert line["config"]["detail"] == "full" and "fp64" not in line
    base = {"C4", "C4x10", "C5", "C5x10"}
    assert {x["config"] for x in d["strong"]["rows"]} == base | {c + "_n32" for c in base} == set(line["strong_summary"]) - {"cols", "src"}
    assert {(x["shard_of"]) for x in d["strong"]["shard_rows"]} == {2, 4, 8}
    assert "value_at_O0" in d["cpu_baseline"] and line["cpu_baseline"]["value_at_O0"] > 1e6


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_share_the_gpu(scaling, tmp_path):
    tmp = tmp_path / "d.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", scaling,
           "--steps", "30", "--warmup", "3", "--regions", "2", "--cpu-seconds", "0", "--fp64-steps", "4", "--strong-reps", "2", "--strong-preheat-ms", "30",
           "--detail-file", str(tmp)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    d = _detail(line, tmp)
    # the roster: who took part (both ranks on the one GPU here, so ONE distinct device -- on a node it is N)
    assert line["world_size"] == 2 and line["backend"] == "gloo" and [r["rank"] for r in line["ranks"]] == [0, 1]
    assert all(r["device"] == 0 and len(r["pci"]) >= 12 and r["host"] for r in line["ranks"]) and d["distinct_devices"] == 1
    assert line["n_gpus"] == 2 and line["strong_summary"]["C4"]["paths_per_gpu"] == 5 * 10 ** 8
    total = 2 * 30 * 10 ** 8 * (2 if scaling == "weak" else 1)     # two regions of 30 steps
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["paths_priced"] == total
    assert d["config"]["paths_per_gpu_per_step"] == (10 ** 8 if scaling == "weak" else 5 * 10 ** 7)
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]      # every path counted exactly once across ranks
    assert "cpu_baseline" not in d and "c_multi" not in d
    rows = {x["config"]: x for x in d["strong"]["rows"]}               # one call sharded over the two ranks
    assert d["strong"]["n_gpus"] == 2 and rows["C4"]["paths_per_gpu"] == 5 * 10 ** 8 and rows["C4"]["paths_priced"] == 10 ** 9
    assert 9.70 < rows["C4"]["value"] < 9.74 and 0.1895 < rows["C5"]["value"] < 0.1905
    # the N > 1 line explains itself: every rank's own shard time, what the collective adds, T(1) on rank 0,
    # the efficiency split, and the 24-byte all-reduce alone -- here two gloo ranks sharing one GPU, so only the plumbing counts
    ssn = line["strong_summary"]
    assert ssn["allreduce_us"]["calls"] == 200 and ssn["allreduce_us"]["median"] > 0
    for c in ("C4", "C5", "C4_n32", "C5_n32"):
        e, full = ssn[c], rows[c]
        assert len(full["t_shard_ms_by_rank"]) == 2 and e["t_shard_ms"] == [pytest.approx(min(full["t_shard_ms_by_rank"]), rel=1e-4),
                                                                              pytest.approx(max(full["t_shard_ms_by_rank"]), rel=1e-4)]
        assert e["collective_ms"] == pytest.approx(e["wall_ms_median"] - e["t_shard_ms"][1], rel=1e-2, abs=1e-3)
        assert e["t1_ms_rank0"] > 0 and e["eff"] == pytest.approx(e["t1_ms_rank0"] / (2 * e["wall_ms_median"]), rel=1e-3)
        assert e["eff_device_side"] == pytest.approx(e["t1_ms_rank0"] / (2 * e["t_shard_ms"][1]), rel=1e-3)
    assert "t_shard_ms" not in ssn["C4x10"]            # base sizes only


def test_bench_rccl_plumbing_world_of_one(tmp_path):
    """The exact launch line the driver uses for N > 1 (torch.distributed.run + RCCL), with one rank:
    process-group creation on the device, bucketed asynchronous all-reduces of the triple rows, the
    device barrier and the MAX over ranks all run through RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "5",
           "--cpu-seconds", "0", "--fp64-steps", "4", "--strong-reps", "2", "--strong-preheat-ms", "30", "--c-multi-seconds", "0",
           "--detail-file", str(tmp_path / "d.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out.stdout)
    assert d["world_size"] == 1 and d["backend"] == "nccl" and d["rccl_version"]
    assert d["n_gpus"] == 1 and d["regions"] == 5 and d["paths_priced"] == 5 * 60 * 10 ** 8
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]
    assert d["value"] > 1e11


def _device_count():
    sys.path.insert(0, ROOT)
    import montecarlocuda_amd as mc
    return mc._lib.lib().mc_device_count()


def test_bench_two_ranks_rccl_one_rank_per_device(tmp_path):
    """The driver's own N = 2 launch line -- torch.distributed.run, backend nccl (= RCCL over xGMI), one rank per device --
    on a box that HAS two devices: the first contact of this code with a communicator of more than one rank.  Skips on the
    one-GPU boxes this build is developed on (there the gloo rehearsal above covers the bookkeeping)."""
    n = _device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 visible devices for a real 2-rank RCCL run ({n} visible)")
    tmp = tmp_path / "d.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "10", "--regions", "3",
           "--fp64-steps", "20", "--strong-reps", "10", "--detail-file", str(tmp)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(out.stdout[-5000:])
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    d = _detail(line, tmp)
    assert line["world_size"] == 2 and line["backend"] == "nccl" and line["rccl_version"]
    assert [r["device"] for r in line["ranks"]] == [0, 1] and d["distinct_devices"] == 2 and len({r["pci"] for r in line["ranks"]}) == 2
    assert line["paths_priced"] == 3 * 100 * 2 * 10 ** 8 and abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]
    rows = {x["config"]: x for x in d["strong"]["rows"]}
    assert rows["C4"]["paths_per_gpu"] == 5 * 10 ** 8 and rows["C4"]["paths_priced"] == 10 ** 9 and 9.70 < rows["C4"]["value"] < 9.74
    assert rows["C5"]["paths_priced"] == 10 ** 7 and 0.1895 < rows["C5"]["value"] < 0.1905
    assert line["value"] > 2e11          # two devices, weak scaling: no less than one device's worth
    # the first measured collective: the line must explain whatever efficiency it shows
    ssn = line["strong_summary"]
    print("allreduce_us", ssn["allreduce_us"], {c: ssn[c] for c in ("C4", "C5")})
    assert 1 < ssn["allreduce_us"]["median"] < 5000
    for c in ("C4", "C5", "C4_n32", "C5_n32"):
        assert 0.3 < ssn[c]["eff"] <= ssn[c]["eff_device_side"] * 1.02 and ssn[c]["collective_ms"] > -0.05


def test_bench_four_ranks_share_the_gpu(tmp_path):
    """Four ranks on the one GPU (gloo; the box admits six processes on the card): the roster has four entries, every path of the
    strong rows is priced exactly once over mc_shard_range(total, rank, 4), the weak headline counts four shards per step."""
    tmp = tmp_path / "d.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--steps", "20", "--warmup", "2",
           "--regions", "2", "--cpu-seconds", "0", "--fp64-steps", "0", "--strong-reps", "2", "--strong-preheat-ms", "20", "--detail-file", str(tmp)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    d = _detail(line, tmp)
    assert line["n_gpus"] == 4 and line["world_size"] == 4 and [r["rank"] for r in line["ranks"]] == [0, 1, 2, 3]
    assert line["paths_priced"] == 2 * 20 * 4 * 10 ** 8 and line["config"]["global_paths_per_step"] == 4 * 10 ** 8
    assert abs(d["price"] - BS) < 3.5 / 1.96 * d["confidence_95"]
    rows = {x["config"]: x for x in d["strong"]["rows"]}
    assert rows["C4"]["paths_per_gpu"] == 25 * 10 ** 7 and rows["C4"]["paths_priced"] == 10 ** 9 and 9.70 < rows["C4"]["value"] < 9.74
    assert rows["C5"]["paths_per_gpu"] == 25 * 10 ** 5 and rows["C5"]["paths_priced"] == 10 ** 7 and 0.1895 < rows["C5"]["value"] < 0.1905
    # the decomposition of every base row over the four ranks, and the 24-byte collective alone (gloo here)
    ssn = line["strong_summary"]
    assert ssn["allreduce_us"]["calls"] == 200
    for c in ("C4", "C5", "C4_n32", "C5_n32"):
        assert len(rows[c]["t_shard_ms_by_rank"]) == 4 and ssn[c]["t_shard_ms"][0] <= ssn[c]["t_shard_ms"][1] and ssn[c]["t1_ms_rank0"] > 0
        assert ssn[c]["eff_device_side"] == pytest.approx(ssn[c]["t1_ms_rank0"] / (4 * ssn[c]["t_shard_ms"][1]), rel=1e-3)
