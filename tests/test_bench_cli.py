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
    # the headline is carried in full precision, everything else at 5 significant digits
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"] and line["timed_region_s"] == d["timed_region_s"]
    assert line["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-4)
    assert line["roofline"]["issue_frac"] == pytest.approx(d["roofline"]["issue_frac"], rel=1e-4)
    assert line["roofline"]["issue_model"]["frac_effective"] <= 1.0 and line["roofline"]["traffic_stale"] is False
    assert line["fp64"]["value"] == pytest.approx(d["fp64"]["value"], rel=1e-4)
    # strong_summary, column form: per config one [torch path, C library] pair per column of `cols` -- T(1), T(shard 0 of 8) and the
    # ratio T(1) / (8 T(shard)) hot and cold, named for what it is: two timings on ONE GPU, not an 8-GPU efficiency
    ss = line["strong_summary"]
    assert ss["cols"] == ["t1_ms", "t_shard8_ms", "shard8_device_side_eff_1gpu", "shard8_device_side_eff_1gpu_cold"] and ss["src"] == ["bench.py", "libmc_multi"]
    assert not any("eff8" in k for k in json.dumps(ss).replace("shard8_device_side_eff_1gpu", "").split('"'))
    rows = {r["config"]: r for r in d["strong"]["rows"]}
    assert set(ss) - {"cols", "src", "c_devices"} == set(rows)
    sh = {(r["config"], r["shard_of"]): r for r in d["strong"]["shard_rows"]}
    csh = {(r["config"], r["shard_of"]): r for r in d["c_multi"]["rows"] if "shard_of" in r}
    for c in ("C4", "C5", "C4_n32", "C5_n32"):
        t1, t8, eff, cold = ss[c]
        assert t1[0] == pytest.approx(rows[c]["wall_ms_median"], rel=1e-4)
        assert t8[0] == pytest.approx(sh[(c, 8)]["wall_ms_median"], rel=1e-4)
        assert eff == [pytest.approx(sh[(c, 8)]["device_side_efficiency"], rel=1e-4), pytest.approx(csh[(c, 8)]["device_side_efficiency"], rel=1e-4)]
        assert cold == [pytest.approx(sh[(c, 8)]["cold"]["device_side_efficiency"], rel=1e-4),
                        pytest.approx(csh[(c, 8)]["cold"]["device_side_efficiency"], rel=1e-4)]
    assert len(ss["C4x10"]) == 3          # the 10x sizes are measured hot only


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
        assert set(line["dropped_for_size"]) <= {"strong_summary.c_devices", "strong_summary", "ranks", "fp64", "device", "configs"}
    else:
        assert line["strong_summary"]["c_devices"]["C4"]["8"][1] == pytest.approx(0.91235)
    # and a record that cannot be made to fit is an error, not a long line
    d["config"]["workload"] = "x" * 5000
    with pytest.raises(RuntimeError):
        bench.compact_line(d, "bench_detail.json")


# ---- the round-6 record: all five BASELINE configs in the driver's line, N > 1 lines that explain themselves ----
def _fixture6():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r06.json")))


def test_compact_line_of_the_round_6_run_carries_every_baseline_config():
    """BASELINE.json names five configs; configs[0] is the CPU run (tests/test_oracle_golden.py), the other four are in the line, each
    with its flop fraction, its issue fractions (at 2.4 GHz and at the clock measured while its kernel ran), its price error and the
    compiled reference's CPU path timed beside it."""
    import bench
    d = _fixture6()
    line = bench.compact_line(d, "bench_detail.json")
    _check_line(line)
    assert "dropped_for_size" not in line
    cf = line["configs"]
    assert list(cf) == ["C2", "C3", "C4", "C5"]
    for name, e in cf.items():
        for k in ("paths_per_s", "kernel_us", "frac", "issue_frac", "issue_frac_clk", "sclk_mhz", "price", "err", "vs", "cpu_baseline"):
            assert k in e, (name, k)
        assert 0.1 < e["frac"] < 0.5 and 0.5 < e["issue_frac"] <= e["issue_frac_clk"] < 1.0, (name, e)   # a ceiling: never reached
        assert 1500 < e["sclk_mhz"] <= 2400
        assert e["cpu_baseline"]["kind"] == "reference" and e["cpu_baseline"]["cores"] == 1 and e["cpu_baseline"]["value"] > 1e4
        assert e["paths_per_s"] / e["cpu_baseline"]["value"] > 1e3
    assert cf["C2"]["paths_per_s"] == pytest.approx(d["value"], rel=1e-4) and cf["C2"]["vs"] == "BS" and cf["C2"]["err"] < 1e-3
    assert cf["C2"]["single_call"][1] < cf["C2"]["paths_per_s"]        # one lone call: no second stream hides its ramp and tail
    assert cf["C3"]["err"] < 3e-3 and cf["C4"]["err"] < 3e-3 and cf["C5"]["err"] < 3e-4
    full = d["configs"]
    assert full["C4"]["cpu"]["object"].endswith("libref_f32_n16.so") and "cpu_dp" in full["C4"]["cpu"]     # the sp object, and why
    assert full["C5"]["cpu"]["object"].endswith("libref_f64_n3.so")


def test_issue_ceiling_is_below_every_measured_kernel_time():
    """The ceiling is a bound: for every workload of tools/bench_all.sh (tests/golden/bench_all_r06.json: kernel alone, ceiling at
    2.4 GHz, clock measured during the launches) ceiling <= measured, also when priced at the measured clock (one
    row sat at 1.000 / 1.014 with the ubench's mixed-stream costs)."""
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_all_r06.json")))["workloads"]
    assert len(rows) >= 10
    for w, r in rows.items():
        assert r["ceiling_us"] < r["kernel_us"], w
        assert r["ceiling_us"] * 2400.0 / r["sclk_mhz"] < r["kernel_us"], (w, r)
        assert r["issue_frac_at_measured_clock"] == pytest.approx(r["ceiling_us"] * 2400.0 / r["sclk_mhz"] / r["kernel_us"], rel=1e-3)


def test_compact_line_of_an_n_rank_strong_run_explains_itself():
    """N > 1: per strong config the ranks' own shard times, what the collective adds, T(1) on rank 0 and the efficiency split; the
    24-byte all-reduce alone.  Built from the round-6 record with the keys strong_scaling_block writes at N > 1."""
    import bench
    d = _fixture6()
    d["n_gpus"] = 8
    for k in ("cpu_baseline", "cpu_all_cores", "c_multi", "configs"):
        d.pop(k, None)
    d["strong"]["shard_rows"] = []
    for r in d["strong"]["rows"]:
        r["paths_per_gpu"] = r["paths_total"] // 8
        t1 = r["wall_ms_median"]
        r.update(t_shard_ms=[t1 / 8 * 0.99, t1 / 8 * 1.01], collective_ms=0.041234, t1_ms_rank0=t1, eff=t1 / (8 * (t1 / 8 * 1.01 + 0.041234)),
                 eff_device_side=1 / 1.01, wall_ms_median=t1 / 8 * 1.01 + 0.041234)
    d["strong"]["allreduce_us"] = {"median": 38.123456, "p10": 35.1, "p90": 44.9, "calls": 200, "backend": "nccl", "what": "..."}
    d.update(world_size=8, backend="nccl", rccl_version="2.26.6",
             ranks=[{"rank": i, "device": i, "pci": "0000:%02x:00.0" % (5 + 16 * i), "host": "mi355x-node-0123456789abcdef"} for i in range(8)])
    line = bench.compact_line(d, "bench_detail.json")
    assert len(json.dumps(line)) <= 4096 and "dropped_for_size" not in line
    ss = line["strong_summary"]
    assert ss["allreduce_us"]["median"] == pytest.approx(38.123, rel=1e-4) and ss["allreduce_us"]["calls"] == 200
    for c in ("C4", "C5", "C4_n32", "C5_n32"):
        e = ss[c]
        assert set(e) >= {"wall_ms_median", "paths_per_gpu", "t_shard_ms", "collective_ms", "t1_ms_rank0", "eff", "eff_device_side"}
        assert e["eff"] < e["eff_device_side"] <= 1.0 and e["t_shard_ms"][0] <= e["t_shard_ms"][1]
        assert e["wall_ms_median"] == pytest.approx(e["t_shard_ms"][1] + e["collective_ms"], rel=1e-3)


def test_scale_report_reads_the_lines_of_one_and_eight_ranks():
    """tools/scale_report.py on a 1-rank and an 8-rank line: weak efficiency from the values, strong efficiency of every config from the
    N = 1 line's T(1), and the N-rank line's own decomposition next to it."""
    import io
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_report
    d = _fixture6()
    one = bench.compact_line(d, "bench_detail.json")
    d8 = copy.deepcopy(d)
    d8["n_gpus"], d8["value"] = 8, 7.6 * d["value"]
    for k in ("cpu_baseline", "cpu_all_cores", "c_multi", "configs"):
        d8.pop(k, None)
    d8["strong"]["shard_rows"] = []
    for r in d8["strong"]["rows"]:
        t1 = r["wall_ms_median"]
        r["paths_per_gpu"] = r["paths_total"] // 8
        r.update(t_shard_ms=[t1 / 8 * 0.99, t1 / 8 * 1.01], collective_ms=0.041, t1_ms_rank0=t1, eff=t1 / (8 * (t1 / 8 * 1.01 + 0.041)),
                 eff_device_side=1 / 1.01, wall_ms_median=t1 / 8 * 1.01 + 0.041)
    d8["strong"]["allreduce_us"] = {"median": 38.1, "p10": 35.0, "p90": 45.0, "calls": 200}
    d8.update(world_size=8, backend="nccl", ranks=[{"rank": i, "device": i, "pci": "0000:%02x:00.0" % (5 + 16 * i), "host": "node"} for i in range(8)])
    out = io.StringIO()
    scale_report.report([one, bench.compact_line(d8, "bench_detail.json")], out)
    text = out.getvalue()
    assert " 8 " in text and "0.9500" in text and "nccl / 8" in text                 # weak efficiency 7.6 / 8
    c5 = next(line for line in text.splitlines() if line.startswith("C5 "))
    t1 = next(r["wall_ms_median"] for r in d["strong"]["rows"] if r["config"] == "C5")
    assert f"{t1 / (8 * (t1 / 8 * 1.01 + 0.041)):.4f}" in c5 and "0.9901" in c5 and "38.1000" in c5
    assert "C5x10" in text and "cold" in text
