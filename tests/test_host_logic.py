"""Host-side logic of the product that needs no GPU: sharding, closing formulas, Cholesky."""
import numpy as np
import pytest

from conftest import fromhex, load_golden


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


def test_shard_range_partitions_exactly(mc):
    for total in (0, 1, 7, 10**8, 10**9 + 7, 2**52 - 1, 2**63 + 12345):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for rank in range(world):
                first, count = mc.shard_range(total, rank, world)
                assert first == pos == (rank * total) // world   # SURVEY 8e rule
                pos += count
            assert pos == total


def test_closing_matches_oracle_and_reference_formula(mc, po):
    for s, s2, n, disc in ((1095721.8224204443, 36197748.012070745, 100000, 0.9523),
                           (368.1204752756539, 106.37671195984049, 2000, 1.0)):
        assert mc.closing(s, s2, n, disc) == po.closing(s, s2, n, disc)
    # against the compiled reference through the golden CVA case (closing is the last step of it)
    c = [c for c in load_golden("ref_mc.json")["cases"] if c["kind"] == "cva" and c["X"] == "f64"][1]
    r = po.host_cva("f64", c["cva"], c["paths"], c["seed"])
    e, ci = mc.closing(r["sum"], r["sum2"], r["n"], 1.0)
    assert e == fromhex(c["expected"]) and ci == fromhex(c["confidence"])


def test_chol_matches_reference_golden(mc):
    for c in load_golden("ref_chol.json")["cases"]:
        m = [[fromhex(x) for x in row] for row in c["c"]]
        want = np.array([[fromhex(x) for x in row] for row in c["a"]])
        got, bad = mc.chol(m, c["X"])
        assert (got.astype(np.float64) == want).all(), (c["X"], c["n"], c["name"])
        zero_cols = int((np.diag(want) == 0).sum())
        assert bad == zero_cols


def test_chol_flags_indefinite_input(mc):
    # the reference driver's N=4 pattern is not positive definite
    m = [[1.0 if i == j else (0.5 if max(i, j) % 2 == 0 else -0.5) for j in range(4)] for i in range(4)]
    _, bad = mc.chol(m)
    assert bad > 0
    _, bad = mc.chol(np.full((4, 4), 0.5) + 0.5 * np.eye(4))
    assert bad == 0


def test_basket_control_mean_matches_oracle_and_simulation(mc, po):
    """Closed-form mean of the geometric-basket control: engine helper == oracle (fp64 formula), and it
    equals the simulated mean of the geometric payoff (difference of the oracle's two estimators)."""
    for X in ("f64", "f32"):
        for n in (1, 3, 4, 16):
            corr = np.full((n, n), 0.4) + 0.6 * np.eye(n)
            L = po.chol(X, corr)
            b = dict(s=[90.0 + 3 * i for i in range(n)], v=[0.15 + 0.02 * i for i in range(n)], p=L.tolist(),
                     d=[0.01 * (i % 3 - 1) for i in range(n)], w=[1.0 + 0.1 * i for i in range(n)], k=100.0 * n, t=1.5, r=0.03)
            assert mc.basket_control_mean(b, X) == pytest.approx(po.basket_control_mean(X, b), rel=1e-13)
        _, plain = po.dev_basket("f64", b, 5, 0, 200000, want_paths=False)
        _, ctrl = po.dev_basket("f64", b, 5, 0, 200000, want_paths=False, control=True)
        assert ctrl["confidence"] < 0.15 * plain["confidence"]
        assert abs(ctrl["expected"] - plain["expected"]) < 3.5 / 1.96 * plain["confidence"]
    with pytest.raises(mc.McError, match=r"w\[a\] > 0"):
        mc.basket_control_mean(dict(b, w=[1.0] * 15 + [-0.5]))


def test_factor_from_cov_matches_reference_golden(mc, po):
    """mc_factor_from_cov_* (SURVEY 8f-2): vols and Cholesky factor of the correlation matrix from a covariance
    matrix, bit for bit what the reference's Chol returns for the hand-normalised matrix (tests/golden/ref_cov.json),
    with the non-positive pivots counted instead of silently zeroed."""
    for c in load_golden("ref_cov.json")["cases"]:
        cov = np.array([[fromhex(x) for x in row] for row in c["cov"]])
        v, p, bad = mc.factor_from_cov(cov, c["X"])
        key = (c["X"], c["n"], c["name"])
        assert (v.astype(np.float64) == np.array([fromhex(x) for x in c["v"]])).all(), key
        assert (p.astype(np.float64) == np.array([[fromhex(x) for x in row] for row in c["a"]])).all(), key
        assert bad == c["bad_pivots"], key
        ov, _, oa, obad = po.factor_from_cov(c["X"], cov)
        assert (ov == v).all() and (oa == p).all() and obad == bad
    # only the lower triangle is read (as Chol does, MonteCarloHost.c:96-99)
    cov = np.array([[0.09, 99.0], [0.03, 0.04]])
    v, p, bad = mc.factor_from_cov(cov)
    assert bad == 0 and v.tolist() == [0.3, 0.2] and p[1, 0] == 0.03 / (0.3 * 0.2)
    # round trip: (diag(v) p)(diag(v) p)^T is the covariance again
    rng = np.random.default_rng(11)
    g = rng.standard_normal((7, 12)) * 0.25
    cov = g @ g.T
    v, p, bad = mc.factor_from_cov(cov)
    m = v[:, None] * p
    assert bad == 0 and np.abs(m @ m.T - cov).max() < 1e-14
    for badcov in ([[0.0, 0.0], [0.0, 1.0]], [[-1.0, 0.0], [0.0, 1.0]], [[float("nan"), 0.0], [0.0, 1.0]], [[1.0, 0.0], [float("inf"), 1.0]]):
        with pytest.raises(ValueError):
            mc.factor_from_cov(badcov)


def test_multi_gpu_host_control_flow_without_a_gpu(tmp_path):
    """libmc_multi.so's GPU-free control flow (csrc/mc_multi_host.hpp), driven by a plain g++ program on fake slots in
    ordinary memory: the read-back polling loop of run_sharded (G = 3 with one slot never written must fail after ONE settle
    of the streams; a slot delivered during the settle still counts), and the launcher-thread crew (every device's job on
    its own thread, no hand-off lost while the workers spin nor after they parked).  One-GPU boxes never run G > 1 on real
    devices, so this is where that logic is covered."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "multi_host_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-I", os.path.join(root, "montecarlocuda_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "multi_host_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "all checks passed" in out.stdout and "FAILED" not in out.stdout, out.stdout + out.stderr


def test_multi_gpu_host_control_flow_is_race_free_under_tsan(tmp_path):
    """The same program under ThreadSanitizer: the crew's hand-off words, the claim protocol by which the calling thread takes
    over the job of a late worker (round 4) and the parked-worker wake-up are data-race free.  Sanitizers run on the CPU build
    only (GPU sanitizers are not available on the pool)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "multi_host_check_tsan"
    build = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread", "-I", os.path.join(root, "montecarlocuda_amd", "csrc"),
                            os.path.join(root, "tests", "cpp", "multi_host_check.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in (build.stderr or "").lower():
        pytest.skip("libtsan is not installed")
    assert build.returncode == 0, build.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all checks passed" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout[-3000:] + out.stderr[-3000:]


def test_pmc_stamp_covers_the_launch(tmp_path):
    """The stamp that decides `traffic_stale` (bench.launch_stamp) covers what a launch IS: the device code object (the
    .hip_fatbin section of the built library), the launch-shape rules (csrc/mc_launch_shape.hpp: grid scales, kernel-family
    limits) and HIPFLAGS -- shown on a patched COPY of the tree (VERDICT r03 "next" #3; round 3 hashed four headers and
    missed the grid rules that had moved into mc_api.hip)."""
    import os
    import re
    import shutil
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    src = os.path.join(root, "montecarlocuda_amd", "csrc")
    copy = tmp_path / "tree" / "montecarlocuda_amd" / "csrc"
    copy.mkdir(parents=True)
    for f in ("libmc_mi355x.so", "mc_launch_shape.hpp", "Makefile"):
        shutil.copy(os.path.join(src, f), copy / f)
    base = bench.launch_stamp(root)
    assert base == bench.launch_stamp(str(tmp_path / "tree")) and len(base["device_code_sha256"]) == 64     # where the tree lies does not matter

    def stamp_after(name, edit, binary=False):
        path = copy / name
        old = path.read_bytes() if binary else path.read_text()
        path.write_bytes(edit(old)) if binary else path.write_text(edit(old))
        out = bench.launch_stamp(str(tmp_path / "tree"))
        path.write_bytes(old) if binary else path.write_text(old)
        return out
    # a grid scale: vanilla f64 from 12 to 16 workgroups per CU
    shape = stamp_after("mc_launch_shape.hpp", lambda t: re.sub(r"GRID_SCALE_VANILLA_F64 = 3", "GRID_SCALE_VANILLA_F64 = 4", t, count=1))
    assert shape["stamp"] != base["stamp"] and shape["launch_shape_sha256"] != base["launch_shape_sha256"]
    assert shape["device_code_sha256"] == base["device_code_sha256"]
    # an optimisation flag
    flags = stamp_after("Makefile", lambda t: t.replace("HIPFLAGS ?= -O3", "HIPFLAGS ?= -O2", 1))
    assert flags["stamp"] != base["stamp"] and "-O2" in flags["hipflags"] and "-O3" in base["hipflags"]
    # one byte of device code
    sect = bench.elf_section(os.path.join(src, "libmc_mi355x.so"), ".hip_fatbin")
    at = (copy / "libmc_mi355x.so").read_bytes().index(sect[:64]) + len(sect) // 2

    def flip(b):
        b = bytearray(b)
        b[at] ^= 1
        return bytes(b)
    code = stamp_after("libmc_mi355x.so", flip, binary=True)
    assert code["stamp"] != base["stamp"] and code["device_code_sha256"] != base["device_code_sha256"]
    # host-only bytes of the library (outside the code object) do not matter
    def flip_host(b):
        b = bytearray(b)
        b[16] ^= 1          # e_type of the ELF header
        return bytes(b)
    host = stamp_after("libmc_mi355x.so", flip_host, binary=True)
    assert host["stamp"] == base["stamp"]
    # ... and bench.py's comparison: a committed profile with another stamp, or another grid, is stale
    committed = {"launch_stamp": base["stamp"], "grid_workgroups": 2048}
    stale = lambda c, stamp, grid: c.get("launch_stamp") != stamp or c.get("grid_workgroups") not in (None, grid)   # noqa: E731
    assert not stale(committed, base["stamp"], 2048) and stale(committed, shape["stamp"], 2048) and stale(committed, base["stamp"], 3072)


def test_committed_issue_model_and_pmc_counts_describe_one_build():
    """profiles/issue_model.json (tools/issue_model.py) and profiles/pmc_traffic.json (tools/summarize_pmc.py) are read by
    bench.py as a pair: both must carry the same launch stamp for every workload, every ceiling must lie at or below the
    mixed-stream estimate, and the histogram's VALU count per path must agree with the counters' hot-loop slope within 1 % --
    or the model must say that it took its counts from the counters instead (the CVA date loops).  Whether the pair also
    describes the CURRENT build is bench.py's run-time check (traffic_stale); here a mismatch is only reported, because the
    counters can be re-collected on a GPU box alone."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = json.load(open(os.path.join(root, "profiles", "issue_model.json")))
    pmc = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    workloads = {"vanilla_f32", "vanilla_f64", "vanilla_f64_n32", "basket4_f32", "basket16_f32", "basket16_f64", "basket16_f64_n32",
                 "cva256_f64", "cva256_f64_n32", "cva256_f32"}
    assert workloads <= set(model) and workloads <= set(pmc)
    stamps = set()
    for w in sorted(workloads):
        m, p = model[w], pmc[w]
        assert m["launch_stamp"] == p["launch_stamp"], w
        stamps.add(m["launch_stamp"])
        assert 0 < m["ceiling_us"] <= m["typical_us"], w
        assert m["min_cycles_per_path"] <= m["typical_cycles_per_path"], w
        assert m.get("cross_check_ok") or m.get("rescaled_to_counters"), w
        slope = p["hot_loop_slope"]["valu_wave_insts_per_path"] * 64.0
        if m.get("cross_check_ok"):
            assert abs(slope / m["model_per_path"]["valu"] - 1.0) <= 0.01, w
        else:   # counts taken from the counters: the model's VALU per path IS the slope
            assert abs(m["valu_per_path"] / slope - 1.0) <= 1e-6 and abs(m["pmc_vs_model"] - 1.0) <= 0.05, w
        # 3674 = the CVA workloads' 1.25e6 paths: 3072 one-lane-per-path workgroups + 602 date-parallel ones for the last 4816 paths (cva_split_kernel)
        assert p["grid_workgroups"] in (2048, 3072, 3674, 6144, 12288) and p["group_size"] == 256, w
    assert len(stamps) == 1
    sys.path.insert(0, root)
    import bench
    live = bench.launch_stamp(root)["stamp"]
    if live not in stamps:
        print("NOTE: the committed counters describe another build than the one in the tree (bench.py will flag traffic_stale): "
              "re-run tools/collect_pmc_all.sh and tools/issue_model.py on a GPU box")


def test_cva_plan_rule_table(tmp_path):
    """csrc/mc_launch_shape.hpp: cva_plan, the cut between one lane per path and the date-parallel form of a CVA call, as a pure host
    function (hipcc, host code only, nothing launched): small calls whole (up to 7/4 wave-trips on a grid of 64 dates or more, 3/4 on a
    shorter one), the last partial trip of a call of up to 64 trips when it is at most 60 % of a trip, nothing on grids that fit one 8-date
    chunk, and the forced settings.  The GPU twin of this table is tests/test_gpu_cva_dates.py::test_small_call_runs_date_parallel_by_default."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "cva_plan_check"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "montecarlocuda_amd", "csrc"),
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "cva_plan_check.hip"), "-o", str(exe)])
    trip = 65536
    cases = {   # (paths, dates, real bytes, forced lanes) -> (one lane per path, date-parallel, log2 lanes)
        (2 * trip, 256, 8, 0): (2 * trip, 0, 0),                      # the reference driver's own call (dp/cvaOpt.cu:12-15): two trips
        (7 * trip // 4, 256, 8, 0): (0, 7 * trip // 4, 2),            # 7/4 trips: whole call date-parallel, lanes for ~4 waves per SIMD
        (7 * trip // 4 + 1, 256, 4, 0): (7 * trip // 4 + 1, 0, 0),    # one path more, a remainder above 60 % of a trip: one lane per path
        (4096, 256, 4, 0): (0, 4096, 5),                              # capped by the grid's 8-date chunks (32 of them)
        (4096, 25, 8, 0): (0, 4096, 2),                               # 25 dates = 4 chunks: at most 4 lanes
        (3 * trip // 4, 25, 4, 0): (0, 3 * trip // 4, 2),
        (trip, 25, 4, 0): (trip, 0, 0),
        (1250000, 256, 8, 0): (19 * trip, 1250000 - 19 * trip, 5),    # C5's shard of 8: 19 whole trips + 4816 paths date-parallel, one launch
        (1250000, 50, 8, 0): (1250000, 0, 0),                         # short grid: no split
        (10000000, 256, 8, 0): (10000000, 0, 0),                      # beyond 64 trips a trip is under 1.6 % of the call
        (19 * trip + 45000, 256, 8, 0): (19 * trip + 45000, 0, 0),    # remainder above 60 % of a trip
        (1000, 3, 8, 0): (1000, 0, 0),                                # one chunk: nothing to share
        (1000, 256, 8, 1): (1000, 0, 0),
        (1000, 256, 8, 16): (0, 1000, 4),
        (1000, 25, 8, 64): (0, 1000, 2),
        (0, 256, 8, 0): (0, 0, 0),
    }
    argv = [str(x) for k in cases for x in k]
    out = subprocess.run([str(exe)] + argv, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = {}
    for ln in out.stdout.splitlines():
        a, b = ln.split("->")
        got[tuple(int(x) for x in a.split())] = tuple(int(x) for x in b.split())
    assert got == cases
