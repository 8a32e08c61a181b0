"""cva_dates_kernel on a real MI355X: a CVA path's dates shared by L = 2 ... 64 adjacent lanes (csrc/mc_kernels.hpp; the
reference walks them serially in one thread, dp/MonteCarloKernel.cu:241-262), against the oracle's device formulas per path and
per sum, against the one-lane-per-path kernel, and on the REFERENCE's own normal stream against the numbers the compiled
MonteCarloHost.c printed (tests/golden/ref_mc.json).

Tolerances are the CVA bounds of tests/test_gpu_parity.py (f64: 1e-13 per path, 1e-12 relative on sums; f32: 2e-5, 3e-6):
the date-parallel form evaluates the same per-date operations on the same normals and differs only in the association of two
sums per path -- W_j = z_1 + ... + z_j is formed as (previous rounds + lanes below + own dates) and sum_j dp_j ee_j per lane
first, then over the lanes."""
import numpy as np
import pytest

from tests.conftest import fromhex, load_golden

pytestmark = pytest.mark.gpu

SEED = 0x4D435F4D49333535
CVA0 = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6)  # reference cvaOpt.cu:22-34
TOL = {"f32": dict(cva=2e-5, rel=3e-6), "f64": dict(cva=1e-13, rel=1e-12)}
LANES = [1, 2, 4, 8, 16, 32, 64]   # 1 = cva_kernel itself (the default rule would run a call this small date-parallel)
# the reference driver's grids (cvaOpt.cu:70-75), 256 (BASELINE C5), tiny ones, and grids either side of the eight-date trips of the
# fp64 one-lane loop (cva_path<double>: four Box-Muller pairs per trip while eight closed-form dates remain) and of the 8-date chunks;
# 700: the fp32 one-lane loop takes its pair rows from scalar registers there (above 16 KB of them), from LDS on every smaller grid
GRIDS = [1, 2, 3, 7, 8, 9, 10, 16, 17, 18, 24, 25, 50, 75, 250, 256, 500, 700]


def f64(a):
    return np.asarray(a, dtype=np.float64)


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


@pytest.fixture()
def eng(mc):
    e = mc.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_grid", GRIDS)
def test_every_lane_count_matches_the_oracle_per_path_and_per_sum(eng, po, X, n_grid):
    """250 dates in f64 end with a NEGATIVE residual maturity (the last date contributes nothing), 256 with exactly 0 (intrinsic
    value: one lane of the path prices it), 500 in f32 with a small positive one; 1, 2, 3 dates fit one 8-date
    chunk, so every forced lane count falls back to one lane per path there (same bound)."""
    c = dict(CVA0, n_grid=n_grid)
    n = 3001   # not a multiple of any lane-group count: the last wave has dead path slots
    want, o = po.dev_cva(X, c, SEED, 11, n)
    for lanes in LANES:
        eng.set_cva_date_lanes(lanes)
        got = f64(eng.cva_paths(c, n, SEED, 11, X))
        assert np.abs(got - f64(want)).max() <= TOL[X]["cva"], (lanes, n_grid)
        e = eng.cva(c, n, SEED, 11, X)
        assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and e.sum2 == pytest.approx(o["sum2"], rel=2 * TOL[X]["rel"]), lanes
        assert e.expected == pytest.approx(o["expected"], rel=TOL[X]["rel"])
        assert e.confidence == pytest.approx(o["confidence"], rel=10 * TOL[X]["rel"])


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_grid_stride_and_range_position(eng, po, X):
    """More paths than lane groups in the grid (a small context: 16 workgroups), a range that starts high and crosses a multiple
    of 2^32 (two launches sharing one ticket set)."""
    import montecarlocuda_amd as mc
    c = dict(CVA0, n_grid=64)
    with mc.Engine(0, 16) as small:
        small.set_cva_date_lanes(8)
        n = 5000    # 16 workgroups x 32 path slots = 512 per trip
        got = f64(small.cva_paths(c, n, SEED, 0, X))
        want, o = po.dev_cva(X, c, SEED, 0, n)
        assert np.abs(got - f64(want)).max() <= TOL[X]["cva"]
        assert small.cva(c, n, SEED, 0, X).sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])
    first = (1 << 32) - 1000
    eng.set_cva_date_lanes(4)
    got = f64(eng.cva_paths(c, 2500, SEED, first, X))
    want, o = po.dev_cva(X, c, SEED, first, 2500)
    assert np.abs(got - f64(want)).max() <= TOL[X]["cva"]
    assert eng.cva(c, 2500, SEED, first, X).sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_antithetic_and_other_inputs(eng, po, X):
    c = dict(s=90.0, k=100.0, r=0.01, v=0.4, t=2.0, defint=0.1, lgd=0.45, n_grid=37)
    for lanes in (1, 2, 8):   # 1: cva_kernel's own antithetic instantiation (a call this small would otherwise never reach it)
        eng.set_cva_date_lanes(lanes)
        eng.set_antithetic(False)
        got = f64(eng.cva_paths(c, 4000, 5, 0, X))
        want, _ = po.dev_cva(X, c, 5, 0, 4000)
        assert np.abs(got - f64(want)).max() <= 4 * TOL[X]["cva"]
        eng.set_antithetic(True)
        c64 = dict(CVA0, n_grid=64)
        got = f64(eng.cva_paths(c64, 2001, SEED, 11, X))
        want, o = po.dev_cva(X, c64, SEED, 11, 2001, antithetic=True)
        assert np.abs(got - f64(want)).max() <= TOL[X]["cva"]
        assert eng.cva(c64, 2001, SEED, 11, X).sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])


def test_fp32_normals_in_the_fp64_kernel(eng, po):
    """The reference's own dp arithmetic (a float normal widened to double, dp/MonteCarloKernel.cu:250) through the date-parallel
    kernel: generator policy GenPhiloxF32N, four normals per block.  Same widened normals as the one-lane-per-path kernel draws,
    so the two agree to fp64 rounding; the oracle's float normals (glibc) sit 2e-5 away, as in tests/test_gpu_normals_f32.py."""
    c = dict(CVA0, n_grid=256)
    eng.set_normals("f32")
    eng.set_cva_date_lanes(1)
    one = f64(eng.cva_paths(c, 2000, SEED, 3, "f64"))
    with po.normals_f32():
        want, _ = po.dev_cva("f64", c, SEED, 3, 2000)
    assert np.abs(one - f64(want)).max() <= 2e-5
    for lanes in (4, 32):
        eng.set_cva_date_lanes(lanes)
        got = f64(eng.cva_paths(c, 2000, SEED, 3, "f64"))
        assert np.abs(got - one).max() <= 1e-13, lanes
        assert eng.cva(c, 2000, SEED, 3, "f64").sum == pytest.approx(one.sum(), rel=1e-12)


def test_xorwow_keeps_one_lane_per_path(eng, po):
    """One XORWOW sequence per lane: a path cannot be entered in the middle, whatever the setting says."""
    c = dict(CVA0, n_grid=64)
    eng.set_generator("xorwow", 0)
    plain = eng.cva(c, 4096, SEED, 0, "f64").sum
    eng.set_cva_date_lanes(8)
    assert eng.cva(c, 4096, SEED, 0, "f64").sum == plain


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_split_call_main_beside_tail(mc, po, X):
    """The automatic rule on a call that ends in a partial wave-trip: the leading whole trips on cva_kernel, the remainder on
    cva_dates_kernel beside it (the context's second stream), one ticket set, one triple.  Against the one-lane-per-path call
    (sums to rounding) and, per path, against the oracle across the cut."""
    c = dict(CVA0, n_grid=256)
    with mc.Engine(0) as e:
        trip = 64 * 4 * e.info()["compute_units"]
        n = 3 * trip + 4816               # C5's shard of 8 has the same remainder: 1 250 000 = 19 x 65 536 + 4816
        e.set_cva_date_lanes(1)
        one = e.cva(c, n, SEED, 0, X)
        e.set_cva_date_lanes(0)
        for rep in range(3):              # repeated: the tickets must be back at zero, the second stream joined
            split = e.cva(c, n, SEED, 0, X)
            assert split.n == n
            assert split.sum == pytest.approx(one.sum, rel=TOL[X]["rel"]) and split.sum2 == pytest.approx(one.sum2, rel=2 * TOL[X]["rel"])
        lo = 3 * trip - 700               # per path across the cut: 700 paths of the main launch, all of the tail
        got = f64(e.cva_paths(c, n, SEED, 0, X))[lo:]
        want, _ = po.dev_cva(X, c, SEED, lo, n - lo)
        assert np.abs(got - f64(want)).max() <= TOL[X]["cva"]
        # the asynchronous entry point on a caller's stream, two calls back to back, then a different product on the same context
        import torch
        out = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
        st = torch.cuda.Stream()
        struct, _keep = e.prepared("cva", X, c)
        for i in range(2):
            e.launch("cva", X, struct, SEED, 0, n, out[i].data_ptr(), st.cuda_stream)
        st.synchronize()
        for i in range(2):
            s_, q_, n_ = out[i].tolist()
            assert n_ == n and s_ == pytest.approx(one.sum, rel=TOL[X]["rel"]) and q_ == pytest.approx(one.sum2, rel=2 * TOL[X]["rel"])
        van = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
        v, o = e.vanilla(van, 100003, SEED, 0, X), po.dev_vanilla(X, van, SEED, 0, 100003, want_paths=False)[1]
        assert v.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])


def test_small_call_runs_date_parallel_by_default(mc, po):
    """The automatic rule (csrc/mc_launch_shape.hpp: cva_plan) prices a small call date-parallel as a whole: up to 7/4 wave-trips on a grid
    of 64 dates or more and 3/4 of one on a shorter grid.  The reference driver's own call (dp/cvaOpt.cu:12-15: 131 072 paths = 2 trips;
    grids 25 ... 500, :70-75) keeps one lane per path in both precisions (measured slower otherwise: profiles/r06_cva_call_latency.log).
    Same estimate either way."""
    with mc.Engine(0) as e:
        cases = ((250, 131072, False), (250, 114688, True), (25, 65536, False), (25, 49152, True))
        for X, n_grid, paths, parallel in [(X, *k) for X in ("f64", "f32") for k in cases]:
            c = dict(CVA0, n_grid=n_grid)
            e.set_cva_date_lanes(0)
            auto = e.cva(c, paths, SEED, 0, X)
            wgs_auto = e.last_launch()[0]
            e.set_cva_date_lanes(1)
            one = e.cva(c, paths, SEED, 0, X)
            assert e.last_launch()[0] == paths // 256
            assert (wgs_auto > paths // 256) == parallel, (X, n_grid, paths, wgs_auto)
            assert auto.sum == pytest.approx(one.sum, rel=TOL[X]["rel"]) and auto.sum2 == pytest.approx(one.sum2, rel=2 * TOL[X]["rel"])


def test_reference_stream_meets_the_goldens(eng, po):
    """The reference's own normals (glibc rand() + Box-Muller, MonteCarloHost.c:117-121) through the date-parallel kernel
    (generator policy GenExternal): the numbers the compiled MonteCarloHost.c printed, as tests/test_gpu_from_normals.py
    demands of the one-lane-per-path kernel (host-order hook: the reference CPU loop prices the exposure at the lagged spot)."""
    cases = [k for k in load_golden("ref_mc.json")["cases"] if k["kind"] == "cva" and k["paths"] <= 20000]
    assert cases
    for k in cases:
        X, c = k["X"], dict(k["cva"])
        n_grid = c["n_grid"]
        z = po.host_gaussians(X, k["seed"], k["paths"] * n_grid).reshape(k["paths"], n_grid)
        for lanes in (1, 4, 16):
            eng.set_cva_date_lanes(lanes)
            est, vals = eng.cva_from_normals(c, z, X, host_order=True)
            closed = po.ref_close(X, vals, 0, c["r"], c["t"])
            want = fromhex(k["expected"])
            tol = 2e-11 if X == "f64" else 1e-3     # f32: the reference forms dp_j by float cancellation
            assert closed["expected"] == pytest.approx(want, rel=tol), (X, lanes, n_grid)
            assert est.expected == pytest.approx(want, rel=max(tol, 5e-5 if X == "f32" else tol))
