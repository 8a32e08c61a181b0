"""Compatibility mode: the reference's launch geometry and per-thread XORWOW streams (mc_*_run_grid_*, mc_grid_normals).

The reference's sample depends on (numBlocks, numThreads): dp/MonteCarloKernel.cu:285-290 seeds one XORWOW state per thread,
curand_init(blockIdx.x + gridDim.x, threadIdx.x, 0); thread t of a block prices paths t, t + T, ... < N_PATH (:146,191,240)
and draws curand_normal() one after the other (:68,78,250).  cuRAND is not in the image; its AMD counterpart -- the one a
HIP build of the reference calls through hipRAND -- is, as headers, so THAT is the pin:

  (1) the engine's per-thread normal streams equal rocRAND's own rocrand_init + rocrand_normal run on this GPU by an
      independent little program (tests/cpp/rocrand_grid_device.hip), BIT FOR BIT;
  (2) and equal the oracle's restatement (orc_grid_normals, itself bit-equal to rocRAND's host-callable engine:
      tests/test_rocrand_xcheck.py) within the device transcendentals' error: 4e-6 absolute;
  (3) a grid call's (sum, sum2) equal, BITWISE, those of the from-normals hook fed the arrangement the reference's loops
      imply (thread t -> paths t, t + T, ...; `draws` consecutive normals per path; the kept second Box-Muller member
      carried from one path to the next) rebuilt on the host from (1)'s streams -- so the arrangement kernel is right,
      and everything downstream is the hot kernels' code, parity-tested in test_gpu_from_normals.py / test_gpu_parity.py;
  (4) per path against the oracle's device formulas on those normals, at the bounds of test_gpu_parity.py;
  (5) prices against closed forms within the confidence interval.
Round 4: the call is FUSED -- the reference's launch itself, every thread's stream in registers (mc_grid.hpp: grid_*_kernel);
(3) is asserted on round 3's STAGED form (normals through HBM: the checker, mc_context_set_grid_form), and
  (6) the fused form's per-path values are the staged form's BIT FOR BIT (mc_*_paths_grid_*), its (sum, sum2) agree within
      the order of the additions (3e-6 fp32 / 1e-12 fp64), for every product, both precisions, every basket size that has a
      fused kernel, geometries with T not dividing N_PATH, N_PATH < T, partial waves (T = 1, 7, 100) and 1024-thread blocks.
Equality with an NVIDIA run of the reference is not claimed: cuRAND seeds XORWOW with other constants ("parity unpinned").
"""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA0 = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6)
BS_EXACT = 10.386270784322328
NP = {"f32": np.float32, "f64": np.float64}
PAY = {"f32": 2e-6, "f64": 1e-14}       # x spot
CVA_ABS = {"f32": 2e-5, "f64": 1e-13}
SUMS = {"f32": 3e-6, "f64": 1e-12}


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


@pytest.fixture(scope="module")
def eng(mc):
    e = mc.Engine(0)
    yield e
    e.close()


def basket_inputs(mc, n, X, rho=0.5):
    v = [0.3 if i % 2 == 0 else 0.2 for i in range(n)]
    L, bad = mc.chol(np.full((n, n), rho) + (1 - rho) * np.eye(n), X)
    assert bad == 0
    return dict(s=[100.0] * n, v=v, p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n, k=100.0, t=1.0, r=0.048790164)


def both_forms(eng, prod, inputs, G, T, per_block, X, fused_exists=True):
    """(staged estimate, fused estimate): the two forms of one launch-geometry call, per-path values compared bit for bit."""
    try:
        eng.set_grid_form("staged")
        st = eng.run_grid(prod, inputs, G, T, per_block, X)
        st_vals = eng.paths_grid(prod, inputs, G, T, per_block, X)
        eng.set_grid_form("fused")
        if not fused_exists:
            import montecarlocuda_amd as mc
            with pytest.raises(mc.McError, match="no fused kernel"):
                eng.run_grid(prod, inputs, G, T, per_block, X)
            eng.set_grid_form("auto")      # falls back to the staged form: same bits as the staged call
            au = eng.run_grid(prod, inputs, G, T, per_block, X)
            assert (au.sum, au.sum2, au.n) == (st.sum, st.sum2, st.n)
            return st, au
        fu = eng.run_grid(prod, inputs, G, T, per_block, X)
        fu_vals = eng.paths_grid(prod, inputs, G, T, per_block, X)
    finally:
        eng.set_grid_form("auto")
    U = np.uint32 if X == "f32" else np.uint64
    assert np.array_equal(st_vals.view(U), fu_vals.view(U)), (prod, X, G, T, per_block, int((st_vals.view(U) != fu_vals.view(U)).sum()))
    assert fu.n == st.n == G * per_block
    assert fu.sum == pytest.approx(st.sum, rel=SUMS[X], abs=1e-300) and fu.sum2 == pytest.approx(st.sum2, rel=SUMS[X], abs=1e-300)
    # the fused call's own sums are the sums of its per-path values
    v = fu_vals.astype(np.float64)
    assert fu.sum == pytest.approx(v.sum(), rel=SUMS[X], abs=1e-300) and fu.sum2 == pytest.approx((v * v).sum(), rel=SUMS[X], abs=1e-300)
    au = eng.run_grid(prod, inputs, G, T, per_block, X)      # the default form is the fused one
    assert (au.sum, au.sum2) == (fu.sum, fu.sum2)
    return st, fu


def cva_draws(t, n_grid, X):
    """dates whose `t -= dt` is still >= 0 in the build's arithmetic (dp/MonteCarloKernel.cu:249)"""
    R = NP[X]
    t = R(t)
    dt = R(t / R(n_grid))
    draws = 0
    for _ in range(n_grid):
        t = R(t - dt)
        if not t >= 0:
            break
        draws += 1
    return draws


def test_streams_equal_rocrand_device_engine_bit_for_bit(eng, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_normal.h"):
        pytest.skip("rocRAND headers / hipcc not available")
    exe = tmp_path / "rocrand_grid_device"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-w", "--offload-arch=gfx950", "-ffp-contract=fast",
                           os.path.join(ROOT, "tests", "cpp", "rocrand_grid_device.hip"), "-o", str(exe)])
    for G, T, count in ((1, 1, 7), (3, 64, 9), (5, 100, 4), (128, 256, 5), (2, 1024, 3)):
        out = subprocess.run([str(exe), str(G), str(T), str(count)], capture_output=True, text=True, check=True).stdout.split()
        want = np.array([int(x, 16) for x in out], dtype=np.uint32).reshape(G, T, count)
        got = eng.grid_normals(G, T, count).view(np.uint32)
        assert (got == want).all(), (G, T, count)


@pytest.mark.parametrize("G,T,count", [(1, 1, 8), (2, 3, 5), (7, 64, 9), (256, 5, 4)])
def test_streams_match_oracle(eng, po, G, T, count):
    got = eng.grid_normals(G, T, count)
    want = po.grid_normals(G, T, count)
    assert np.isfinite(got).all()
    assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 4e-6
    # blocks are seeded apart, threads are subsequences of one seed: no two streams coincide
    flat = got.reshape(G * T, count)
    assert len({r.tobytes() for r in flat}) == G * T


# the last four make the fused kernels cut every thread's stream into 32, 2, 2 and 8 pieces (mc_api.hip: grid_pieces)
GEOMS = [(1, 1, 1), (1, 64, 64), (3, 64, 200), (4, 100, 37), (2, 256, 1000), (5, 7, 1001), (2, 1024, 5000), (3, 128, 7),
         (1, 64, 70000), (3, 100, 9001), (2, 256, 20000), (4, 128, 33001)]


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("G,T,per_block", GEOMS)
def test_vanilla_grid_is_the_hot_kernel_on_the_reference_arrangement(mc, eng, po, X, G, T, per_block):
    streams = eng.grid_normals(G, T, po.grid_draws_per_thread(T, per_block, 1))
    z = po.grid_path_normals(streams, per_block, 1).reshape(-1).astype(NP[X])       # dp: the float normal widened
    e, fused = both_forms(eng, "vanilla", VAN, G, T, per_block, X)
    h, vals = eng.vanilla_from_normals(VAN, z, X)
    assert e.n == G * per_block == h.n
    assert (e.sum, e.sum2) == (h.sum, h.sum2)
    want, o = po.dev_vanilla_on_normals(X, VAN, z)
    assert np.abs(vals.astype(np.float64) - want).max() <= PAY[X] * VAN["s"]
    assert e.sum == pytest.approx(o["sum"], rel=SUMS[X], abs=1e-9) and fused.sum == pytest.approx(o["sum"], rel=SUMS[X], abs=1e-9)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_assets", [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 33])
def test_basket_grid_carries_the_kept_normal_across_paths(mc, eng, po, X, n_assets):
    """Odd asset counts: a path's last Box-Muller pair is split with the next path of the same thread."""
    b = basket_inputs(mc, n_assets, X)
    for G, T, per_block in ((3, 64, 200), (2, 10, 33), (1, 256, 100), (2, 64, 4100), (1, 32, 8200)):   # the last two: 2 and 8 pieces per thread
        streams = eng.grid_normals(G, T, po.grid_draws_per_thread(T, per_block, n_assets))
        g = po.grid_path_normals(streams, per_block, n_assets).astype(NP[X])
        e, fused = both_forms(eng, "basket", b, G, T, per_block, X, fused_exists=n_assets <= 16)
        h, vals = eng.basket_from_normals(b, g, X)
        assert e.n == G * per_block
        assert (e.sum, e.sum2) == (h.sum, h.sum2), (G, T, per_block)
        want, o = po.dev_basket_on_normals(X, b, g, 0)
        assert np.abs(vals.astype(np.float64) - want).max() <= PAY[X] * 100.0 * 4
        assert e.sum == pytest.approx(o["sum"], rel=SUMS[X]) and fused.sum == pytest.approx(o["sum"], rel=SUMS[X])


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_grid", [1, 2, 3, 5, 50, 256])
def test_cva_grid_draws_only_for_the_dates_that_draw(mc, eng, po, X, n_grid):
    """`if ((t -= dt) >= 0)` (dp/MonteCarloKernel.cu:249): with t = 1 and three dates the float build's last date has
    t < 0 and draws nothing -- the thread's stream then moves on by two normals per path, not three."""
    c = dict(CVA0, n_grid=n_grid)
    draws = cva_draws(c["t"], n_grid, X)
    assert 1 <= draws <= n_grid
    if X == "f32" and n_grid == 3:
        assert draws == 2
    for G, T, per_block in ((3, 32, 70), (2, 32, 1100)):      # the second: 2 pieces per thread
        streams = eng.grid_normals(G, T, po.grid_draws_per_thread(T, per_block, draws))
        z = np.zeros((G * per_block, n_grid), dtype=NP[X])
        z[:, :draws] = po.grid_path_normals(streams, per_block, draws)
        e, fused = both_forms(eng, "cva", c, G, T, per_block, X)
        h, vals = eng.cva_from_normals(c, z, X)
        assert (e.sum, e.sum2, e.n) == (h.sum, h.sum2, G * per_block)
        want, o = po.dev_cva_on_normals(X, c, z.astype(np.float64), 0)
        assert np.abs(vals.astype(np.float64) - want).max() <= CVA_ABS[X]


def test_grid_prices_agree_with_closed_forms(mc, eng, po):
    # the reference's own shape: 256 blocks of 256 threads (vanillaOpt.cu), 2^22 paths
    for X in ("f32", "f64"):
        e = eng.run_grid("vanilla", VAN, 256, 256, 16384, X)
        assert e.n == 1 << 22
        assert abs(e.expected - BS_EXACT) < 3.5 / 1.96 * e.confidence
    c = dict(CVA0, n_grid=50)
    e = eng.run_grid("cva", c, 64, 128, 2048, "f64")
    ref = eng.cva(c, 1 << 17, precision="f64")
    assert abs(e.expected - ref.expected) < 3.5 / 1.96 * math.hypot(e.confidence, ref.confidence)
    b = basket_inputs(mc, 4, "f32")
    e = eng.run_grid("basket", b, 64, 128, 4096, "f32")
    ref = eng.basket(b, 1 << 20, precision="f32")
    assert abs(e.expected - ref.expected) < 3.5 / 1.96 * math.hypot(e.confidence, ref.confidence)
    # a different geometry is a different sample of the same size
    assert eng.run_grid("vanilla", VAN, 128, 256, 32768, "f64").sum != eng.run_grid("vanilla", VAN, 256, 256, 16384, "f64").sum
    # repeatable, and the context's generator settings neither matter nor are disturbed
    a = eng.run_grid("vanilla", VAN, 16, 64, 1000, "f64")
    eng.set_generator("xorwow", 5)
    try:
        assert eng.run_grid("vanilla", VAN, 16, 64, 1000, "f64").sum == a.sum
        x1 = eng.vanilla(VAN, 4096, precision="f64").sum
    finally:
        eng.set_generator("philox", 0)
    p1 = eng.vanilla(VAN, 4096, precision="f64").sum
    assert x1 != p1
    assert eng.run_grid("vanilla", VAN, 16, 64, 1000, "f64").sum == a.sum


def test_fused_form_speed_shape_and_state_cache(mc, eng):
    """The reference drivers' shape (512 blocks of 128 threads, vanillaOpt.cu:13-15) at 1e8 paths: the fused call no longer
    moves 0.4 GB of normals through HBM (0.40 ms in round 3); a geometry's start states are set up once and kept for the last
    four geometries (a second call of a cached geometry is not slower than the first by a set-up)."""
    per_block = 10 ** 8 // 512
    eng.run_grid("vanilla", VAN, 512, 128, per_block, "f32")            # first use: state set-up
    best = min(eng.run_grid("vanilla", VAN, 512, 128, per_block, "f32").kernel_ms for _ in range(5))
    assert best < 0.30, best                                               # measured ~0.1 ms; the bound is loose on purpose
    geoms = [(64, 64), (32, 128), (16, 256), (8, 512), (4, 1024)]          # five geometries through a cache of four
    first = {g: eng.run_grid("vanilla", VAN, g[0], g[1], 1000, "f64").sum for g in geoms}
    for g in geoms + geoms[::-1]:
        assert eng.run_grid("vanilla", VAN, g[0], g[1], 1000, "f64").sum == first[g]
    e = eng.run_grid("vanilla", VAN, 512, 128, per_block, "f32")
    assert abs(e.expected - BS_EXACT) < 3.5 / 1.96 * e.confidence and e.n == 512 * per_block


def test_grid_argument_errors(mc, eng):
    for G, T, per in ((0, 64, 10), (1, 0, 10), (1, 1025, 10), (1, 64, 0), (1 << 20, 64, 1), (1024, 64, 1 << 22)):
        with pytest.raises(mc.McError, match="launch geometry"):
            eng.run_grid("vanilla", VAN, G, T, per, "f32")
    eng.set_antithetic(True)
    try:
        with pytest.raises(mc.McError, match="plain estimator"):
            eng.run_grid("vanilla", VAN, 4, 64, 100, "f32")
    finally:
        eng.set_antithetic(False)
    with pytest.raises(mc.McError):
        eng.grid_normals(1, 1, 0)


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_legacy_symbols_under_the_reference_geometry(mc, eng, po, X, monkeypatch):
    """MC_RNG=xorwow_grid: dev_vanillaOpt / dev_basketOpt / dev_cvaEquityOption honour (numBlocks, numThreads) as the
    reference does; N_PATH = sims / numBlocks (dp/MonteCarloKernel.cu:491,508,524)."""
    import ctypes as C
    monkeypatch.setenv("MC_RNG", "xorwow_grid")
    L = C.CDLL(mc._lib.LEGACY[X])
    OptionData, MultiOptionData, OptionValue, CVA = po.ref_types(X, 3)
    L.dev_vanillaOpt.argtypes = [C.POINTER(OptionData), C.c_int, C.c_int, C.c_int]
    L.dev_vanillaOpt.restype = OptionValue
    L.dev_basketOpt.argtypes = [C.POINTER(MultiOptionData), C.c_int, C.c_int, C.c_int]
    L.dev_basketOpt.restype = OptionValue
    L.dev_cvaEquityOption.argtypes = [C.POINTER(CVA), C.c_int, C.c_int, C.c_int]
    L.dev_cvaEquityOption.restype = OptionValue
    R = NP[X]
    blocks, threads, sims = 48, 128, 48 * 1000 + 17
    o = OptionData(*[VAN[k] for k in "skrvt"])
    v = L.dev_vanillaOpt(C.byref(o), blocks, threads, sims)
    e = eng.run_grid("vanilla", VAN, blocks, threads, 1000, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))
    assert L.dev_vanillaOpt(C.byref(o), blocks, 64, sims).Expected != v.Expected      # numThreads shapes the sample
    b = basket_inputs(mc, 3, X)
    m = MultiOptionData()
    for i in range(3):
        m.s[i], m.v[i], m.d[i], m.w[i] = b["s"][i], b["v"][i], b["d"][i], b["w"][i]
        for j in range(3):
            m.p[i][j] = b["p"][i][j]
    m.k, m.t, m.r = b["k"], b["t"], b["r"]
    v = L.dev_basketOpt(C.byref(m), blocks, threads, sims)
    e = eng.run_grid("basket", b, blocks, threads, 1000, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))
    c = dict(CVA0, n_grid=50)
    s = CVA(c["defint"], c["lgd"], 0, OptionData(*[c[k] for k in "skrvt"]), c["n_grid"])
    v = L.dev_cvaEquityOption(C.byref(s), blocks, threads, sims)
    e = eng.run_grid("cva", c, blocks, threads, 1000, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))


def test_python_mirror_follows_the_same_switch(mc, eng, monkeypatch):
    monkeypatch.setenv("MC_RNG", "xorwow_grid")
    v = mc.dev_vanillaOpt(mc.OptionData(**VAN), 48, 128, 48 * 1000 + 17)
    e = eng.run_grid("vanilla", VAN, 48, 128, 1000, "f64")
    assert (v.Expected, v.Confidence) == (e.expected, e.confidence)
