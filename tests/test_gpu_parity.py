"""Hop B and C on a real MI355X: the HIP engine, called through the C ABI, against the oracle on
identical Philox counters (per path and per sum), against closed-form Black-Scholes, and -- at
BASELINE.json's full sizes -- through size-independent properties (shard additivity, geometry
independence, statistical agreement with analytic targets).

TOLERANCES (north_star: "within a stated fp32/fp64 tolerance").  Both sides evaluate the same
real-number formulas on the same uniforms; they differ only in rounding:
  f32  device uses v_log/v_sqrt/v_sin/v_cos/v_exp_f32 (about 1 ulp each) and base-2 folding;
       the oracle uses glibc float libm.  Measured on MI355X (profiles/r01_first_contact_explore.log):
       normals 4.8e-7 abs, payoffs 3e-5..4.6e-5 abs on values of O(10..100), CVA 2.5e-6 abs,
       sums 1e-8..4e-7 rel.  Stated bounds carry a 4-6x margin:
           normal        |dz|  <= 2e-6
           payoff        |dp|  <= 2e-6 * spot      (2e-4 at S=100)
           CVA per path  |dv|  <= 2e-5, sums/estimates rel <= 3e-6
  f64  device uses its own log / sincos / exp / sqrt (mc_math_f64.hpp, <= 1-2 ulp), the oracle glibc (< 1 ulp).
       Measured: normals 2.9e-15, payoffs 1.1e-13 abs, CVA 6e-15, sums 5e-15 rel.  Stated bounds:
           normal <= 2e-14, payoff <= 1e-14 * spot (1e-12 at S=100), CVA <= 1e-13,
           sums/estimates rel <= 1e-12
Integer work (Philox words -> uniforms) is exact on both sides; any mismatch there would show
as O(1) differences, not as rounding.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)          # reference vanillaOpt.cu:22-26
CVA0 = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6)  # reference cvaOpt.cu:22-34
BS_EXACT = 10.386270784322328   # exact Black-Scholes for VAN
TOL = {
    "f32": dict(z=2e-6, pay=2e-6, cva=2e-5, rel=3e-6),
    "f64": dict(z=2e-14, pay=1e-14, cva=1e-13, rel=1e-12),
}


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


@pytest.fixture(scope="module")
def eng(mc):
    e = mc.Engine(0)
    yield e
    e.close()


SEED = 0x4D435F4D49333535


def basket_inputs(mc, n, X, rho=0.5):
    """S=100, w=1/n, vols alternating 0.3/0.2, K=100, r=0.048790164, T=1, equicorrelation."""
    v = [0.3 if i % 2 == 0 else 0.2 for i in range(n)]
    corr = np.full((n, n), rho) + (1 - rho) * np.eye(n)
    L, bad = mc.chol(corr, X)
    assert bad == 0
    return dict(s=[100.0] * n, v=v, p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n, k=100.0, t=1.0, r=0.048790164)


def f64(a):
    return np.asarray(a, dtype=np.float64)


# ---- the generator ------------------------------------------------------------------------
@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("domain,block,first", [(1, 0, 0), (2, 3, 12345), (3, 63, (1 << 32) - 100), (1, 0, (7 << 32) + 5)])
def test_normals_match_oracle(eng, po, X, domain, block, first):
    n = 512
    got = f64(eng.normals(SEED, domain, first, n, block, X))
    want = np.array([po.dev_normals(X, SEED, domain, first + u, block) for u in range(n)], dtype=np.float64)
    assert np.abs(got - want).max() <= TOL[X]["z"]


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_normals_are_standard(eng, X):
    z = f64(eng.normals(SEED + 1, 1, 0, 1 << 20, 0, X)).ravel()
    n = z.size
    assert abs(z.mean()) < 5 / math.sqrt(n)
    assert abs(z.var() - 1) < 5 * math.sqrt(2 / n)
    assert abs((z ** 3).mean()) < 5 * math.sqrt(15 / n)
    assert abs((z ** 4).mean() - 3) < 5 * math.sqrt(96 / n)


# ---- vanilla ------------------------------------------------------------------------------
@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("first,n", [(0, 1), (0, 2), (1, 1), (3, 1), (1, 3), (2, 4), (3, 9), (5, 20000),
                                     ((1 << 34) - 7, 40), ((3 << 33) + 1, 1001)])
def test_vanilla_per_path_and_sums(eng, po, X, first, n):
    got = f64(eng.vanilla_paths(VAN, n, SEED, first, X))
    want, o = po.dev_vanilla(X, VAN, SEED, first, n)
    assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * VAN["s"]
    e = eng.vanilla(VAN, n, SEED, first, X)
    assert e.n == n
    assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"], abs=TOL[X]["pay"] * VAN["s"])
    assert e.sum2 == pytest.approx(o["sum2"], rel=TOL[X]["rel"], abs=TOL[X]["pay"] * VAN["s"] ** 2)
    if n > 1000:
        assert e.expected == pytest.approx(o["expected"], rel=TOL[X]["rel"])
        assert e.confidence == pytest.approx(o["confidence"], rel=TOL[X]["rel"])


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("opt", [dict(s=105.0, k=95.0, r=0.02, v=0.35, t=2.5), dict(s=40.0, k=55.0, r=0.0, v=0.6, t=0.25),
                                 dict(s=1.0, k=1.0, r=0.1, v=0.05, t=10.0), dict(s=100.0, k=100.0, r=0.05, v=0.0, t=1.0)])
def test_vanilla_other_options(eng, po, X, opt):
    n = 30000
    got = f64(eng.vanilla_paths(opt, n, 77, 0, X))
    want, o = po.dev_vanilla(X, opt, 77, 0, n)
    scale = opt["s"] * math.exp(abs(opt["r"]) * opt["t"] + 3 * opt["v"] * math.sqrt(opt["t"]))  # size of S_T in play
    assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * max(scale, opt["s"]) * 4
    e = eng.vanilla(opt, n, 77, 0, X)
    assert e.sum == pytest.approx(o["sum"], rel=4 * TOL[X]["rel"])


def test_vanilla_c2_full_size_vs_black_scholes(eng):
    """BASELINE configs[1]: 1e8 fp32 paths on one GPU vs closed form.  At 1e8 paths CI ~ 0.003, so
    the 1e-3 target is checked on the 1e10-path run (CI ~ 3e-4); both must sit inside 3.5 CI."""
    e = eng.vanilla(VAN, 10 ** 8, SEED, 0, "f32")
    assert e.confidence == pytest.approx(0.003022, rel=2e-3)
    assert abs(e.expected - BS_EXACT) < 3.5 / 1.96 * e.confidence
    big = eng.vanilla(VAN, 10 ** 10, SEED, 0, "f32")
    assert big.n == 10 ** 10
    assert abs(big.expected - BS_EXACT) < 1e-3
    assert abs(big.expected - BS_EXACT) < 3.5 / 1.96 * big.confidence


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_vanilla_shards_add_up_and_geometry_does_not_matter(mc, eng, X):
    """Size-independent properties at 1e8 paths: the union of 8 contiguous shards equals the whole
    range (what the multi-GPU path relies on), and the grid size only changes summation order."""
    total = 10 ** 8 + 3
    # f64: only the order of fp64 additions changes.  f32: a lane adds 16 scaled payoffs in fp32 before
    # each flush to fp64, and which 16 go together depends on the geometry; every such partial carries
    # a relative rounding error <= 16 * 2^-24, independent and zero-mean over ~6e6 partials
    # (expected ~1e-7 / sqrt(6e6) = 4e-11, measured 7e-11): bound 2e-9.
    rel = 1e-12 if X == "f64" else 2e-9
    whole = eng.vanilla(VAN, total, SEED, 0, X)
    s = s2 = 0.0
    n = 0
    for r in range(8):
        first, cnt = mc.shard_range(total, r, 8)
        e = eng.vanilla(VAN, cnt, SEED, first, X)
        s, s2, n = s + e.sum, s2 + e.sum2, n + e.n
    assert n == total
    assert s == pytest.approx(whole.sum, rel=rel) and s2 == pytest.approx(whole.sum2, rel=rel)
    with mc.Engine(0, blocks=311) as small:
        other = small.vanilla(VAN, total, SEED, 0, X)
    assert other.sum == pytest.approx(whole.sum, rel=rel) and other.sum2 == pytest.approx(whole.sum2, rel=rel)
    # bitwise reproducible for a fixed geometry
    again = eng.vanilla(VAN, total, SEED, 0, X)
    assert (again.sum, again.sum2) == (whole.sum, whole.sum2)
    # a different seed is a different sample
    assert eng.vanilla(VAN, total, SEED + 1, 0, X).sum != whole.sum


# ---- basket -------------------------------------------------------------------------------
@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_assets", [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 14, 15, 16])
def test_basket_per_path_and_sums(mc, eng, po, X, n_assets):
    b = basket_inputs(mc, n_assets, X)
    n = 6000
    first = 7 if n_assets != 4 else (1 << 32) - 3000   # one case straddles the 2^32 unit boundary
    got = f64(eng.basket_paths(b, n, SEED, first, X))
    want, o = po.dev_basket(X, b, SEED, first, n)
    assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * 100.0 * 2
    e = eng.basket(b, n, SEED, first, X)
    assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and e.sum2 == pytest.approx(o["sum2"], rel=TOL[X]["rel"])
    assert e.expected == pytest.approx(o["expected"], rel=TOL[X]["rel"])
    assert e.confidence == pytest.approx(o["confidence"], rel=TOL[X]["rel"])


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_basket_reference_n3_inputs(mc, eng, po, X):
    """The reference driver's own 3-asset data incl. its singular correlation matrix (basketOpt.cu:34-61):
    the factor has a zero last column (zero-pivot rule), weights 1/3, drift 0 plus a non-zero drift variant."""
    L, bad = mc.chol([[1, -.5, -.5], [-.5, 1, -.5], [-.5, -.5, 1]], X)
    assert bad == 1
    for d in ([0.0, 0.0, 0.0], [0.01, -0.02, 0.03]):
        b = dict(s=[100.0] * 3, v=[0.2, 0.3, 0.2], p=L.tolist(), d=d, w=[1 / 3] * 3, k=100.0, t=1.0, r=0.048790164)
        got = f64(eng.basket_paths(b, 5000, SEED, 0, X))
        want, o = po.dev_basket(X, b, SEED, 0, 5000)
        assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * 200.0


def test_basket_collapses_to_black_scholes(mc, eng):
    """n identical, perfectly correlated assets are one asset: the basket price must agree with
    closed-form Black-Scholes within the CI (SURVEY 8c hop C)."""
    n = 4
    L = np.zeros((n, n))
    L[:, 0] = 1.0   # Cholesky factor of the all-ones correlation matrix
    b = dict(s=[100.0] * n, v=[0.2] * n, p=L.tolist(), d=[0.0] * n, w=[0.25] * n, k=100.0, t=1.0, r=0.048790)
    for X in ("f32", "f64"):
        e = eng.basket(b, 4 * 10 ** 7, SEED, 0, X)
        assert abs(e.expected - BS_EXACT) < 3.5 / 1.96 * e.confidence


def test_basket_c3_c4_full_size_properties(mc, eng, po):
    """BASELINE configs[2] (n=4, 1e8 paths, f32) and configs[3] (n=16, 1e9 paths, f64) at full
    size on one GPU: shard additivity, f32-vs-f64 statistical agreement, and agreement with the
    reference CPU path (sp host at 1e6 paths, seed 12345: tests/golden/ref_mc.json ~10.33 / 9.71)."""
    b4 = basket_inputs(mc, 4, "f32")
    whole = eng.basket(b4, 10 ** 8, SEED, 0, "f32")
    s = sum(eng.basket(b4, cnt, SEED, first, "f32").sum for first, cnt in (mc.shard_range(10 ** 8, r, 8) for r in range(8)))
    assert s == pytest.approx(whole.sum, rel=2e-9)   # f32: which 16 payoffs share an fp32 partial depends on the split
    d4 = eng.basket(basket_inputs(mc, 4, "f64"), 10 ** 8, SEED + 9, 0, "f64")
    assert abs(whole.expected - d4.expected) < 4 / 1.96 * math.hypot(whole.confidence, d4.confidence)
    host4 = po.host_basket("f32", b4, 200000, 12345)   # reference CPU algorithm (sp formula), own stream
    assert abs(whole.expected - host4["expected"]) < 4 / 1.96 * host4["confidence"]
    b16 = basket_inputs(mc, 16, "f64")
    whole16 = eng.basket(b16, 10 ** 9, SEED, 0, "f64")
    assert whole16.n == 10 ** 9
    s = sum(eng.basket(b16, cnt, SEED, first, "f64").sum for first, cnt in (mc.shard_range(10 ** 9, r, 8) for r in range(8)))
    assert s == pytest.approx(whole16.sum, rel=1e-12)
    host16 = po.host_basket("f32", basket_inputs(mc, 16, "f32"), 100000, 12345)
    assert abs(whole16.expected - host16["expected"]) < 4 / 1.96 * host16["confidence"]


# ---- CVA ----------------------------------------------------------------------------------
def test_cva_last_date_a_hair_before_maturity_stays_finite(eng, po):
    """75 dates in fp64 leave a residual maturity of 1.4e-15 on the last date: d1 ~ 1e7 there and
    exp(-d1^2/2) has an argument of -1e14.  Every path must come out finite and equal to the oracle's
    (a table-driven exp whose rounding trick is only valid for |x| < 2e7 once turned half of them into NaN)."""
    c = dict(CVA0, n_grid=75)
    got = eng.cva_paths(c, 200000, SEED, 0, "f64")
    assert np.isfinite(got).all()
    want, _ = po.dev_cva("f64", c, SEED, 0, 20000)
    assert np.abs(got[:20000] - f64(want)).max() <= TOL["f64"]["cva"]
    assert np.isfinite(eng.cva_paths(c, 200000, SEED, 0, "f32")).all()


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_grid", [1, 2, 3, 25, 50, 75, 250, 256, 500])
def test_cva_per_path_and_sums(eng, po, X, n_grid):
    """Grid sizes of the reference driver (cvaOpt.cu:70-75) plus 256 (BASELINE C5) and tiny grids.
    250 in f64 ends with a NEGATIVE residual maturity (last date contributes 0), 256 with exactly 0
    (intrinsic value), 500 in f32 with a small positive one."""
    c = dict(CVA0, n_grid=n_grid)
    n = 3000
    got = f64(eng.cva_paths(c, n, SEED, 11, X))
    want, o = po.dev_cva(X, c, SEED, 11, n)
    assert np.abs(got - f64(want)).max() <= TOL[X]["cva"]
    e = eng.cva(c, n, SEED, 11, X)
    assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and e.sum2 == pytest.approx(o["sum2"], rel=2 * TOL[X]["rel"])
    assert e.expected == pytest.approx(o["expected"], rel=TOL[X]["rel"])
    assert e.confidence == pytest.approx(o["confidence"], rel=10 * TOL[X]["rel"])


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_cva_other_inputs(eng, po, X):
    c = dict(s=90.0, k=100.0, r=0.01, v=0.4, t=2.0, defint=0.1, lgd=0.45, n_grid=37)
    got = f64(eng.cva_paths(c, 4000, 5, 0, X))
    want, _ = po.dev_cva(X, c, 5, 0, 4000)
    assert np.abs(got - f64(want)).max() <= 4 * TOL[X]["cva"]


def cva_analytic(c):
    """E[C(S_t, T-t)] = C_0 e^{rt} under the pricing measure, so
    E[CVA] = LGD * sum_j dp_j * C_0 * e^{r t_j}   (SURVEY 8d C5), with C_0 the Black-Scholes value."""
    from math import erf, exp, log, sqrt
    s, k, r, v, t = (c[x] for x in "skrvt")
    d1 = (log(s / k) + (r + 0.5 * v * v) * t) / (v * sqrt(t))
    d2 = d1 - v * sqrt(t)
    phi = lambda x: 0.5 * (1 + erf(x / sqrt(2)))  # noqa: E731
    c0 = s * phi(d1) - k * exp(-r * t) * phi(d2)
    dt = t / c["n_grid"]
    return c["lgd"] * sum((exp(-c["defint"] * dt * (j - 1)) - exp(-c["defint"] * dt * j)) * c0 * exp(r * dt * j)
                          for j in range(1, c["n_grid"] + 1))


def test_cva_c5_full_size_vs_analytic(mc, eng):
    """BASELINE configs[4]: 256 dates x 1e7 paths, fp64, full size on one GPU: within 3.5 CI of the
    analytic target (the Hastings CDF's 1e-5 price error is far below the CI), shards add up."""
    c = dict(CVA0, n_grid=256)
    whole = eng.cva(c, 10 ** 7, SEED, 0, "f64")
    target = cva_analytic(c)
    assert abs(whole.expected - target) < 3.5 / 1.96 * whole.confidence + 2e-6
    s = sum(eng.cva(c, cnt, SEED, first, "f64").sum for first, cnt in (mc.shard_range(10 ** 7, r, 8) for r in range(8)))
    assert s == pytest.approx(whole.sum, rel=1e-12)
    f = eng.cva(c, 10 ** 7, SEED, 0, "f32")
    assert abs(f.expected - target) < 3.5 / 1.96 * f.confidence + 2e-6
    assert f.expected == pytest.approx(whole.expected, abs=4 / 1.96 * whole.confidence)


# ---- the legacy entry points and the error contract ---------------------------------------------
@pytest.mark.parametrize("X", ["f64", "f32"])
def test_legacy_symbols_drop_in(mc, eng, po, X):
    """dev_vanillaOpt / dev_basketOpt / dev_cvaEquityOption with the reference's structs and the
    reference's path-count rule numBlocks * (sims / numBlocks) (MonteCarloKernel.cu:491,508,524)."""
    L = C.CDLL(mc._lib.LEGACY[X])
    OptionData, MultiOptionData, OptionValue, CVA = po.ref_types(X, 3)
    L.dev_vanillaOpt.argtypes = [C.POINTER(OptionData), C.c_int, C.c_int, C.c_int]
    L.dev_vanillaOpt.restype = OptionValue
    L.dev_basketOpt.argtypes = [C.POINTER(MultiOptionData), C.c_int, C.c_int, C.c_int]
    L.dev_basketOpt.restype = OptionValue
    L.dev_cvaEquityOption.argtypes = [C.POINTER(CVA), C.c_int, C.c_int, C.c_int]
    L.dev_cvaEquityOption.restype = OptionValue
    R = np.dtype(po.NP[X]).type
    sims, blocks = 131072 * 3 + 77, 512
    paths = blocks * (sims // blocks)
    o = OptionData(*[VAN[k] for k in "skrvt"])
    v = L.dev_vanillaOpt(C.byref(o), blocks, 128, sims)
    e = eng.vanilla(VAN, paths, SEED, 0, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))
    b = basket_inputs(mc, 3, X)
    m = MultiOptionData()
    for i in range(3):
        m.s[i], m.v[i], m.d[i], m.w[i] = b["s"][i], b["v"][i], b["d"][i], b["w"][i]
        for j in range(3):
            m.p[i][j] = b["p"][i][j]
    m.k, m.t, m.r = b["k"], b["t"], b["r"]
    v = L.dev_basketOpt(C.byref(m), blocks, 128, sims)
    e = eng.basket(b, paths, SEED, 0, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))
    c = dict(CVA0, n_grid=50)
    s = CVA(c["defint"], c["lgd"], 0, OptionData(*[c[k] for k in "skrvt"]), c["n_grid"])
    v = L.dev_cvaEquityOption(C.byref(s), 1024, 256, 131072)
    e = eng.cva(c, 131072, SEED, 0, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_legacy_symbols_with_an_explicit_seed(mc, eng, po, X):
    """dev_*_ex(..., seed): SURVEY 8b -- the reference's API has no seed parameter; these variants take one."""
    L = C.CDLL(mc._lib.LEGACY[X])
    OptionData, MultiOptionData, OptionValue, CVA = po.ref_types(X, 3)
    L.dev_vanillaOpt_ex.argtypes = [C.POINTER(OptionData), C.c_int, C.c_int, C.c_int, C.c_uint64]
    L.dev_vanillaOpt_ex.restype = OptionValue
    L.dev_cvaEquityOption_ex.argtypes = [C.POINTER(CVA), C.c_int, C.c_int, C.c_int, C.c_uint64]
    L.dev_cvaEquityOption_ex.restype = OptionValue
    L.dev_vanillaOpt.argtypes = [C.POINTER(OptionData), C.c_int, C.c_int, C.c_int]
    L.dev_vanillaOpt.restype = OptionValue
    R = np.dtype(po.NP[X]).type
    o = OptionData(*[VAN[k] for k in "skrvt"])
    for seed in (1, 777, 2 ** 64 - 1):
        v = L.dev_vanillaOpt_ex(C.byref(o), 512, 128, 131072, seed)
        e = eng.vanilla(VAN, 131072, seed, 0, X)
        assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))
    d = L.dev_vanillaOpt(C.byref(o), 512, 128, 131072)     # the default seed again afterwards
    e = eng.vanilla(VAN, 131072, SEED, 0, X)
    assert (d.Expected, d.Confidence) == (R(e.expected), R(e.confidence))
    c = dict(CVA0, n_grid=50)
    s = CVA(c["defint"], c["lgd"], 0, OptionData(*[c[k] for k in "skrvt"]), c["n_grid"])
    v = L.dev_cvaEquityOption_ex(C.byref(s), 64, 256, 8192, 99)
    e = eng.cva(c, 8192, 99, 0, X)
    assert (v.Expected, v.Confidence) == (R(e.expected), R(e.confidence))


def test_python_mirror_of_reference_interface(mc, eng):
    v = mc.dev_vanillaOpt(mc.OptionData(**VAN), 512, 128, 131072 * 8)
    e = eng.vanilla(VAN, 131072 * 8, SEED, 0, "f64")
    assert (v.Expected, v.Confidence) == (e.expected, e.confidence)


def test_invalid_arguments_are_errors_not_crashes(mc, eng):
    with pytest.raises(mc.McError):
        eng.vanilla(VAN, 0)
    with pytest.raises(mc.McError):
        eng.vanilla(dict(VAN, s=-1.0), 10)
    with pytest.raises(mc.McError):
        eng.cva(dict(CVA0, n_grid=0), 10)
    n = 65
    with pytest.raises(mc.McError, match="1..64"):
        eng.basket(dict(s=[1.0] * n, v=[.1] * n, p=np.eye(n).tolist(), d=[0.0] * n, w=[1 / n] * n, k=1.0, t=1.0, r=0.0), 10)
    with pytest.raises(mc.McError):
        mc.Engine(99)
    # models whose exponent would leave the range of a double are refused, not priced as garbage
    with pytest.raises(mc.McError, match="range of a double"):
        eng.vanilla(dict(VAN, v=60.0, t=4.0), 10, SEED, 0, "f64")
    for n in (3, 12, 40):
        wild = dict(s=[100.0] * n, v=[90.0] * n, p=np.eye(n).tolist(), d=[0.0] * n, w=[1 / n] * n, k=100.0, t=1.0, r=0.0)
        with pytest.raises(mc.McError, match="range of a double"):
            eng.basket(wild, 10, SEED, 0, "f64")
    with pytest.raises(mc.McError, match="range of a double"):
        eng.cva(dict(CVA0, v=3.0e5, n_grid=50), 10, SEED, 0, "f64")


# ---- randomized inputs ----------------------------------------------------------------------------
def _rand_inputs(seed):
    rng = np.random.default_rng(seed)
    van = dict(s=float(rng.uniform(5, 500)), k=0.0, r=float(rng.uniform(-0.01, 0.12)), v=float(rng.uniform(0.02, 0.9)),
               t=float(rng.uniform(0.05, 5.0)))
    van["k"] = van["s"] * float(rng.uniform(0.6, 1.5))
    n = int(rng.integers(1, 17))
    a = rng.standard_normal((n, n + 2))
    corr = a @ a.T
    d = np.sqrt(np.diag(corr))
    corr = corr / d[:, None] / d[None, :]
    w = rng.uniform(0.0, 1.0, n)
    w /= w.sum()
    bsk = dict(s=rng.uniform(20, 200, n).tolist(), v=rng.uniform(0.05, 0.6, n).tolist(), corr=corr, d=rng.uniform(-0.05, 0.05, n).tolist(),
               w=w.tolist(), k=float(rng.uniform(60, 140)), t=float(rng.uniform(0.1, 3.0)), r=float(rng.uniform(0.0, 0.08)))
    cva = dict(s=float(rng.uniform(50, 150)), k=float(rng.uniform(60, 140)), r=float(rng.uniform(0.0, 0.08)), v=float(rng.uniform(0.1, 0.6)),
               t=float(rng.uniform(0.25, 3.0)), defint=float(rng.uniform(0.0, 0.2)), lgd=float(rng.uniform(0.1, 1.0)),
               n_grid=int(rng.integers(1, 200)))
    return van, bsk, cva


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("case", range(12))
def test_random_inputs_match_oracle(mc, eng, po, X, case):
    """Random market data, seeds and unaligned path ranges: per-path parity for all three products.
    Tolerances scale with the size of the values in play (spot / basket level)."""
    van, bsk, cva = _rand_inputs(1000 + case)
    seed = 0xC0FFEE00 + case * 7919
    first = (case * 1234567) % 1000003
    # vanilla
    n = 4001 + case
    got = f64(eng.vanilla_paths(van, n, seed, first, X))
    want, o = po.dev_vanilla(X, van, seed, first, n)
    level = van["s"] * math.exp(abs(van["r"]) * van["t"] + 4.5 * van["v"] * math.sqrt(van["t"]))
    assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * level * 2
    e = eng.vanilla(van, n, seed, first, X)
    assert e.sum == pytest.approx(o["sum"], rel=4 * TOL[X]["rel"], abs=TOL[X]["pay"] * level)
    # basket (Cholesky by the engine's own mc_chol, compared with the oracle's below)
    L, bad = mc.chol(bsk["corr"], X)
    assert bad == 0 and (L == po.chol(X, bsk["corr"])).all()
    b = dict(bsk, p=L.tolist())
    got = f64(eng.basket_paths(b, 2001, seed, first, X))
    want, o = po.dev_basket(X, b, seed, first, 2001)
    level = sum(wi * si * math.exp(4.5 * vi * math.sqrt(b["t"])) for wi, si, vi in zip(b["w"], b["s"], b["v"]))
    assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * level * 4
    # CVA
    got = f64(eng.cva_paths(cva, 1001, seed, first, X))
    want, o = po.dev_cva(X, cva, seed, first, 1001)
    scale = cva["lgd"] * cva["s"] / 100.0 * max(1.0, cva["defint"] * cva["t"] * 10)
    assert np.abs(got - f64(want)).max() <= TOL[X]["cva"] * max(scale, 0.05) * 4


def test_profile_hooks_report_kernel_time(eng):
    eng.profile(2)
    for i in range(6):
        eng.vanilla(VAN, 10 ** 7, SEED, i * 10 ** 7, "f32")
    samples, total_ms = eng.profile_read()
    eng.profile(0)
    assert samples == 3 and 0.003 < total_ms / samples < 0.2   # a 1e7-path launch takes ~10 us
    assert eng.profile_read() == (0, 0.0)


# ---- antithetic variates (an estimator the reference does not have) -------------------
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_antithetic_estimator_matches_oracle_and_reduces_variance(mc, po, X):
    """Per-sample parity with the oracle's antithetic twin (same tolerances as the plain estimator),
    unbiasedness against the closed forms, and the point of it: a smaller confidence interval."""
    cva = dict(CVA0, n_grid=64)   # dt = 1/64 is exact: all 64 dates are live (50 would lose its last date in f64)
    with mc.Engine(0) as e:
        e.set_antithetic(True)
        # vanilla: per sample, sums, and an unaligned range
        n, first = 20003, 5
        got = f64(e.vanilla_paths(VAN, n, SEED, first, X))
        want, o = po.dev_vanilla(X, VAN, SEED, first, n, antithetic=True)
        assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * VAN["s"]
        est = e.vanilla(VAN, n, SEED, first, X)
        assert est.n == n and est.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])
        assert est.confidence == pytest.approx(o["confidence"], rel=4 * TOL[X]["rel"])
        # baskets (f32 goes through the packed two-path kernel)
        for n_assets in (1, 3, 4, 11, 16):
            b = basket_inputs(mc, n_assets, X)
            got = f64(e.basket_paths(b, 4001, SEED, 7, X))
            want, o = po.dev_basket(X, b, SEED, 7, 4001, antithetic=True)
            assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * 200.0
            assert e.basket(b, 4001, SEED, 7, X).sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])
        # CVA, incl. a grid that ends on the intrinsic-value date
        for n_grid in (50, 256):
            c = dict(CVA0, n_grid=n_grid)
            got = f64(e.cva_paths(c, 2001, SEED, 11, X))
            want, o = po.dev_cva(X, c, SEED, 11, 2001, antithetic=True)
            assert np.abs(got - f64(want)).max() <= TOL[X]["cva"]
            assert e.cva(c, 2001, SEED, 11, X).sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])
        # unbiased, and tighter than plain Monte Carlo on the same number of normals
        anti = e.vanilla(VAN, 4 * 10 ** 7, SEED, 0, X)
        e.set_antithetic(False)
        plain = e.vanilla(VAN, 4 * 10 ** 7, SEED, 0, X)
        assert abs(anti.expected - BS_EXACT) < 3.5 / 1.96 * anti.confidence
        assert anti.confidence < 0.55 * plain.confidence     # variance down by > 3.3x for the at-the-money call
        e.set_antithetic(True)
        anti_cva = e.cva(cva, 10 ** 6, SEED, 0, X)
        e.set_antithetic(False)
        plain_cva = e.cva(cva, 10 ** 6, SEED, 0, X)
        assert abs(anti_cva.expected - cva_analytic(cva)) < 3.5 / 1.96 * anti_cva.confidence + 2e-6
        assert anti_cva.confidence < 0.5 * plain_cva.confidence


def test_async_launches_are_graph_capturable(mc):
    """The *_launch_* entry points do no allocation or synchronisation, so a caller may capture a batch
    of pricing calls into a hipGraph (torch.cuda.CUDAGraph is only the capture plumbing here)."""
    torch = pytest.importorskip("torch")
    eng = mc.Engine(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = torch.zeros((6, 3), dtype=torch.float64, device="cuda")
        prepared = [eng.prepared("vanilla", "f32", VAN)[0], eng.prepared("basket", "f64", basket_inputs(mc, 4, "f64")),
                    eng.prepared("cva", "f64", dict(CVA0, n_grid=64))[0]]
        calls = [("vanilla", "f32", prepared[0]), ("basket", "f64", prepared[1][0]), ("cva", "f64", prepared[2])]

        def enqueue():
            for i in range(6):
                prod, X, struct = calls[i % 3]
                eng.launch(prod, X, struct, SEED, i * 100003, 200001, out[i].data_ptr(), torch.cuda.current_stream().cuda_stream)
        enqueue()
        torch.cuda.synchronize()
        eager = out.clone()
        out.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            enqueue()
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert bool((out == eager).all())
    eng.close()


def test_largest_single_call_and_the_refusal_beyond(mc, eng):
    """One call covers at most 8 segments of 2^31 units (include/mc_mi355x.h "Sizes"): 2^36 fp32 vanilla paths in one call
    (6.9e10, ~35 ms) price to Black-Scholes within the confidence interval and add up from two halves; one unit more is
    refused with a message that says what to do."""
    n = 1 << 36
    e = eng.vanilla(VAN, n, SEED, 0, "f32")
    assert e.n == n
    assert abs(e.expected - BS_EXACT) < 3.5 / 1.96 * e.confidence + 2e-5      # + the fp32 constants' bias
    lo = eng.vanilla(VAN, n // 2, SEED, 0, "f32")
    hi = eng.vanilla(VAN, n // 2, SEED, n // 2, "f32")
    assert lo.sum + hi.sum == pytest.approx(e.sum, rel=1e-13) and lo.n + hi.n == n
    with pytest.raises(mc.McError, match="split it"):
        eng.vanilla(VAN, n + 4, SEED, 0, "f32")
    with pytest.raises(mc.McError, match="split it"):
        eng.cva(dict(CVA0, n_grid=4), (1 << 34) + 1, SEED, 0, "f64")


def test_one_context_per_host_thread_is_safe(mc):
    """one context per host thread.  Four threads, each with its own context, price a mix of calls
    concurrently (ctypes releases the GIL inside a call); every result equals the serial one bit for bit, and an error raised
    in one thread carries that thread's own message (mc_last_error is thread-local)."""
    import threading
    jobs = []
    for i in range(24):
        prod = ("vanilla", "basket", "cva")[i % 3]
        X = ("f32", "f64")[(i // 3) % 2]
        inp = {"vanilla": VAN, "basket": basket_inputs(mc, (3, 4, 16, 20)[i % 4], X), "cva": dict(CVA0, n_grid=(12, 50)[i % 2])}[prod]
        jobs.append((prod, X, inp, 20000 + 997 * i, SEED + i, 1000003 * i))
    with mc.Engine(0) as e:
        serial = [getattr(e, p)(inp, n, seed, first, X) for p, X, inp, n, seed, first in jobs]
    results, errors = {}, {}

    def worker(t):
        try:
            with mc.Engine(0) as e:
                for rep in range(6):
                    for j in range(t, len(jobs), 4):
                        p, X, inp, n, seed, first = jobs[j]
                        r = getattr(e, p)(inp, n, seed, first, X)
                        results[(t, rep, j)] = (r.sum, r.sum2, r.n)
                try:
                    e.vanilla(dict(VAN, s=-1.0 - t), 10)
                except mc.McError as ex:
                    errors[t] = str(ex)
        except Exception as ex:      # noqa: BLE001
            errors[t] = "thread failed: " + repr(ex)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert len(results) == 6 * len(jobs)
    for (t, rep, j), got in results.items():
        assert got == (serial[j].sum, serial[j].sum2, serial[j].n), (t, rep, j)
    assert sorted(errors) == [0, 1, 2, 3] and all("s>0" in m or "need" in m for m in errors.values()), errors


def test_armed_launch_and_publish_deliver_into_pinned_slots(mc):
    """mc_context_arm_direct / mc_context_publish (what libmc_multi and bench.py's strong rows read results back with):
    the next asynchronous launch's last workgroup stores the triple into pinned host memory, a one-lane kernel does the
    same for any device triple; the host polls the n word.  Bits equal the synchronous call's; the sentinel is reset per
    call; a synchronous call in between cancels a pending arming."""
    torch = pytest.importorskip("torch")
    eng = mc.Engine(0)
    out = torch.zeros(3, dtype=torch.float64, device="cuda")
    for prod, X, inputs, n in (("vanilla", "f32", VAN, 10 ** 6 + 3), ("basket", "f64", basket_inputs(mc, 16, "f64"), 30001),
                               ("cva", "f64", dict(CVA0, n_grid=64), 5000)):
        ref = getattr(eng, prod)(inputs, n, SEED, 7, X)
        struct, keep = eng.prepared(prod, X, inputs)
        for rep in range(40):
            slot = eng.arm_direct()
            assert slot[2] == -1.0
            eng.launch(prod, X, struct, SEED, 7, n, out.data_ptr(), eng.stream)
            s, q, cnt = eng.wait_slot(slot)
            assert (s, q, cnt) == (ref.sum, ref.sum2, float(n))
        torch.cuda.synchronize()
        assert out.tolist() == [ref.sum, ref.sum2, float(n)]          # d_triple is written as well
        pub = eng.publish(out.data_ptr(), eng.stream)
        assert eng.wait_slot(pub) == (ref.sum, ref.sum2, float(n))
    # an arming followed by a synchronous call is cancelled: the later launch must not touch the slot
    slot = eng.arm_direct()
    eng.vanilla(VAN, 4096, SEED, 0, "f32")
    struct, keep = eng.prepared("vanilla", "f32", VAN)
    eng.launch("vanilla", "f32", struct, SEED, 0, 4096, out.data_ptr(), eng.stream)
    torch.cuda.synchronize()
    assert slot[2] == -1.0
    eng.set_finish(False)     # the two-launch form has no last arriver to do it
    with pytest.raises(mc.McError, match="fused"):
        eng.arm_direct()
    eng.close()


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_assets", [17, 24, 30, 33, 64])
def test_generic_basket_beyond_compiled_sizes(mc, eng, po, X, n_assets):
    """n > 16 runs the LDS-staged generic kernel (the reference's N is any compile-time constant):
    same stream, same estimator, same tolerances as the specialised kernels; also antithetic."""
    b = basket_inputs(mc, n_assets, X, rho=0.3)
    n, first = 3001, (1 << 32) - 1500          # straddles the 2^32 unit boundary: two segments
    got = f64(eng.basket_paths(b, n, SEED, first, X))
    want, o = po.dev_basket(X, b, SEED, first, n)
    assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * 100.0 * 4
    e = eng.basket(b, n, SEED, first, X)
    assert e.sum == pytest.approx(o["sum"], rel=2 * TOL[X]["rel"]) and e.confidence == pytest.approx(o["confidence"], rel=4 * TOL[X]["rel"])
    with mc.Engine(0) as anti:
        anti.set_antithetic(True)
        got = f64(anti.basket_paths(b, 1001, SEED, 3, X))
        want, _ = po.dev_basket(X, b, SEED, 3, 1001, antithetic=True)
        assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * 100.0 * 4


@pytest.mark.parametrize("n_assets", [3, 8, 9, 13, 16])
def test_basket_kernel_families_agree_bitwise_in_f64(mc, eng, n_assets):
    """The three fp64 basket kernels -- constants as kernel arguments / LDS (basket_kernel), scalar-loaded tiles
    with normals in registers (basket_tiled_kernel, the default for 9..16 assets), generic tiled with normals in
    LDS (basket_dyn_kernel) -- run the same fma chains in the same order (padding adds exact zeros), so their
    per-path payoffs are identical bits.  The families are selected through MC_BASKET_STATIC_MAX_F64 /
    MC_BASKET_TILED_MIN, which are read once per process: hence the child processes."""
    import json
    import subprocess
    import sys
    import tempfile
    b = basket_inputs(mc, n_assets, "f64", rho=0.4)
    default = eng.basket_paths(b, 5000, SEED, 77, "f64")
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "import montecarlocuda_amd as mc\n"
            "b = json.load(open(sys.argv[1]))\n"
            "with mc.Engine(0) as e:\n"
            "    np.save(sys.argv[2], e.basket_paths(b, 5000, %d, 77, 'f64'))\n" % (ROOT, SEED))
    families = {"arguments": dict(MC_BASKET_STATIC_MAX_F64="16"),
                "tiled": dict(MC_BASKET_STATIC_MAX_F64="8", MC_BASKET_TILED_MIN="9"),
                "generic": dict(MC_BASKET_STATIC_MAX_F64="0", MC_BASKET_TILED_MIN="1000")}
    with tempfile.TemporaryDirectory() as d:
        json.dump(b, open(os.path.join(d, "b.json"), "w"))
        for name, env in families.items():
            subprocess.run([sys.executable, "-c", code, os.path.join(d, "b.json"), os.path.join(d, name + ".npy")], check=True,
                           env=dict(os.environ, **env), timeout=300)
            got = np.load(os.path.join(d, name + ".npy"))
            assert got.dtype == np.float64 and np.array_equal(got, default), name


@pytest.mark.parametrize("n_assets,anti,cv", [(16, 0, 0), (16, 1, 1), (13, 0, 1), (14, 1, 0)])
def test_basket_matrix_core_family_matches_default_and_oracle(mc, po, n_assets, anti, cv):
    """basket_mfma_f64_kernel (MC_BASKET_MFMA=1; off by default because it measured 5-6 % slower): the
    Cholesky step of 64 paths as 16 v_mfma_f64_16x16x4_f64, every lane generating the normals the B operand wants
    from it, the basket sum closed across lane groups with v_permlane32_swap / v_permlane16_swap.  Same units, same
    stream, same per-lane sums as the other families; only the order of the additions inside x and inside the basket
    differs, so per-path payoffs agree to a few ulp (not bit for bit), and with the oracle within the fp64 tolerance."""
    import json
    import subprocess
    import sys
    import tempfile
    n = 5003   # not a multiple of 64: the last wave prices paths beyond the range and drops them
    b = basket_inputs(mc, n_assets, "f64", rho=0.4)
    with mc.Engine(0) as e:
        e.set_antithetic(bool(anti)), e.set_control_variate(bool(cv))
        default = e.basket_paths(b, n, SEED, 77, "f64")
        ref = e.basket(b, 200001, SEED, 5, "f64")
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "import montecarlocuda_amd as mc\n"
            "b = json.load(open(sys.argv[1]))\n"
            "with mc.Engine(0) as e:\n"
            "    e.set_antithetic(%d); e.set_control_variate(%d)\n"
            "    np.save(sys.argv[2], e.basket_paths(b, %d, %d, 77, 'f64'))\n"
            "    r = e.basket(b, 200001, %d, 5, 'f64')\n"
            "    print(json.dumps({'sum': r.sum, 'sum2': r.sum2, 'n': r.n}))\n" % (ROOT, anti, cv, n, SEED, SEED))
    with tempfile.TemporaryDirectory() as d:
        json.dump(b, open(os.path.join(d, "b.json"), "w"))
        out = subprocess.run([sys.executable, "-c", code, os.path.join(d, "b.json"), os.path.join(d, "m.npy")], check=True,
                             env=dict(os.environ, MC_BASKET_MFMA="1"), timeout=300, capture_output=True, text=True)
        got = np.load(os.path.join(d, "m.npy"))
        sums = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert got.shape == default.shape and got.dtype == np.float64
    assert not np.array_equal(got, default), "identical bits: the matrix-core family did not run"
    assert np.max(np.abs(got - default) / np.maximum(np.abs(default), 1.0)) < 2e-12
    want, _ = po.dev_basket("f64", b, SEED, 77, n, antithetic=bool(anti), control=bool(cv))
    assert np.abs(got - f64(want)).max() <= TOL["f64"]["pay"] * 100.0 * 4
    assert sums["n"] == 200001
    assert sums["sum"] == pytest.approx(ref.sum, rel=1e-12) and sums["sum2"] == pytest.approx(ref.sum2, rel=1e-12)


def test_generic_basket_alternates_with_cva_on_one_context(eng, mc, po):
    """The generic basket's constants and the CVA date table share the context's table buffer: calls of
    both kinds interleaved must each see their own data."""
    b = basket_inputs(mc, 20, "f64", rho=0.3)
    c = dict(CVA0, n_grid=64)
    _, ob = po.dev_basket("f64", b, SEED, 0, 2001)
    _, oc = po.dev_cva("f64", c, SEED, 0, 2001)
    for _ in range(3):
        assert eng.basket(b, 2001, SEED, 0, "f64").sum == pytest.approx(ob["sum"], rel=1e-12)
        assert eng.cva(c, 2001, SEED, 0, "f64").sum == pytest.approx(oc["sum"], rel=1e-12)


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_basket_control_variate_matches_oracle_and_reduces_variance(mc, po, X):
    """Geometric-basket control variate: per-path values payoff(arith) - payoff(geo) against
    the oracle twin (specialised, packed and generic kernels, with and without antithetic), the closed-form
    mean added back by mc_basket_run_*, and the variance actually falling by two orders of magnitude."""
    with mc.Engine(0) as e, mc.Engine(0) as plain:
        e.set_control_variate(True)
        for n_assets in (1, 3, 4, 11, 14, 16, 20):
            b = basket_inputs(mc, n_assets, X, rho=0.5)
            b["w"] = [(1.0 + 0.2 * (i % 3)) / n_assets for i in range(n_assets)]      # unequal weights, sum != 1
            for anti in (False, True):
                e.set_antithetic(anti)
                got = f64(e.basket_paths(b, 3001, SEED, 7, X))
                want, o = po.dev_basket(X, b, SEED, 7, 3001, antithetic=anti, control=True)
                assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * 100.0 * 4
                est = e.basket(b, 3001, SEED, 7, X)
                assert est.expected == pytest.approx(o["expected"], rel=max(TOL[X]["rel"], 1e-9))
                # the per-path values are differences of two nearly equal payoffs (exactly equal at n=1): their
                # sum is small and carries the per-path rounding of both sides
                assert est.sum == pytest.approx(o["sum"], rel=20 * TOL[X]["rel"], abs=3001 * TOL[X]["pay"] * 100 * 0.05)
        e.set_antithetic(False)
        b4 = basket_inputs(mc, 4, X)
        cv, mc_plain = e.basket(b4, 10 ** 7, SEED, 0, X), plain.basket(b4, 10 ** 7, SEED, 0, X)
        assert cv.confidence < 0.12 * mc_plain.confidence                     # variance down by > 70x
        assert abs(cv.expected - mc_plain.expected) < 3.5 / 1.96 * mc_plain.confidence
        with pytest.raises(mc.McError, match=r"w\[a\] > 0"):
            e.basket(dict(b4, w=[0.5, 0.5, 0.5, -0.5]), 1000, SEED, 0, X)


def test_legacy_env_switches_estimators(mc, po):
    """MC_ANTITHETIC / MC_CONTROL_VARIATE reach the legacy GPU symbols and the CPU twin alike: run the C
    basket driver with both set; its CPU and GPU legs must agree to rounding and beat the plain CI."""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "drivers", "basketOpt_f64")
    import re
    import subprocess
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)], stdout=subprocess.DEVNULL)

    def run(env):
        out = subprocess.run([exe, "8"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert out.returncode == 0, out.stderr
        nums = [float(x) for x in re.findall(r"^-?\d+\.\d+", out.stdout[out.stdout.index("-\tResults"):], flags=re.M)]
        return nums   # cpu price, cpu ci, cpu time, gpu price, gpu ci, diff, time, speedup
    plain = run({})
    # the reference's own N=3 data has a singular correlation matrix and a zero-vol-free setup: fine for CV
    both = run({"MC_ANTITHETIC": "1", "MC_CONTROL_VARIATE": "1"})
    assert abs(both[0] - both[3]) < 2e-6 and abs(both[1] - both[4]) < 2e-6        # CPU twin == GPU
    assert both[4] < 0.6 * plain[4]    # tighter CI (only ~2x here: the reference data anti-correlate the assets)
    assert abs(both[3] - plain[3]) < 3.5 / 1.96 * plain[4]                       # same price


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_vanilla_pathwise_greeks(eng, po, X):
    """Price, delta and vega in one pass: sums against the oracle twin on the same counters
    (unaligned range), the price leg equal to the pricing kernel's, and all three within their CI of the
    closed forms  N(d1)  and  S sqrt(T) phi(d1)  at 1e8 paths."""
    n, first = 50003, 7
    gp, gd, gv = eng.vanilla_greeks(VAN, n, SEED, first, X)
    op, od, ov = po.dev_vanilla_greeks(X, VAN, SEED, first, n)
    rel = 4 * TOL[X]["rel"]
    for g, o in ((gp, op), (gd, od), (gv, ov)):
        assert g.n == n and g.sum == pytest.approx(o["sum"], rel=rel) and g.sum2 == pytest.approx(o["sum2"], rel=rel)
        assert g.expected == pytest.approx(o["expected"], rel=rel) and g.confidence == pytest.approx(o["confidence"], rel=rel)
    assert gp.sum == pytest.approx(eng.vanilla(VAN, n, SEED, first, X).sum, rel=rel)
    big = eng.vanilla_greeks(VAN, 10 ** 8, SEED, 0, X)
    s, k, r, v, t = (VAN[c] for c in "skrvt")
    d1 = (math.log(s / k) + (r + 0.5 * v * v) * t) / (v * math.sqrt(t))
    exact = (BS_EXACT, 0.5 * math.erfc(-d1 / math.sqrt(2)), s * math.sqrt(t) * math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi))
    for g, want in zip(big, exact):
        assert abs(g.expected - want) < 3.5 / 1.96 * g.confidence, (g.expected, want)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_assets", [1, 4, 9, 12, 16])
@pytest.mark.parametrize("strike", [-1000.0, -3.5, 0.0])
def test_basket_zero_and_negative_strike(mc, eng, po, X, n_assets, strike):
    """The reference's max(basket - K, 0) (dp/MonteCarloKernel.cu:99-100) holds for any K.  The fp32 kernels
    of <= 12 assets fold the max into a [0,1] clamp after a power-of-two rescale: the scale must cover
    basket + |K| for K < 0 (with K = -1000 on spots of 100 every path is worth ~1100, not the scale)."""
    b = dict(basket_inputs(mc, n_assets, X), k=strike)
    for anti in (False, True):
        eng.set_antithetic(anti)
        try:
            got = f64(eng.basket_paths(b, 4001, SEED, 3, X))
            e = eng.basket(b, 4001, SEED, 3, X)
        finally:
            eng.set_antithetic(False)
        want, o = po.dev_basket(X, b, SEED, 3, 4001, antithetic=anti)
        level = 100.0 * 4 + abs(strike)
        assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * level * 2
        assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and e.sum2 == pytest.approx(o["sum2"], rel=2 * TOL[X]["rel"])
        assert got.min() >= -strike * 0.999   # every path is in the money by at least |K|


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("strike", [-250.0, -3.5, 0.0])
def test_vanilla_zero_and_negative_strike(mc, eng, po, X, strike):
    """The reference's max(S_T - K, 0) (dp/MonteCarloKernel.cu:70) holds for any K; round 2 refused K <= 0 in the vanilla
    entry points while the basket ones accepted it.  The fp32 hot kernel folds the max into a [0,1] clamp after a
    power-of-two rescale: |K| must be inside that scale for K < 0 (every path is then worth S_T + |K|).  Per path through
    the masked kernel, sums through the hot kernel (aligned range) and through the edge launches (unaligned)."""
    opt = dict(VAN, k=strike)
    for anti in (False, True):
        eng.set_antithetic(anti)
        try:
            got = f64(eng.vanilla_paths(opt, 4001, SEED, 3, X))
            e = eng.vanilla(opt, 4001, SEED, 3, X)
            hot = eng.vanilla(opt, 1 << 16, SEED, 0, X)
            g3 = eng.vanilla_greeks(opt, 4001, SEED, 3, X)[0] if not anti else None
        finally:
            eng.set_antithetic(False)
        want, o = po.dev_vanilla(X, opt, SEED, 3, 4001, antithetic=anti)
        _, oh = po.dev_vanilla(X, opt, SEED, 0, 1 << 16, want_paths=False, antithetic=anti)
        level = VAN["s"] * 3 + abs(strike)
        assert np.abs(got - f64(want)).max() <= TOL[X]["pay"] * level
        assert got.min() >= -strike          # S_T > 0: every path is in the money by more than |K|
        assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and e.sum2 == pytest.approx(o["sum2"], rel=2 * TOL[X]["rel"])
        assert hot.sum == pytest.approx(oh["sum"], rel=TOL[X]["rel"]) and hot.sum2 == pytest.approx(oh["sum2"], rel=2 * TOL[X]["rel"])
        assert hot.expected == pytest.approx(oh["expected"], rel=TOL[X]["rel"])
        if g3 is not None:
            assert g3.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"])
    # with K <= 0 the price is the forward minus the discounted strike: E = S - K exp(-rT), within the CI
    big = eng.vanilla(opt, 10 ** 7, SEED, 0, X)
    exact = VAN["s"] - strike * math.exp(-VAN["r"] * VAN["t"])
    assert abs(big.expected - exact) < 3.5 / 1.96 * big.confidence + 1e-5 * exact


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_greeks_refuse_out_of_range_models(mc, eng, X):
    """The Greeks entry point applies the pricing paths' exponent guards (a huge v sqrt(t) used to return
    inf / NaN sums with MC_OK)."""
    wild = dict(VAN, v=60.0, t=4.0)
    with pytest.raises(mc.McError, match="range of the simulation type"):
        eng.vanilla_greeks(wild, 1000, SEED, 0, X)
    g = eng.vanilla_greeks(dict(VAN, v=0.9, t=4.0), 1000, SEED, 0, X)
    assert all(math.isfinite(q.sum) and math.isfinite(q.sum2) for q in g)
