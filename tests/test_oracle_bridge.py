"""Hop A': the oracle's DEVICE-formula family (orc_dev_*, what the HIP kernels are compared with) tied to the
reference's own outputs.

Round 2's chain was  compiled MonteCarloHost.c ==bitwise== orc_host_*  and  HIP ==tolerance== orc_dev_*, with nothing
between orc_host_* and orc_dev_* but shared closed forms and statistics.  Here the device formulas
(MonteCarloKernel.cu:67-71 vanilla, :74-101 basket, :241-262 CVA, as restated in oracle/mc_oracle_impl.h) are evaluated
on the REFERENCE's normal stream -- glibc rand() + Box-Muller, MonteCarloHost.c:111-121, which orc_host_gaussians
reproduces (the bit-for-bit agreement of orc_host_* with the compiled reference pins it) -- and the per-path values are
pushed through the reference's own accumulation and closing (sequential sums in `real`, MonteCarloHost.c:196-228).
What comes out is compared with tests/golden/ref_mc.json (numbers printed by the unmodified MonteCarloHost.c) and, live,
with oracle/_ref.

STATED BOUNDS (per path, device formula vs the reference CPU formula on the same normal):
  vanilla f64      bit-equal for T = 1 (the golden's option: v sqrt(T) z == z sqrt(T) v when sqrt(T) == 1);
                   <= 4 ulp of S_T otherwise (one product associates differently: 1 ulp of an exponent of size ~2, then
                   the exponential's own rounding; measured 2.3)  -> (E, CI) == golden, bit for bit
  vanilla f32      the sp reference forms the exponent in DOUBLE ((r - 0.5 v v) T is a double expression) and rounds once;
                   the device twin rounds drift and diffusion to float first: <= 4 ulp of S_T (measured 2.8)
                   -> E within 2e-6 relative
  basket f32       bit-equal (same expression order; the reference multiplies the factor's structural zeros, which adds
                   exact zeros)  -> (E, CI) == golden, bit for bit, N = 3, 4, 16
  basket f64       the dp reference CPU path drops the volatility from the diffusion; with the oracle's
                   compat bit the device formulas reproduce it bit for bit -> (E, CI) == golden, N = 3, 4, 16; without
                   the bit (the real device formula) the result is checked against the sp golden's neighbourhood
  CVA              with the reference CPU loop's ordering (exposure at the lagged spot, a draw at every date) and its
                   dp_j arithmetic switched on, the device formulas differ from the reference CPU path only in how the
                   GBM step's diffusion is associated: per path <= 1e-12 relative (f64), 2e-4 relative (f32: 250-500
                   fp32 steps)  -> E within those bounds of the golden
"""
import math

import numpy as np
import pytest

from conftest import fromhex, load_golden

MC = [c for c in load_golden("ref_mc.json")["cases"]]
ULP = {"f32": 2.0 ** -23, "f64": 2.0 ** -52}


def _gaussians(po, X, seed, count):
    return po.host_gaussians(X, seed, count)


def _basket_inputs(c):
    return dict(c["basket"], p=[[fromhex(x) for x in row] for row in c["factor"]])


# ---- the taps are the pinned host family ----------------------------------------------------------------------------
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_host_taps_do_not_change_the_pinned_results(po, X):
    """orc_host_*_paths = orc_host_* with per-path values handed back: same (E, CI) bits, and the reference's
    accumulation applied to the tapped values gives them again."""
    opt = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
    vals, r = po.host_vanilla_paths(X, opt, 5000, 777)
    assert r == po.host_vanilla(X, opt, 5000, 777)
    again = po.ref_close(X, vals, 1, opt["r"], opt["t"])
    assert (again["expected"], again["confidence"]) == (r["expected"], r["confidence"])
    cva = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=25)
    vals, r = po.host_cva_paths(X, cva, 300, 5)
    assert r == po.host_cva(X, cva, 300, 5)
    again = po.ref_close(X, vals, 0, cva["r"], cva["t"])
    assert (again["expected"], again["confidence"]) == (r["expected"], r["confidence"])


# ---- vanilla -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in MC if c["kind"] == "vanilla" and c["paths"] <= 100000],
                         ids=lambda c: f"{c['X']}-{c['paths']}-{c['seed']}")
def test_vanilla_device_formula_on_reference_stream_vs_golden(po, c):
    X, opt, n = c["X"], c["opt"], c["paths"]
    z = _gaussians(po, X, c["seed"], n)
    dev, _ = po.dev_vanilla_on_normals(X, opt, z)
    ref, _ = po.host_vanilla_paths(X, opt, n, c["seed"])
    closed = po.ref_close(X, dev, 1, opt["r"], opt["t"])
    want_e, want_ci = fromhex(c["expected"]), fromhex(c["confidence"])
    if X == "f64":
        assert (dev == ref).all()
        assert (closed["expected"], closed["confidence"]) == (want_e, want_ci)
    else:
        s_t = ref.astype(np.float64) + opt["k"]
        assert (np.abs(dev.astype(np.float64) - ref) <= 4 * ULP[X] * np.maximum(s_t, opt["s"])).all()
        assert closed["expected"] == pytest.approx(want_e, rel=2e-6)
        assert closed["confidence"] == pytest.approx(want_ci, rel=2e-5)


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_vanilla_device_formula_general_maturity(po, X):
    """T != 1: sqrt(T) no longer drops out, the two formulas associate v sqrt(T) z differently: a few ulp of S_T."""
    opt = dict(s=105.0, k=95.0, r=0.02, v=0.35, t=2.5)
    n = 20000
    z = _gaussians(po, X, 99, n)
    dev, _ = po.dev_vanilla_on_normals(X, opt, z)
    ref, _ = po.host_vanilla_paths(X, opt, n, 99)
    s_t = ref.astype(np.float64) + opt["k"]
    assert (np.abs(dev.astype(np.float64) - ref) <= 4 * ULP[X] * np.maximum(s_t, opt["s"])).all()


# ---- basket ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in MC if c["kind"] == "basket"],
                         ids=lambda c: f"{c['X']}-n{c['n']}-{c['corr_name']}-{c['paths']}")
def test_basket_device_formulas_on_reference_stream_vs_golden(po, c):
    X, n, paths = c["X"], c["n"], c["paths"]
    b = _basket_inputs(c)
    g = _gaussians(po, X, c["seed"], paths * n).reshape(paths, n)
    # f32: the sp reference and the device agree on the model; f64: the dp reference CPU path drops the volatility
    # from the diffusion (MonteCarloHost.c:180): reproduced by the oracle's compat bit, never by the product
    mode = 0 if X == "f32" else po.BASKET_NO_VOL
    dev, _ = po.dev_basket_on_normals(X, b, g, mode)
    ref, _ = po.host_basket_paths(X, b, paths, c["seed"])
    assert (dev == ref).all()
    closed = po.ref_close(X, dev, 1, b["r"], b["t"])
    assert (closed["expected"], closed["confidence"]) == (fromhex(c["expected"]), fromhex(c["confidence"]))


@pytest.mark.parametrize("n", [3, 4, 16])
def test_basket_true_device_formula_f64_equals_sp_formula_in_double(po, n):
    """The REAL device formula in fp64 (volatility in the diffusion) has no dp golden to meet -- the dp CPU path is the
    buggy one.  It is the sp reference's formula carried out in double: on the same normals its per-path payoffs agree
    with the sp golden's formula (orc_host_basket_paths f32, bit-pinned above) to fp32 rounding."""
    c = next(c for c in MC if c["kind"] == "basket" and c["X"] == "f32" and c["n"] == n and c["corr_name"] == "equi0.5" and c["paths"] == 1000)
    b = _basket_inputs(c)
    g32 = _gaussians(po, "f32", c["seed"], 1000 * n).reshape(1000, n)
    ref32, _ = po.host_basket_paths("f32", b, 1000, c["seed"])
    dev64, _ = po.dev_basket_on_normals("f64", b, g32.astype(np.float64), 0)
    assert np.abs(dev64 - ref32).max() <= 3e-5 * 100.0


# ---- CVA ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in MC if c["kind"] == "cva"],
                         ids=lambda c: f"{c['X']}-{c['cva']['n_grid']}x{c['paths']}-{c['seed']}")
def test_cva_device_loop_in_host_order_on_reference_stream_vs_golden(po, c):
    X, cva, paths = c["X"], c["cva"], c["paths"]
    z = _gaussians(po, X, c["seed"], paths * cva["n_grid"]).reshape(paths, cva["n_grid"])
    flags = po.CVA_HOST_ORDER | po.CVA_REF_DP | po.CVA_REF_T0
    dev, _ = po.dev_cva_on_normals(X, cva, z, flags)
    ref, _ = po.host_cva_paths(X, cva, paths, c["seed"])
    rel = 1e-12 if X == "f64" else 2e-4
    scale = np.abs(ref.astype(np.float64)).max()
    assert np.abs(dev.astype(np.float64) - ref).max() <= rel * scale
    closed = po.ref_close(X, dev, 0, cva["r"], cva["t"])
    assert closed["expected"] == pytest.approx(fromhex(c["expected"]), rel=rel)
    assert closed["confidence"] == pytest.approx(fromhex(c["confidence"]), rel=10 * rel)


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_cva_switches_are_the_only_differences(po, X):
    """Turning the compat switches off one at a time moves the result by what each documented deviation is worth:
    dp_j via expm1 (rounding only in f64; 5e-4 relative in f32, where the reference's difference of two floats near 1
    loses four digits: SURVEY 2.3 #9), device ordering (exposure at the NEW spot: a different estimator of the same
    CVA, equal within the confidence interval)."""
    cva = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=250)
    paths = 4000
    z = _gaussians(po, X, 12345, paths * 250).reshape(paths, 250)
    full, rf = po.dev_cva_on_normals(X, cva, z, po.CVA_HOST_ORDER | po.CVA_REF_DP | po.CVA_REF_T0)
    no_dp, rn = po.dev_cva_on_normals(X, cva, z, po.CVA_HOST_ORDER | po.CVA_REF_T0)
    assert rn["expected"] == pytest.approx(rf["expected"], rel=1e-11 if X == "f64" else 2e-3)
    dev_order, rd = po.dev_cva_on_normals(X, cva, z, 0)
    assert abs(rd["expected"] - rf["expected"]) < 4 * (rd["confidence"] + rf["confidence"]) / 1.96


# ---- live against the compiled reference ------------------------------------------------------------------------------
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_bridge_live_against_compiled_reference(po, X):
    """Where oracle/_ref is present: other seeds and market data than the goldens hold."""
    if not po.ref_available(X, 3):
        pytest.skip("oracle/_ref not built here")
    ref = po.Ref(X, 3)
    rng = np.random.default_rng(7)
    for seed in (3, 2024):
        opt = dict(s=float(rng.uniform(50, 150)), k=float(rng.uniform(60, 140)), r=float(rng.uniform(0, 0.1)), v=float(rng.uniform(0.1, 0.5)), t=1.0)
        n = 3000
        dev, _ = po.dev_vanilla_on_normals(X, opt, _gaussians(po, X, seed, n))
        closed = po.ref_close(X, dev, 1, opt["r"], opt["t"])
        e, ci = ref.vanilla(opt, n, seed)
        if X == "f64":
            assert (closed["expected"], closed["confidence"]) == (e, ci)
        else:
            assert closed["expected"] == pytest.approx(e, rel=2e-6) and closed["confidence"] == pytest.approx(ci, rel=2e-5)
        corr = [[1.0, 0.3, 0.1], [0.3, 1.0, -0.2], [0.1, -0.2, 1.0]]
        L = ref.chol(corr)
        b = dict(s=[90.0, 100.0, 110.0], v=[0.25, 0.15, 0.35], p=L.tolist(), d=[0.0, 0.01, -0.01], w=[0.2, 0.5, 0.3], k=97.0, t=1.5, r=0.03)
        g = _gaussians(po, X, seed, n * 3).reshape(n, 3)
        dev, _ = po.dev_basket_on_normals(X, b, g, 0 if X == "f32" else po.BASKET_NO_VOL)
        closed = po.ref_close(X, dev, 1, b["r"], b["t"])
        assert (closed["expected"], closed["confidence"]) == ref.basket(b, n, seed)
