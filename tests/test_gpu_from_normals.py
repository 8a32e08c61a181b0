"""The direct hop: the REFERENCE's normal stream through the HIP kernels, compared with numbers the compiled
reference printed (tests/golden/ref_mc.json) -- north_star's "match MonteCarloHost.c on the same seeds", literally.

The test hooks mc_*_from_normals_* (include/mc_mi355x.h) run the pricing call's own simulation kernel for the size --
vanilla_f32_kernel / vanilla_kernel<f64>, basket_f32_kernel<3|4> / basket_kernel<f64, 3|4>, the tiled kernels at 16 assets,
cva_kernel -- instantiated with the external-normals generator policy (mc_rng.hpp: GenExternal): payoff, per-lane sums
and their fp32 flushes, DPP/LDS reduction and the last-arriver final reduction are the hot path's code; only the
Box-Muller step is replaced by a load.  The stream fed in is glibc rand() + Box-Muller (MonteCarloHost.c:111-121) as
the oracle's orc_host_gaussians reproduces it (pinned: orc_host_* equals the compiled reference bit for bit).

What is compared, and the STATED TOLERANCES:
  (1) per path, HIP vs the oracle's device formulas on the same normals (orc_dev_*_on_normals; the CPU test
      tests/test_oracle_bridge.py ties those to the goldens bit for bit / to a few ulp): the bounds of
      tests/test_gpu_parity.py -- f32: 2e-6 S per payoff, 2e-5 per CVA value; f64: 1e-14 S, 1e-13.
  (2) the HIP per-path values pushed through the REFERENCE's accumulation and closing (sequential sums in `real`,
      MonteCarloHost.c:196-228 = orc_ref_close) against the golden (Expected, Confidence):
          f64 vanilla, basket (N = 3, 4, 16)   1e-13 relative  (the exponential's last bits)
          f32 vanilla, basket                   2e-6 relative on E, 2e-5 on CI
          f64 CVA                               2e-11 relative (the reference forms dp_j as a difference of two
                                                exponentials near 1: 1e-12; this engine via expm1)
          f32 CVA                               1e-3 relative (the same difference in float loses four digits: SURVEY 2.3 #9)
  (3) the kernel's own fp64 (sum, sum2) -- hot loop, flushes, final reduction -- against the fp64 sums of its per-path
      values: 3e-6 relative (f32 partial sums) / 1e-12 (f64).
  (4) the estimate the HIP path returns (fp64 accumulation) against the golden: as (2) plus what the reference's
      accumulator loses -- f32 at 1e5 paths: 5e-5 relative (a float running sum of 1e6).
      At BASELINE configs[0]'s own size (1e6 paths: the six vanilla goldens of seeds 12345 / 777 / 1, both precisions) the
      float accumulator's loss is DERIVED per case instead (accumulator_model below): the reference adds path k to a running sum
      held in `real` (MonteCarloHost.c:196-219), and a floating-point running sum is always a multiple of its own ulp -- so each
      addition rounds the ADDEND to the sum's ulp: error_k = rn(x_k / ulp(S_k)) ulp(S_k) - x_k, a deterministic function of the
      values.  At 1e6 fp32 paths the sum of payoffs passes 2^23 (ulp 1) and the sum of squares 2^28 (ulp 32): payoffs^2 below 16
      vanish altogether -- a systematic loss, not a zero-mean walk (a first version of this test modelled a random walk of
      sqrt(sum ulp^2 / 12) and the confidence missed its 6-sigma bound at 8.6 sigma).  The model applied to the HIP per-path
      values reproduces the compiled reference's printed numbers to 5e-7 (E) and 1e-6 (CI) where the exact sums are 2e-5 ... 4e-5
      and 1.6e-4 away; the test asserts (a) golden == model within (2)'s tolerance and (b) the HIP estimate (fp64 accumulation) is off
      the golden by exactly the modelled loss, to the same tolerance.
"""
import math

import numpy as np
import pytest

from conftest import fromhex, load_golden

pytestmark = pytest.mark.gpu

MC = load_golden("ref_mc.json")["cases"]
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


def f64(a):
    return np.asarray(a, dtype=np.float64)


def _own_sums(e, vals, X):
    v = f64(vals)
    assert e.n == len(v)
    assert e.sum == pytest.approx(v.sum(), rel=SUMS[X])
    assert e.sum2 == pytest.approx((v * v).sum(), rel=2 * SUMS[X])


def accumulator_model(vals, X, r, t):
    """(E, CI) with exact sums and (E, CI) as the reference's sequential accumulation in `real` delivers them, from a model of that
    accumulation: every addition rounds the addend to the running sum's ulp (the sum itself is a multiple of it).  Closing as
    MonteCarloHost.c:220-228: E discounted, the confidence half-width not."""
    v = f64(vals)
    n = len(v)
    ty = np.float32 if X == "f32" else np.float64
    sums = []
    for seq in (v, v * v):
        run = np.cumsum(seq)
        ulp = np.spacing(np.maximum(run, np.finfo(ty).tiny).astype(ty)).astype(np.float64)
        sums.append((run[-1], run[-1] + float((np.rint(seq / ulp) * ulp - seq).sum())))

    def close(S, S2):
        return math.exp(-r * t) * S / n, 1.96 * math.sqrt((n * S2 - S * S) / (n * (n - 1.0))) / math.sqrt(n)
    return close(sums[0][0], sums[1][0]), close(sums[0][1], sums[1][1])


def _basket_inputs(c):
    return dict(c["basket"], p=[[fromhex(x) for x in row] for row in c["factor"]])


# ---- vanilla -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in MC if c["kind"] == "vanilla"],
                         ids=lambda c: f"{c['X']}-{c['paths']}-{c['seed']}")
def test_vanilla_hot_kernel_on_reference_stream_vs_golden(eng, po, c):
    """Up to 1e6 paths: BASELINE configs[0] itself (dp/vanillaOpt.cu:22-26 option, dp/MonteCarloHost.c:185-229 loop), the numbers
    the compiled reference printed for seeds 12345 / 777 / 1 in both precisions."""
    X, opt, n = c["X"], c["opt"], c["paths"]
    z = po.host_gaussians(X, c["seed"], n)
    e, vals = eng.vanilla_from_normals(opt, z, X)
    want, _ = po.dev_vanilla_on_normals(X, opt, z)
    assert np.abs(f64(vals) - f64(want)).max() <= PAY[X] * opt["s"]                      # (1)
    closed = po.ref_close(X, vals, 1, opt["r"], opt["t"])                                # (2)
    ge, gci = fromhex(c["expected"]), fromhex(c["confidence"])
    big = n > 100000
    # (2) at 1e6 fp32 paths: a per-path difference of d <= 2e-6 S now and then moves an addend across a rounding boundary of the
    # running sum (ulp 1 for the payoffs, 32 for their squares): a few ulps on 1.09e7 / 3.3e8 -- 3e-6 / 4e-5 instead of 2e-6 / 2e-5
    if n >= 1000:
        assert closed["expected"] == pytest.approx(ge, rel=1e-13 if X == "f64" else (3e-6 if big else 2e-6))
        assert closed["confidence"] == pytest.approx(gci, rel=1e-12 if X == "f64" else (4e-5 if big else 2e-5))
    _own_sums(e, vals, X)                                                                # (3)
    if n >= 1000:                                                                        # (4)
        (e_exact, ci_exact), (e_model, ci_model) = accumulator_model(vals, X, opt["r"], opt["t"])
        loss_e, loss_ci = (e_exact - e_model) / e_exact, (ci_exact - ci_model) / ci_exact
        tol_e, tol_ci = (1e-12, 1e-11) if X == "f64" else ((3e-6, 4e-5) if big else (2e-6, 2e-5))
        # (a) the model of the reference's accumulator, fed with the HIP per-path values, lands on the reference's printed numbers
        assert e_model == pytest.approx(ge, rel=tol_e) and ci_model == pytest.approx(gci, rel=tol_ci), (loss_e, loss_ci)
        # (b) the HIP estimate (fp64 accumulation inside the kernel) is off the golden by that modelled loss and nothing else
        assert e.expected - ge == pytest.approx(e_exact - e_model, abs=tol_e * ge)
        assert e.confidence - gci == pytest.approx(ci_exact - ci_model, abs=tol_ci * gci)
        if big and X == "f32":   # what the float accumulator loses at BASELINE configs[0]'s size: 2e-5 ... 4e-5 of E, 1.6e-4 of the CI
            assert 5e-6 < abs(loss_e) < 1e-4 and 5e-5 < loss_ci < 4e-4, (loss_e, loss_ci)
        elif not big:            # up to 1e5 paths the constants of rounds 3-4 hold
            assert e.expected == pytest.approx(ge, rel=1e-12 if X == "f64" else 5e-5)
            assert e.confidence == pytest.approx(gci, rel=1e-11 if X == "f64" else 5e-4)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n", [1, 3, 5, 1023, 262144 + 3])
def test_vanilla_from_normals_edges(eng, po, X, n):
    """Partial last unit (masked kernel), a grid larger than the range, a lane running many trips."""
    opt = dict(s=105.0, k=95.0, r=0.02, v=0.35, t=2.5)
    z = po.host_gaussians(X, 31, n)
    e, vals = eng.vanilla_from_normals(opt, z, X)
    want, o = po.dev_vanilla_on_normals(X, opt, z)
    assert np.abs(f64(vals) - f64(want)).max() <= PAY[X] * opt["s"] * 3
    _own_sums(e, vals, X)
    assert e.sum == pytest.approx(o["sum"], rel=SUMS[X], abs=PAY[X] * opt["s"])


# ---- basket ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in MC if c["kind"] == "basket"],
                         ids=lambda c: f"{c['X']}-n{c['n']}-{c['corr_name']}-{c['paths']}")
def test_basket_kernels_on_reference_stream_vs_golden(eng, po, c):
    X, n, paths = c["X"], c["n"], c["paths"]
    b = _basket_inputs(c)
    g = po.host_gaussians(X, c["seed"], paths * n).reshape(paths, n)
    # f64: the dp reference CPU path forms the diffusion without the volatility (MonteCarloHost.c:180); its goldens can
    # only be met on that model -- a test switch of the hook folds the constants accordingly, the kernel is the same
    no_vol = X == "f64"
    e, vals = eng.basket_from_normals(b, g, X, no_vol=no_vol)
    want, _ = po.dev_basket_on_normals(X, b, g, po.BASKET_NO_VOL if no_vol else 0)
    # (1); on the no-volatility model the "volatility" is 100 %: payoffs reach the thousands, the bound follows the value
    assert (np.abs(f64(vals) - f64(want)) <= PAY[X] * 4 * np.maximum(100.0, f64(want) + b["k"])).all()
    closed = po.ref_close(X, vals, 1, b["r"], b["t"])                                    # (2)
    ge, gci = fromhex(c["expected"]), fromhex(c["confidence"])
    assert closed["expected"] == pytest.approx(ge, rel=1e-13 if X == "f64" else 2e-6)
    assert closed["confidence"] == pytest.approx(gci, rel=1e-12 if X == "f64" else 2e-5)
    _own_sums(e, vals, X)                                                                # (3)
    assert e.expected == pytest.approx(ge, rel=1e-12 if X == "f64" else 5e-5)            # (4)
    assert e.confidence == pytest.approx(gci, rel=1e-11 if X == "f64" else 5e-4)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n", [3, 4, 16])
def test_basket_true_device_model_on_reference_stream(eng, po, X, n):
    """The product's own model (volatility in the diffusion, as the device and the sp reference have it) on the
    reference's stream, both precisions, against the oracle's device formulas; in fp32 that IS the sp golden's model."""
    c = next(c for c in MC if c["kind"] == "basket" and c["X"] == X and c["n"] == n and c["corr_name"] == "equi0.5" and c["paths"] == 1000)
    b = _basket_inputs(c)
    g = po.host_gaussians(X, c["seed"], 1000 * n).reshape(1000, n)
    e, vals = eng.basket_from_normals(b, g, X)
    want, o = po.dev_basket_on_normals(X, b, g, 0)
    assert np.abs(f64(vals) - f64(want)).max() <= PAY[X] * 100.0 * 2
    assert e.sum == pytest.approx(o["sum"], rel=SUMS[X]) and e.sum2 == pytest.approx(o["sum2"], rel=2 * SUMS[X])


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n,paths", [(5, 777), (16, 70001), (4, 600001), (3, 2)])
def test_basket_from_normals_other_sizes(eng, po, X, n, paths):
    """The generic kernel (5 assets), odd path counts on the two-paths-per-lane kernels, hot loops of many trips."""
    rng = np.random.default_rng(n * 1000 + paths)
    a = rng.standard_normal((n, n + 2))
    corr = a @ a.T
    d = np.sqrt(np.diag(corr))
    L = po.chol(X, corr / d[:, None] / d[None, :])
    b = dict(s=rng.uniform(50, 150, n).tolist(), v=rng.uniform(0.1, 0.5, n).tolist(), p=L.tolist(), d=rng.uniform(-0.02, 0.02, n).tolist(),
             w=(np.ones(n) / n).tolist(), k=100.0, t=1.5, r=0.03)
    g = po.host_gaussians(X, 5, paths * n).reshape(paths, n)
    e, vals = eng.basket_from_normals(b, g, X)
    want, o = po.dev_basket_on_normals(X, b, g, 0)
    assert np.abs(f64(vals) - f64(want)).max() <= PAY[X] * 150.0 * 4
    _own_sums(e, vals, X)
    assert e.sum == pytest.approx(o["sum"], rel=SUMS[X])


# ---- CVA ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in MC if c["kind"] == "cva"],
                         ids=lambda c: f"{c['X']}-{c['cva']['n_grid']}x{c['paths']}-{c['seed']}")
def test_cva_kernel_in_host_order_on_reference_stream_vs_golden(eng, po, c):
    X, cva, paths = c["X"], c["cva"], c["paths"]
    z = po.host_gaussians(X, c["seed"], paths * cva["n_grid"]).reshape(paths, cva["n_grid"])
    e, vals = eng.cva_from_normals(cva, z, X, host_order=True)
    want, _ = po.dev_cva_on_normals(X, cva, z, po.CVA_HOST_ORDER)          # product dp_j and tau = 0 rule, host ordering
    assert np.abs(f64(vals) - f64(want)).max() <= CVA_ABS[X] * 2                          # (1)
    closed = po.ref_close(X, vals, 0, cva["r"], cva["t"])                                 # (2)
    ge, gci = fromhex(c["expected"]), fromhex(c["confidence"])
    assert closed["expected"] == pytest.approx(ge, rel=2e-11 if X == "f64" else 1e-3)
    assert closed["confidence"] == pytest.approx(gci, rel=2e-10 if X == "f64" else 1e-2)
    _own_sums(e, vals, X)                                                                 # (3)
    assert e.expected == pytest.approx(ge, rel=2e-11 if X == "f64" else 1e-3)             # (4)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_grid,paths", [(1, 100), (7, 1001), (256, 3000), (250, 20001)])
def test_cva_kernel_device_order_on_reference_stream(eng, po, X, n_grid, paths):
    """The product's own ordering (exposure at the NEW spot: MonteCarloKernel.cu:249-252) on the reference's stream
    against the oracle's device loop: grids that end in a partial block, in the intrinsic-value date (256: residual
    maturity exactly 0) and in a negative residual maturity (250 in fp64)."""
    cva = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=n_grid)
    z = po.host_gaussians(X, 99, paths * n_grid).reshape(paths, n_grid)
    e, vals = eng.cva_from_normals(cva, z, X)
    want, o = po.dev_cva_on_normals(X, cva, z, 0)
    assert np.abs(f64(vals) - f64(want)).max() <= CVA_ABS[X] * 2
    _own_sums(e, vals, X)
    assert e.sum == pytest.approx(o["sum"], rel=SUMS[X] * 10)


def test_from_normals_refuses_what_it_does_not_implement(mc, eng, po):
    z = po.host_gaussians("f64", 1, 16)
    opt = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0)
    eng.set_antithetic(True)
    try:
        with pytest.raises(mc.McError, match="plain estimator"):
            eng.vanilla_from_normals(opt, z, "f64")
    finally:
        eng.set_antithetic(False)
    # and the hook leaves the context as it found it: a pricing call afterwards is the Philox one
    a = eng.vanilla(opt, 4096, precision="f64")
    eng.vanilla_from_normals(opt, z, "f64")
    b = eng.vanilla(opt, 4096, precision="f64")
    assert (a.sum, a.sum2) == (b.sum, b.sum2)
