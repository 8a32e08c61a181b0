"""MC_NORMALS_F32: the reference's own "double precision" arithmetic as an opt-in mode of the fp64 kernels.

The reference's dp kernels draw `double z = curand_normal(...)` -- a FLOAT normal widened to double
(dp/MonteCarloKernel.cu:68,78,250).  mc_context_set_normals(ctx, MC_NORMALS_F32) / MC_F64_NORMALS=f32 makes
the fp64 kernels do the same: one Philox block yields FOUR normals through the fp32 hardware transcendentals
(mc_rng.hpp: GenPhiloxF32N), everything downstream of the normal stays fp64.  Default stays true fp64 normals.

Checked here:
  * the mode's normals ARE the fp32 kernels' normals, bit for bit (same block, widened) -- and within the fp32 bound
    (2e-6) of the oracle's twin (orc_set_normals_f32);
  * with the normals taken as given, everything downstream is fp64-exact: the kernels' per-path values against the
    oracle's device formulas evaluated on THE DEVICE'S OWN normals at the fp64 bounds (1e-14 S per payoff, 1e-13 per
    CVA value), for the vanilla kernels, every basket family (kernel-argument, LDS-staged, tiled, generic) and the CVA
    kernel -- C4's 16 assets and C5's 256 dates included; sums at 1e-12;
  * end to end against the oracle in its own f32-normals mode at the fp32 bounds (the two sides' float normals differ
    in the last bits: v_log/v_sin/v_cos_f32 vs glibc);
  * prices agree with the native mode and with closed forms within the confidence intervals.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x4D435F4D49333535
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA0 = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6)
BS_EXACT = 10.386270784322328


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


@pytest.fixture()
def eng(mc):
    e = mc.Engine(0)
    e.set_normals("f32")
    yield e
    e.close()


def f64(a):
    return np.asarray(a, dtype=np.float64)


def basket_inputs(mc, n, rho=0.5):
    v = [0.3 if i % 2 == 0 else 0.2 for i in range(n)]
    L, bad = mc.chol(np.full((n, n), rho) + (1 - rho) * np.eye(n), "f64")
    assert bad == 0
    return dict(s=[100.0] * n, v=v, p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n, k=100.0, t=1.0, r=0.048790164)


def device_normals(eng, domain, first, n_units, n_per_unit):
    """The device's own normals of units first .. first + n_units - 1: (n_units, n_per_unit), blocks of 4."""
    blocks = (n_per_unit + 3) // 4
    z = np.concatenate([eng.normals(SEED, domain, first, n_units, b, "f64") for b in range(blocks)], axis=1)
    return z[:, :n_per_unit]


@pytest.mark.parametrize("domain,block,first", [(1, 0, 0), (2, 3, 12345), (3, 63, (1 << 32) - 100)])
def test_mode_normals_are_the_fp32_normals_widened(mc, eng, po, domain, block, first):
    n = 512
    got = eng.normals(SEED, domain, first, n, block, "f64")
    assert got.shape == (n, 4)
    native32 = eng.normals(SEED, domain, first, n, block, "f32")
    assert (got == native32.astype(np.float64)).all()
    with po.normals_f32():
        assert po.dev_npb("f64") == 4
        want = np.array([po.dev_normals("f64", SEED, domain, first + u, block) for u in range(n)])
    assert po.dev_npb("f64") == 8      # native fp64: eight normals per block (stream version 2)
    assert np.abs(got - want).max() <= 2e-6


@pytest.mark.parametrize("first,n", [(0, 1), (3, 9), (5, 20000), ((1 << 34) - 7, 41)])
def test_vanilla_f64_on_fp32_normals(mc, eng, po, first, n):
    got = eng.vanilla_paths(VAN, n, SEED, first, "f64")
    u0, u1 = first // 4, (first + n + 3) // 4
    z = device_normals(eng, 1, u0, u1 - u0, 4).reshape(-1)[first - 4 * u0: first - 4 * u0 + n]
    want, o = po.dev_vanilla_on_normals("f64", VAN, z)
    assert np.abs(got - want).max() <= 1e-14 * VAN["s"]
    e = eng.vanilla(VAN, n, SEED, first, "f64")
    assert e.n == n and e.sum == pytest.approx(o["sum"], rel=1e-12, abs=1e-12) and e.sum2 == pytest.approx(o["sum2"], rel=1e-12, abs=1e-10)
    with po.normals_f32():
        _, oo = po.dev_vanilla("f64", VAN, SEED, first, n)
    if n > 1000:
        assert e.expected == pytest.approx(oo["expected"], rel=3e-6)


@pytest.mark.parametrize("n_assets", [1, 3, 4, 7, 8, 9, 12, 16, 17, 24, 33])
def test_basket_f64_families_on_fp32_normals(mc, eng, po, n_assets):
    """Kernel-argument (1..3), LDS-staged (4..8), tiled (9..32) and generic (33) fp64 kernels in the mode."""
    b = basket_inputs(mc, n_assets)
    first, n = 3, 1500
    for anti in (False, True):
        eng.set_antithetic(anti)
        try:
            got = eng.basket_paths(b, n, SEED, first, "f64")
            e = eng.basket(b, n, SEED, first, "f64")
        finally:
            eng.set_antithetic(False)
        if not anti:
            g = device_normals(eng, 2, first, n, n_assets)
            want, o = po.dev_basket_on_normals("f64", b, g, 0)
            assert np.abs(got - want).max() <= 1e-14 * 100.0 * 4
            assert e.sum == pytest.approx(o["sum"], rel=1e-12) and e.sum2 == pytest.approx(o["sum2"], rel=1e-12)
        with po.normals_f32():
            want32, oo = po.dev_basket("f64", b, SEED, first, n, antithetic=anti)
        assert np.abs(got - want32).max() <= 2e-6 * 100.0 * 4
        assert e.sum == pytest.approx(oo["sum"], rel=3e-6)


@pytest.mark.parametrize("n_grid", [1, 2, 3, 5, 250, 256])
def test_cva_f64_on_fp32_normals(mc, eng, po, n_grid):
    c = dict(CVA0, n_grid=n_grid)
    first, n = 7, 700
    got = eng.cva_paths(c, n, SEED, first, "f64")
    z = device_normals(eng, 3, first, n, n_grid)
    want, o = po.dev_cva_on_normals("f64", c, z, 0)
    assert np.abs(got - want).max() <= 1e-13
    e = eng.cva(c, n, SEED, first, "f64")
    assert e.sum == pytest.approx(o["sum"], rel=1e-12) and e.sum2 == pytest.approx(o["sum2"], rel=1e-12)
    eng.set_antithetic(True)
    try:
        ga = eng.cva_paths(c, n, SEED, first, "f64")
    finally:
        eng.set_antithetic(False)
    wa, _ = po.dev_cva_on_normals("f64", c, z, 0, antithetic=True)
    assert np.abs(ga - wa).max() <= 1e-13
    with po.normals_f32():
        w32, _ = po.dev_cva("f64", c, SEED, first, n)
    assert np.abs(got - w32).max() <= 2e-5


def test_full_size_prices_agree_with_native_mode_and_closed_forms(mc, eng):
    native = mc.Engine(0)
    try:
        a = eng.vanilla(VAN, 10 ** 8, SEED, 0, "f64")
        b = native.vanilla(VAN, 10 ** 8, SEED, 0, "f64")
        assert abs(a.expected - BS_EXACT) < 3.5 / 1.96 * a.confidence
        assert abs(a.expected - b.expected) < 3.5 / 1.96 * math.hypot(a.confidence, b.confidence)
        bk = basket_inputs(mc, 16)
        a = eng.basket(bk, 2 * 10 ** 7, SEED, 0, "f64")
        b = native.basket(bk, 2 * 10 ** 7, SEED, 0, "f64")
        assert abs(a.expected - b.expected) < 3.5 / 1.96 * math.hypot(a.confidence, b.confidence)
        c = dict(CVA0, n_grid=256)
        a = eng.cva(c, 10 ** 6, SEED, 0, "f64")
        b = native.cva(c, 10 ** 6, SEED, 0, "f64")
        assert abs(a.expected - b.expected) < 3.5 / 1.96 * math.hypot(a.confidence, b.confidence)
        # shard additivity in the mode: the counter is still the global path index
        lo = eng.cva(c, 300000, SEED, 0, "f64")
        hi = eng.cva(c, 700000, SEED, 300000, "f64")
        assert lo.sum + hi.sum == pytest.approx(a.sum, rel=1e-12)
    finally:
        native.close()


def test_mode_is_refused_where_it_is_not_implemented(mc, eng):
    with pytest.raises(mc.McError, match="native normals"):
        eng.vanilla_greeks(VAN, 1000, SEED, 0, "f64")
    eng.set_generator("xorwow", 0)
    try:
        with pytest.raises(mc.McError, match="Philox"):
            eng.vanilla(VAN, 4096, SEED, 0, "f64")
    finally:
        eng.set_generator("philox", 0)
    # the fp32 entry points are unaffected by the mode -- their Greeks included (planes_run refused them all)
    native = mc.Engine(0)
    try:
        assert eng.vanilla(VAN, 10 ** 6, SEED, 0, "f32").sum == native.vanilla(VAN, 10 ** 6, SEED, 0, "f32").sum
        for a, b in zip(eng.vanilla_greeks(VAN, 10 ** 5, SEED, 3, "f32"), native.vanilla_greeks(VAN, 10 ** 5, SEED, 3, "f32")):
            assert a.sum == b.sum and a.sum2 == b.sum2 and a.n == b.n
        for a, b in zip(eng.vanilla_greeks_lr(VAN, 10 ** 5, SEED, 3, "f32"), native.vanilla_greeks_lr(VAN, 10 ** 5, SEED, 3, "f32")):
            assert a.sum == b.sum and a.sum2 == b.sum2
        bk = basket_inputs(mc, 4)
        pa, da, va = eng.basket_greeks(bk, 20000, SEED, 0, "f32")
        pb, db, vb = native.basket_greeks(bk, 20000, SEED, 0, "f32")
        assert pa.sum == pb.sum and [x.sum for x in da] == [x.sum for x in db] and [x.sum for x in va] == [x.sum for x in vb]
        c = dict(CVA0, n_grid=16)
        for a, b in zip(eng.cva_greeks(c, 20000, SEED, 0, "f32"), native.cva_greeks(c, 20000, SEED, 0, "f32")):
            assert a.sum == b.sum and a.sum2 == b.sum2
        with pytest.raises(mc.McError, match="native normals"):      # the fp64 ones still refuse it
            eng.cva_greeks(c, 1000, SEED, 0, "f64")
    finally:
        native.close()


def test_environment_variable_selects_the_mode_for_the_legacy_symbols(mc, monkeypatch):
    monkeypatch.setenv("MC_F64_NORMALS", "f32")
    e = mc.Engine(0)
    try:
        # the Python mirror follows the C context, which reads the variable at creation (normals() used to allocate
        # 8 per unit and hand back a half-written array)
        assert e._normals_f32 is True
        z = e.normals(SEED, 1, 0, 8, 0, "f64")
        assert z.shape == (8, 4) and np.isfinite(z).all() and np.abs(z).max() < 6.77
        assert np.array_equal(z, e.normals(SEED, 1, 0, 8, 0, "f32").astype(np.float64))
    finally:
        e.close()
    monkeypatch.delenv("MC_F64_NORMALS")
    e = mc.Engine(0)
    try:
        assert e._normals_f32 is False and e.normals(SEED, 1, 0, 8, 0, "f64").shape == (8, 8)
    finally:
        e.close()
