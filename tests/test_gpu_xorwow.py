"""XORWOW as the engine's second generator (the reference draws from cuRAND XORWOW: dp/MonteCarloKernel.cu:
285-290,68,78,250).  On the GPU: the generator's words against the oracle's (which tests/test_rocrand_xcheck.py pins to
rocRAND's own engine word for word), then the three products in XORWOW mode against the oracle's XORWOW twin -- one
sequence per lane, lane l = subsequence base + l pricing units unit0 + l, unit0 + l + lanes, ... -- per path and per
sum, plus the statistics a generator must get right and the contract's error cases."""
import math

import numpy as np
import pytest

from test_gpu_parity import BS_EXACT, CVA0, SEED, TOL, VAN, basket_inputs, cva_analytic

pytestmark = pytest.mark.gpu
NPB = {"f32": 4, "f64": 8}   # paths per vanilla unit (f64: stream version 2)


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


def lanes_of(eng, n_units):
    return min(eng.blocks, -(-n_units // 256)) * 256


def test_xorwow_words_match_oracle(mc, po):
    with mc.Engine(0) as e:
        for seed, first, n_sub, each in ((SEED, 0, 600, 9), (1, 2 ** 20 - 3, 70, 5), (2 ** 64 - 1, 2 ** 47, 33, 4), (0, 2 ** 48 - 40, 40, 3)):
            got = e.xorwow_words(seed, first, n_sub, each)
            want = np.array([po.xorwow_words(seed, first + s, each) for s in range(n_sub)], dtype=np.uint32)
            assert (got == want).all(), (seed, first)
        with pytest.raises(mc.McError, match="below 2\\^48"):
            e.xorwow_words(1, 2 ** 48 - 3, 8, 2)


@pytest.mark.parametrize("anti", [False, True])
@pytest.mark.parametrize("blocks,first,n", [(1, 0, 3000), (2, 5, 4001), (0, 3, 9001)])
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_xorwow_products_match_oracle(mc, po, X, blocks, first, n, anti):
    base = 1000
    with mc.Engine(0, blocks) as e:
        e.set_generator("xorwow", base)
        e.set_antithetic(anti)
        # vanilla: units are Philox-block-sized groups of paths; the range starts and ends inside a unit
        u0, u1 = first // NPB[X], -(-(first + n) // NPB[X])
        got = e.vanilla_paths(VAN, n, SEED, first, X).astype(np.float64)
        est = e.vanilla(VAN, n, SEED, first, X)
        with po.xorwow_mode(SEED, base, lanes_of(e, u1 - u0), u0):
            # the oracle asks for a unit's normals when it meets the unit's first live path: ascending, once each
            want, o = po.dev_vanilla(X, VAN, SEED, first, n, antithetic=anti)
        assert np.abs(got - want.astype(np.float64)).max() <= TOL[X]["pay"] * VAN["s"]
        assert est.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and est.sum2 == pytest.approx(o["sum2"], rel=TOL[X]["rel"])
        assert est.n == n and est.confidence == pytest.approx(o["confidence"], rel=4 * TOL[X]["rel"])
        # baskets (any n runs the generic kernel; 5 and 13 are padded to 8 and 16 normals per path) and the CVA
        for n_assets in (1, 4, 5, 13, 40):
            b = basket_inputs(mc, n_assets, X, rho=0.4)
            m = n // 4
            got = e.basket_paths(b, m, SEED, first, X).astype(np.float64)
            est = e.basket(b, m, SEED, first, X)
            with po.xorwow_mode(SEED, base, lanes_of(e, m), first):
                want, o = po.dev_basket(X, b, SEED, first, m, antithetic=anti)
            assert np.abs(got - want.astype(np.float64)).max() <= TOL[X]["pay"] * 100.0 * 4, n_assets
            assert est.sum == pytest.approx(o["sum"], rel=2 * TOL[X]["rel"]), n_assets
        for n_grid in (1, 25, 64, 250):
            c = dict(CVA0, n_grid=n_grid)
            m = n // 8
            got = e.cva_paths(c, m, SEED, first, X).astype(np.float64)
            est = e.cva(c, m, SEED, first, X)
            with po.xorwow_mode(SEED, base, lanes_of(e, m), first):
                want, o = po.dev_cva(X, c, SEED, first, m, antithetic=anti)
            assert np.abs(got - want.astype(np.float64)).max() <= TOL[X]["cva"], n_grid
            assert est.sum == pytest.approx(o["sum"], rel=2 * TOL[X]["rel"]), n_grid


def test_xorwow_many_trips_per_lane(mc, po):
    """One workgroup, 3e5 paths: every lane walks hundreds of units through its own sequence."""
    with mc.Engine(0, blocks=1) as e:
        e.set_generator("xorwow", 0)
        for X in ("f32", "f64"):
            n = 300_001
            est = e.vanilla(VAN, n, 77, 0, X)
            with po.xorwow_mode(77, 0, 256, 0):
                _, o = po.dev_vanilla(X, VAN, 77, 0, n, want_paths=False)
            assert est.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and est.sum2 == pytest.approx(o["sum2"], rel=TOL[X]["rel"])


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_xorwow_statistics_and_prices(mc, X):
    """What any generator owes the estimator: the closed-form prices within the confidence interval, and samples that
    really are different when the seed or the subsequence base changes -- and identical when nothing does."""
    with mc.Engine(0) as e:
        e.set_generator("xorwow", 0)
        a = e.vanilla(VAN, 10 ** 8, SEED, 0, X)
        assert abs(a.expected - BS_EXACT) < 3.5 / 1.96 * a.confidence and a.confidence == pytest.approx(0.003022, rel=5e-3)
        again = e.vanilla(VAN, 10 ** 8, SEED, 0, X)
        assert (again.sum, again.sum2) == (a.sum, a.sum2)                       # fixed seed: the same call, the same bits (:289)
        assert e.vanilla(VAN, 10 ** 8, SEED + 1, 0, X).sum != a.sum
        e.set_generator("xorwow", e.blocks * 256)                               # the next device's subsequences
        other = e.vanilla(VAN, 10 ** 8, SEED, 0, X)
        assert other.sum != a.sum and abs(other.expected - a.expected) < 4 / 1.96 * math.hypot(a.confidence, other.confidence)
        e.set_generator("xorwow", 0)
        c = dict(CVA0, n_grid=64)
        v = e.cva(c, 10 ** 6, SEED, 0, X)
        assert abs(v.expected - cva_analytic(c)) < 3.5 / 1.96 * v.confidence + 2e-6
        b = basket_inputs(mc, 4, X)
        xb = e.basket(b, 10 ** 7, SEED, 0, X)
        e.set_generator("philox")
        pb = e.basket(b, 10 ** 7, SEED, 0, X)
        assert abs(xb.expected - pb.expected) < 4 / 1.96 * math.hypot(xb.confidence, pb.confidence)
        assert e.vanilla(VAN, 10 ** 8, SEED, 0, X).sum != a.sum                # back on Philox: another stream


def test_xorwow_contract(mc):
    with mc.Engine(0) as e:
        e.set_generator("xorwow", 0)
        with pytest.raises(mc.McError, match="one call is one launch"):
            e.basket(basket_inputs(mc, 4, "f32"), 3 * 10 ** 9, SEED, 0, "f32")          # more than 2^31 units
        with pytest.raises(mc.McError, match="one call is one launch"):
            e.cva(dict(CVA0, n_grid=4), 1000, SEED, (1 << 32) - 500, "f32")             # across a multiple of 2^32
        with pytest.raises(mc.McError, match="Philox generator with native normals only"):
            e.vanilla_greeks(VAN, 1000, SEED, 0, "f64")
        with pytest.raises(mc.McError):
            e.set_generator("xorwow", 2 ** 48)
            e.vanilla(VAN, 1000, SEED, 0, "f64")


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_xorwow_aligned_ranges_run_the_hot_kernels(mc, po, X):
    """Whole-unit ranges (what the legacy symbols ask for) run the hot vanilla kernels in XORWOW mode too: same lanes and
    draws as the masked form, so the sums agree with the oracle's XORWOW twin and with an unaligned neighbour range."""
    with mc.Engine(0, blocks=2) as e:
        e.set_generator("xorwow", 3)
        for anti in (False, True):
            e.set_antithetic(anti)
            n = 400_000
            est = e.vanilla(VAN, n, SEED, 0, X)
            with po.xorwow_mode(SEED, 3, 512, 0):
                _, o = po.dev_vanilla(X, VAN, SEED, 0, n, want_paths=False, antithetic=anti)
            assert est.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"]) and est.sum2 == pytest.approx(o["sum2"], rel=TOL[X]["rel"])
            paths = e.vanilla_paths(VAN, 4000, SEED, 0, X).astype(np.float64)       # the masked form, same start
            with po.xorwow_mode(SEED, 3, lanes_of(e, 4000 // NPB[X]), 0):
                want, _ = po.dev_vanilla(X, VAN, SEED, 0, 4000, antithetic=anti)
            assert np.abs(paths - want.astype(np.float64)).max() <= TOL[X]["pay"] * VAN["s"]


def test_legacy_symbols_switch_generator_through_MC_RNG():
    """MC_RNG=xorwow reaches dev_vanillaOpt / dev_cvaEquityOption (legacy_abi.c): another sample than Philox's, the same
    one on every run (the reference's fixed seed, dp/MonteCarloKernel.cu:289), also through the multi-device path."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "drivers", "vanillaOpt_f32")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)], stdout=subprocess.DEVNULL)

    def gpu_price(env):
        out = subprocess.run([exe, "80", "--no-cpu"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        nums = [float(x) for x in re.findall(r"^-?\d+\.\d+", out.stdout, flags=re.M)]
        return nums[0], nums[1]
    philox, xorwow, again = gpu_price({}), gpu_price({"MC_RNG": "xorwow"}), gpu_price({"MC_RNG": "xorwow"})
    assert xorwow == again and xorwow != philox
    assert abs(xorwow[0] - BS_EXACT) < 3.5 / 1.96 * xorwow[1] and abs(xorwow[0] - philox[0]) < 5 / 1.96 * xorwow[1]
    multi = gpu_price({"MC_RNG": "xorwow", "MC_DEVICES": "0"})
    assert multi == xorwow
