"""Greeks beyond the vanilla pathwise ones (the reference prices only): likelihood-ratio delta / vega of
the vanilla call, pathwise AND likelihood-ratio delta and vega per asset of the basket call and of the CVA.  Each estimator has
an oracle twin on the same Philox counters (sums compared at the stated tolerances), a closed-form or
finite-difference target (common random numbers), and runs through both forms of the final reduction."""
import math

import numpy as np
import pytest

from test_gpu_parity import BS_EXACT, CVA0, SEED, TOL, VAN, basket_inputs, cva_analytic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


@pytest.fixture(scope="module")
def eng(mc):
    e = mc.Engine(0)
    yield e
    e.close()


def same(g, o, rel):
    assert g.n == o["n"]
    assert g.sum == pytest.approx(o["sum"], rel=rel) and g.sum2 == pytest.approx(o["sum2"], rel=rel)
    assert g.expected == pytest.approx(o["expected"], rel=rel) and g.confidence == pytest.approx(o["confidence"], rel=4 * rel)


def bs_greeks(o):
    s, k, r, v, t = (o[c] for c in "skrvt")
    d1 = (math.log(s / k) + (r + 0.5 * v * v) * t) / (v * math.sqrt(t))
    return 0.5 * math.erfc(-d1 / math.sqrt(2)), s * math.sqrt(t) * math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_vanilla_likelihood_ratio_greeks(mc, eng, po, X):
    n, first = 50003, 7
    rel = 8 * TOL[X]["rel"]   # the scores multiply the payoff by z^2-sized factors
    for g, o in zip(eng.vanilla_greeks_lr(VAN, n, SEED, first, X), po.dev_vanilla_greeks_lr(X, VAN, SEED, first, n)):
        same(g, o, rel)
    with mc.Engine(0, blocks=1) as one:   # many trips per lane, three planes, one workgroup
        for g, o in zip(one.vanilla_greeks_lr(VAN, 400_001, SEED, 3, X), po.dev_vanilla_greeks_lr(X, VAN, SEED, 3, 400_001)):
            same(g, o, rel)
    big = eng.vanilla_greeks_lr(VAN, 10 ** 8, SEED, 0, X)
    pw = eng.vanilla_greeks(VAN, 10 ** 8, SEED, 0, X)
    delta, vega = bs_greeks(VAN)
    for g, want in zip(big, (BS_EXACT, delta, vega)):
        assert abs(g.expected - want) < 3.5 / 1.96 * g.confidence, (g.expected, want)
    assert big[0].sum == pytest.approx(pw[0].sum, rel=rel)                        # same payoffs
    assert big[1].confidence > 1.5 * pw[1].confidence and big[2].confidence > 1.5 * pw[2].confidence   # the known price of LR
    with pytest.raises(mc.McError, match="v>0 and t>0"):
        eng.vanilla_greeks_lr(dict(VAN, v=0.0), 1000, SEED, 0, X)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_assets", [1, 3, 4, 16, 20, 64])
def test_basket_greeks_match_oracle(mc, eng, po, X, n_assets):
    b = basket_inputs(mc, n_assets, X, rho=0.4)
    b["d"] = [0.01 * ((i % 3) - 1) for i in range(n_assets)]
    b["w"] = [(1.0 + 0.25 * (i % 2)) / n_assets for i in range(n_assets)]
    n, first = (20001 if n_assets < 64 else 3001), (1 << 32) - 7000 if n_assets == 4 else 5     # one case straddles 2^32 units: two launches
    price, delta, vega = eng.basket_greeks(b, n, SEED, first, X)
    op, od, ov = po.dev_basket_greeks(X, b, SEED, first, n)
    rel = 4 * TOL[X]["rel"]
    same(price, op, rel)
    assert len(delta) == len(vega) == n_assets
    for g, o in zip(delta, od):
        same(g, o, rel)
    for g, o in zip(vega, ov):
        # vega terms change sign from path to path: the sum is a difference of comparable magnitudes
        assert g.sum == pytest.approx(o["sum"], rel=20 * rel, abs=n * 100 * TOL[X]["pay"]) and g.sum2 == pytest.approx(o["sum2"], rel=20 * rel)
    # the price plane is the pricing kernel's estimate (folded constants there, the reference's unfolded form here)
    assert price.sum == pytest.approx(eng.basket(b, n, SEED, first, X).sum, rel=max(rel, 1e-10))


def test_basket_greeks_vs_closed_form_and_finite_differences(mc, eng):
    # one asset = the vanilla call: N(d1) and S sqrt(T) phi(d1)
    one = dict(s=[100.0], v=[0.2], p=[[1.0]], d=[0.0], w=[1.0], k=100.0, t=1.0, r=0.048790)
    price, delta, vega = eng.basket_greeks(one, 4 * 10 ** 7, SEED, 0, "f64")
    want_d, want_v = bs_greeks(VAN)
    assert abs(price.expected - BS_EXACT) < 3.5 / 1.96 * price.confidence
    assert abs(delta[0].expected - want_d) < 3.5 / 1.96 * delta[0].confidence
    assert abs(vega[0].expected - want_v) < 3.5 / 1.96 * vega[0].confidence
    # four assets: central differences of the price on common random numbers
    b = basket_inputs(mc, 4, "f64", rho=0.4)
    n = 2 * 10 ** 7
    _, delta, vega = eng.basket_greeks(b, n, SEED, 0, "f64")
    for a in range(4):
        h = 0.05
        up, dn = dict(b, s=list(b["s"])), dict(b, s=list(b["s"]))
        up["s"][a] += h
        dn["s"][a] -= h
        fd = (eng.basket(up, n, SEED, 0, "f64").expected - eng.basket(dn, n, SEED, 0, "f64").expected) / (2 * h)
        assert abs(fd - delta[a].expected) < 4 / 1.96 * delta[a].confidence + 1e-4, (a, fd, delta[a].expected)
        hv = 1e-3
        up, dn = dict(b, v=list(b["v"])), dict(b, v=list(b["v"]))
        up["v"][a] += hv
        dn["v"][a] -= hv
        fd = (eng.basket(up, n, SEED, 0, "f64").expected - eng.basket(dn, n, SEED, 0, "f64").expected) / (2 * hv)
        assert abs(fd - vega[a].expected) < 4 / 1.96 * vega[a].confidence + 2e-3, (a, fd, vega[a].expected)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_grid", [1, 25, 64, 250, 256])
def test_cva_delta_matches_oracle(mc, eng, po, X, n_grid):
    c = dict(CVA0, n_grid=n_grid)
    n = 5001
    cva, delta, vega = eng.cva_greeks(c, n, SEED, 11, X)
    oc, od, ov = po.dev_cva_greeks(X, c, SEED, 11, n)
    rel = 4 * TOL[X]["rel"]
    same(cva, oc, rel)
    same(delta, od, rel)
    # the path term of the vega changes sign from path to path: sums of comparable magnitudes cancel
    assert vega.n == ov["n"] and vega.sum == pytest.approx(ov["sum"], rel=40 * rel, abs=n * 20 * TOL[X]["cva"])
    assert vega.sum2 == pytest.approx(ov["sum2"], rel=40 * rel)
    assert cva.sum == pytest.approx(eng.cva(c, n, SEED, 11, X).sum, rel=rel)     # the pricing kernel's CVA


def test_cva_delta_vs_analytic_and_finite_differences(mc, eng):
    c = dict(CVA0, n_grid=64)
    n = 2 * 10 ** 6
    cva, delta, vega = eng.cva_greeks(c, n, SEED, 0, "f64")
    # E[CVA] = LGD C0(S0) sum_j dp_j e^{r t_j}  =>  d/dS0 = LGD N(d1) sum_j dp_j e^{r t_j}
    dt = c["t"] / c["n_grid"]
    weight = sum((math.exp(-c["defint"] * dt * (j - 1)) - math.exp(-c["defint"] * dt * j)) * math.exp(c["r"] * dt * j)
                 for j in range(1, c["n_grid"] + 1))
    want = c["lgd"] * bs_greeks(c)[0] * weight
    assert abs(cva.expected - cva_analytic(c)) < 3.5 / 1.96 * cva.confidence + 2e-6
    assert abs(delta.expected - want) < 3.5 / 1.96 * delta.confidence + 1e-6, (delta.expected, want)
    h = 0.05
    fd = (eng.cva(dict(c, s=c["s"] + h), n, SEED, 0, "f64").expected - eng.cva(dict(c, s=c["s"] - h), n, SEED, 0, "f64").expected) / (2 * h)
    assert abs(fd - delta.expected) < 4 / 1.96 * delta.confidence + 1e-6
    # vega: d/dsigma of LGD C0(sigma) sum_j dp_j e^{r t_j} = LGD S sqrt(T) phi(d1) sum_j ..., and central differences in sigma
    want_v = c["lgd"] * bs_greeks(c)[1] * weight
    assert abs(vega.expected - want_v) < 3.5 / 1.96 * vega.confidence + 1e-5, (vega.expected, want_v)
    hv = 1e-3
    fd = (eng.cva(dict(c, v=c["v"] + hv), n, SEED, 0, "f64").expected - eng.cva(dict(c, v=c["v"] - hv), n, SEED, 0, "f64").expected) / (2 * hv)
    assert abs(fd - vega.expected) < 4 / 1.96 * vega.confidence + 2e-4, (fd, vega.expected)
    f32 = eng.cva_greeks(c, n, SEED, 0, "f32")
    assert abs(f32[2].expected - want_v) < 3.5 / 1.96 * f32[2].confidence + 1e-4


def test_greeks_run_the_same_through_both_finish_forms(mc):
    b = basket_inputs(mc, 5, "f64", rho=0.3)
    c = dict(CVA0, n_grid=32)
    with mc.Engine(0, blocks=3) as fused, mc.Engine(0, blocks=3) as two:
        two.set_finish(False)
        for rep in range(2):
            a, bq = fused.basket_greeks(b, 100_003, SEED, 1, "f64"), two.basket_greeks(b, 100_003, SEED, 1, "f64")
            assert (a[0].sum, a[0].sum2) == (bq[0].sum, bq[0].sum2)
            assert [(g.sum, g.sum2) for g in a[1] + a[2]] == [(g.sum, g.sum2) for g in bq[1] + bq[2]]
            a, bq = fused.cva_greeks(c, 50_001, SEED, 1, "f32"), two.cva_greeks(c, 50_001, SEED, 1, "f32")
            assert [(g.sum, g.sum2) for g in a] == [(g.sum, g.sum2) for g in bq]
            a, bq = fused.vanilla_greeks_lr(VAN, 200_001, SEED, 1, "f32"), two.vanilla_greeks_lr(VAN, 200_001, SEED, 1, "f32")
            assert [(g.sum, g.sum2) for g in a] == [(g.sum, g.sum2) for g in bq]
        with pytest.raises(mc.McError, match="plain estimator"):
            fused.set_antithetic(True)
            fused.basket_greeks(b, 1000, SEED, 0, "f64")


# ---- likelihood-ratio Greeks of the basket and of the CVA (DESIGN claimed them for all three products) ----
@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_assets", [1, 3, 4, 9, 16, 20])
def test_basket_likelihood_ratio_greeks_match_oracle(mc, eng, po, X, n_assets):
    """The payoff times the score of the joint lognormal density, y = L^-T g formed from the host-inverted factor: against the
    oracle twin on the same normals.  9 and 20 assets: a last chunk of one and of four assets (8 per pass)."""
    b = basket_inputs(mc, n_assets, X, rho=0.4)
    b["d"] = [0.01 * ((i % 3) - 1) for i in range(n_assets)]
    b["w"] = [(1.0 + 0.25 * (i % 2)) / n_assets for i in range(n_assets)]
    n, first = 20001, ((1 << 32) - 7000 if n_assets == 4 else 5)
    price, delta, vega = eng.basket_greeks(b, n, SEED, first, X, lr=True)
    op, od, ov = po.dev_basket_greeks(X, b, SEED, first, n, lr=True)
    rel = 8 * TOL[X]["rel"]
    same(price, op, rel)
    pw = eng.basket_greeks(b, n, SEED, first, X)
    assert (price.sum, price.sum2) == (pw[0].sum, pw[0].sum2)          # the same payoffs, bit for bit, as the pathwise kernel's
    for g, o in zip(delta, od):
        # score terms change sign from path to path; y is a sum of up to n products with entries of L^-T (|M| up to ~3)
        assert g.n == o["n"] and g.sum == pytest.approx(o["sum"], rel=40 * rel, abs=n * 20 * TOL[X]["pay"])
        assert g.sum2 == pytest.approx(o["sum2"], rel=40 * rel)
    for g, o in zip(vega, ov):
        assert g.sum == pytest.approx(o["sum"], rel=40 * rel, abs=n * 4000 * TOL[X]["pay"]) and g.sum2 == pytest.approx(o["sum2"], rel=40 * rel)


def test_basket_likelihood_ratio_greeks_agree_with_pathwise_and_closed_form(mc, eng):
    one = dict(s=[100.0], v=[0.2], p=[[1.0]], d=[0.0], w=[1.0], k=100.0, t=1.0, r=0.048790)
    n = 4 * 10 ** 7
    price, delta, vega = eng.basket_greeks(one, n, SEED, 0, "f64", lr=True)
    van = eng.vanilla_greeks_lr(VAN, n, SEED, 0, "f64")
    want_d, want_v = bs_greeks(VAN)
    assert abs(price.expected - BS_EXACT) < 3.5 / 1.96 * price.confidence
    assert abs(delta[0].expected - want_d) < 3.5 / 1.96 * delta[0].confidence
    assert abs(vega[0].expected - want_v) < 3.5 / 1.96 * vega[0].confidence
    assert delta[0].confidence == pytest.approx(van[1].confidence, rel=0.02) and vega[0].confidence == pytest.approx(van[2].confidence, rel=0.02)
    # four and sixteen correlated assets: both estimators are unbiased for the same derivative
    for n_assets, paths in ((4, 2 * 10 ** 7), (16, 4 * 10 ** 6)):
        b = basket_inputs(mc, n_assets, "f64", rho=0.4)
        b["d"] = [0.02 * ((i % 3) - 1) for i in range(n_assets)]
        _, dl, vl = eng.basket_greeks(b, paths, SEED, 0, "f64", lr=True)
        _, dp, vp = eng.basket_greeks(b, paths, SEED, 0, "f64")
        for a in range(n_assets):
            assert abs(dl[a].expected - dp[a].expected) < 4 / 1.96 * math.hypot(dl[a].confidence, dp[a].confidence), (n_assets, a)
            assert abs(vl[a].expected - vp[a].expected) < 4 / 1.96 * math.hypot(vl[a].confidence, vp[a].confidence), (n_assets, a)
            assert dl[a].confidence > dp[a].confidence        # the known price of the likelihood ratio on a smooth payoff
    with pytest.raises(mc.McError, match="non-singular"):     # the reference driver's own 3 x 3 correlation: L[2][2] = 0
        L, bad = mc.chol(np.full((3, 3), -0.5) + 1.5 * np.eye(3), "f64")
        eng.basket_greeks(dict(s=[100.0] * 3, v=[0.3, 0.2, 0.3], p=L.tolist(), d=[0.0] * 3, w=[1 / 3] * 3, k=100.0, t=1.0, r=0.05), 1000, SEED, 0,
                          "f64", lr=True)


@pytest.mark.parametrize("X", ["f32", "f64"])
@pytest.mark.parametrize("n_grid", [1, 25, 64, 250, 256])
def test_cva_likelihood_ratio_greeks_match_oracle(mc, eng, po, X, n_grid):
    c = dict(CVA0, n_grid=n_grid)
    n = 5001
    cva, delta, vega = eng.cva_greeks(c, n, SEED, 11, X, lr=True)
    oc, od, ov = po.dev_cva_greeks(X, c, SEED, 11, n, lr=True)
    rel = 4 * TOL[X]["rel"]
    same(cva, oc, rel)
    pw = eng.cva_greeks(c, n, SEED, 11, X)
    assert (cva.sum, cva.sum2) == (pw[0].sum, pw[0].sum2)                      # the same CVA plane as the pathwise kernel
    # the scores change sign from path to path and grow with the number of dates: sums of comparable magnitudes cancel
    assert delta.sum == pytest.approx(od["sum"], rel=40 * rel, abs=n * 20 * TOL[X]["cva"]) and delta.sum2 == pytest.approx(od["sum2"], rel=40 * rel)
    assert vega.sum == pytest.approx(ov["sum"], rel=40 * rel, abs=n * 400 * TOL[X]["cva"]) and vega.sum2 == pytest.approx(ov["sum2"], rel=40 * rel)


def test_cva_likelihood_ratio_greeks_vs_analytic(mc, eng):
    c = dict(CVA0, n_grid=16)
    n = 2 * 10 ** 7
    cva, delta, vega = eng.cva_greeks(c, n, SEED, 0, "f64", lr=True)
    dt = c["t"] / c["n_grid"]
    weight = sum((math.exp(-c["defint"] * dt * (j - 1)) - math.exp(-c["defint"] * dt * j)) * math.exp(c["r"] * dt * j)
                 for j in range(1, c["n_grid"] + 1))
    want_d, want_v = (c["lgd"] * g * weight for g in bs_greeks(c))
    assert abs(cva.expected - cva_analytic(c)) < 3.5 / 1.96 * cva.confidence + 2e-6
    assert abs(delta.expected - want_d) < 3.5 / 1.96 * delta.confidence, (delta.expected, want_d)
    assert abs(vega.expected - want_v) < 3.5 / 1.96 * vega.confidence, (vega.expected, want_v)
    pw = eng.cva_greeks(c, n, SEED, 0, "f64")
    assert delta.confidence > 3 * pw[1].confidence and vega.confidence > 3 * pw[2].confidence     # what the scores cost
