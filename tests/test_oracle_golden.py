"""Hop A: the CPU oracle's "host" family reproduces the reference's own outputs BIT FOR BIT.

The expected values in tests/golden/ref_*.json were produced by the unmodified reference
MonteCarloHost.c (see tests/golden/gen_golden.py).  Where oracle/_ref is present (dev
container, or the prebuilt files on the GPU box) the same comparison is repeated live against
the compiled reference on extra seeds.  All comparisons here are exact (==), both precisions.
"""
import numpy as np
import pytest

from conftest import fromhex, load_golden


def test_philox_known_answers(po):
    for c in load_golden("philox_kat.json")["cases"]:
        out = po.philox([int(x, 16) for x in c["ctr"]], [int(x, 16) for x in c["key"]])
        assert out == [int(x, 16) for x in c["out"]]


def test_bs_call_grid_bitwise(po):
    cases = load_golden("ref_bs_call.json")["cases"]
    assert len(cases) > 500
    for c in cases:
        got = po.bs_call(c["X"], c["s"], c["k"], c["r"], c["v"], c["t"])
        assert float(got) == fromhex(c["out"]), c


def test_bs_call_known_answer(po):
    # host_bsCall(S=K=100, r=0.048790, v=0.2, T=1)
    assert po.bs_call("f64", 100, 100, 0.048790, 0.2, 1) == 10.386262209826107
    assert float(po.bs_call("f32", 100, 100, 0.048790, 0.2, 1)) == 10.386257171630859
    # Hastings' CDF is within 1e-5 of the exact Black-Scholes value 10.386270784322328
    assert abs(po.bs_call("f64", 100, 100, 0.048790, 0.2, 1) - 10.386270784322328) < 1e-5


def test_chol_bitwise(po):
    for c in load_golden("ref_chol.json")["cases"]:
        m = [[fromhex(x) for x in row] for row in c["c"]]
        want = np.array([[fromhex(x) for x in row] for row in c["a"]])
        got = po.chol(c["X"], m).astype(np.float64)
        assert (got == want).all(), (c["X"], c["n"], c["name"])
        # structural zeros above the diagonal (reference MonteCarloHost.c:95)
        assert (np.triu(got, 1) == 0).all()


def test_chol_zero_pivot_rule(po):
    # the driver's N=3 matrix is singular: the last pivot is 0 and the column stays 0
    L = po.chol("f64", [[1, -.5, -.5], [-.5, 1, -.5], [-.5, -.5, 1]])
    assert L[2, 2] == 0.0 and L[1, 1] == 0.86602540378443871


def test_uniform_stream_bitwise(po):
    for c in load_golden("ref_uniforms.json")["cases"]:
        got = po.host_uniforms(c["X"], c["seed"], len(c["u"])).astype(np.float64)
        assert (got == np.array([fromhex(x) for x in c["u"]])).all()


def _run_mc(po, c):
    X = c["X"]
    if c["kind"] == "vanilla":
        return po.host_vanilla(X, c["opt"], c["paths"], c["seed"])
    if c["kind"] == "cva":
        return po.host_cva(X, c["cva"], c["paths"], c["seed"])
    b = dict(c["basket"], p=[[fromhex(x) for x in row] for row in c["factor"]])
    return po.host_basket(X, b, c["paths"], c["seed"])  # reference-host formula per precision


def test_monte_carlo_bitwise(po):
    cases = load_golden("ref_mc.json")["cases"]
    kinds = {c["kind"] for c in cases}
    assert kinds == {"vanilla", "basket", "cva"}
    for c in cases:
        r = _run_mc(po, c)
        assert r["expected"] == fromhex(c["expected"]), c
        assert r["confidence"] == fromhex(c["confidence"]), c


def test_basket_factor_matches_chol(po):
    # the factor stored with each basket golden is Chol(corr) (driver: basketOpt.cu:96-99)
    for c in load_golden("ref_mc.json")["cases"]:
        if c["kind"] != "basket":
            continue
        L = po.chol(c["X"], c["corr"]).astype(np.float64)
        assert (L == np.array([[fromhex(x) for x in row] for row in c["factor"]])).all()


def test_survey_known_answers(po):
    """SURVEY 8c known-answer values (seed 12345): fp64 vanilla 10^6 paths, CVA 10^4 x 250."""
    opt = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
    r = po.host_vanilla("f64", opt, 1000000, 12345)
    assert r["expected"] == 10.368900408112896 and r["confidence"] == 0.030238134711122567
    c = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=250)
    r = po.host_cva("f64", c, 10000, 12345)
    assert r["expected"] == 0.18663829631337181 and r["confidence"] == 0.0027413593852186457


# ---- live comparison with the compiled reference, extra seeds ---------------------------
@pytest.mark.parametrize("X", ["f64", "f32"])
def test_live_reference_extra_seeds(po, X):
    if not po.ref_available(X, 3):
        pytest.skip("oracle/_ref not built here")
    ref = po.Ref(X, 3)
    opt = dict(s=105.0, k=95.0, r=0.02, v=0.35, t=2.5)
    for seed in (3, 99, 2024):
        e, ci = ref.vanilla(opt, 5000, seed)
        r = po.host_vanilla(X, opt, 5000, seed)
        assert (r["expected"], r["confidence"]) == (e, ci)
    cva = dict(s=90.0, k=100.0, r=0.01, v=0.4, t=2.0, defint=0.1, lgd=0.45, n_grid=37)
    for seed in (3, 99):
        e, ci = ref.cva(cva, 300, seed)
        r = po.host_cva(X, cva, 300, seed)
        assert (r["expected"], r["confidence"]) == (e, ci)
    corr = [[1, .3, -.2], [.3, 1, .1], [-.2, .1, 1]]
    L = ref.chol(corr)
    assert (L == po.chol(X, corr)).all()
    b = dict(s=[90.0, 100.0, 120.0], v=[0.1, 0.25, 0.4], p=L.tolist(), d=[0.01, 0.0, -0.02],
             w=[0.2, 0.3, 0.5], k=101.0, t=0.75, r=0.03)
    for seed in (3, 99):
        e, ci = ref.basket(b, 4000, seed)
        r = po.host_basket(X, b, 4000, seed)
        assert (r["expected"], r["confidence"]) == (e, ci)


def test_factor_from_cov_bitwise(po):
    """Covariance input: normalise to vols + correlation, then the reference's Chol.  The golden
    correlation and factor come from numpy scalars of the target precision and the compiled reference Chol
    (tests/golden/gen_golden_cov.py); the oracle twin must reproduce v, the correlation and the factor bit for bit."""
    cases = load_golden("ref_cov.json")["cases"]
    assert len(cases) >= 24 and {c["n"] for c in cases} == {3, 4, 16}
    for c in cases:
        cov = np.array([[fromhex(x) for x in row] for row in c["cov"]])
        v, corr, a, bad = po.factor_from_cov(c["X"], cov)
        key = (c["X"], c["n"], c["name"])
        assert (v.astype(np.float64) == np.array([fromhex(x) for x in c["v"]])).all(), key
        assert (corr.astype(np.float64) == np.array([[fromhex(x) for x in row] for row in c["corr"]])).all(), key
        assert (a.astype(np.float64) == np.array([[fromhex(x) for x in row] for row in c["a"]])).all(), key
        assert bad == c["bad_pivots"], key
    # live against the compiled reference where it is present
    if po.ref_available("f64", 4):
        rng = np.random.default_rng(5)
        for X in ("f64", "f32"):
            g = rng.standard_normal((4, 9)) * 0.2
            v, corr, a, bad = po.factor_from_cov(X, g @ g.T)
            assert (po.Ref(X, 4).chol(corr.tolist()) == a).all() and bad == 0


def test_reference_at_O0_gives_the_same_bits(po):
    """The reference's own Makefile builds the host file without -O (Makefile:157,252-253); the goldens were produced at
    -O2 -ffp-contract=off.  Where the compiled reference is present, its -O0 build must return the same bits (SURVEY 8c
    "Determinism") -- bench.py times that build as the -O0 footnote of the CPU baseline."""
    if not po.ref_available("f64", 3, "_O0"):
        pytest.skip("oracle/_ref -O0 build not present")
    for c in load_golden("ref_mc.json")["cases"]:
        if c["kind"] == "vanilla" and c["paths"] <= 100000:
            e, ci = po.Ref(c["X"], 3, "_O0").vanilla(c["opt"], c["paths"], c["seed"])
            assert (e, ci) == (fromhex(c["expected"]), fromhex(c["confidence"])), c
