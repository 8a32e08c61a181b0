"""Steady state of the hot kernels against the oracle (hop B, sums).

The per-path tests of test_gpu_parity.py use <= 50 000 paths on the default 2048 x 256 grid, where every
lane prices at most one unit: only the peeled "partial last trip" of the fp32 kernels runs there.  Here a
SMALL grid (1, 2 or 7 workgroups) prices 2e5 ... 1e6 paths, so a lane runs hundreds of full trips: the main
loop of vanilla_f32_kernel / basket_f32_kernel with its packed-fp32 partial sums, the every-8-trips flush
to fp64 (a lane of Engine(0, blocks=1) pricing 1e6 fp32 vanilla paths flushes 122 times), the pairing of
units (i, i + stride) in the two-paths-per-lane fp32 basket kernels, the strided loops of every fp64 /
tiled / generic / CVA kernel, the range edges and the on-device final reduction -- all compared with the
oracle's plain sequential fp64 sums over the same Philox counters (oracle/mc_oracle_impl.h: orc_dev_*).
Replaces the strided loops of dp/MonteCarloKernel.cu:147-156,193-199,241-262.

Tolerances are the stated sum tolerances of test_gpu_parity.py (f32 3e-6 rel, f64 1e-12 rel): the per-path
rounding differences are zero-mean, so the sums agree far better than any single path.
"""
import numpy as np
import pytest

from test_gpu_parity import CVA0, SEED, TOL, VAN, basket_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


@pytest.fixture(scope="module")
def small(mc):
    """Engines with 1, 2 and 7 workgroups (7: a stride that divides nothing)."""
    engs = {b: mc.Engine(0, blocks=b) for b in (1, 2, 7)}
    yield engs
    for e in engs.values():
        e.close()


def check(e, o, X, n, rel_scale=1.0):
    rel = TOL[X]["rel"] * rel_scale
    assert e.n == n == o["n"]
    assert e.sum == pytest.approx(o["sum"], rel=rel)
    assert e.sum2 == pytest.approx(o["sum2"], rel=rel)
    assert e.expected == pytest.approx(o["expected"], rel=rel)
    assert e.confidence == pytest.approx(o["confidence"], rel=10 * rel)


@pytest.mark.parametrize("anti", [False, True])
@pytest.mark.parametrize("blocks,first,n", [(1, 0, 1_000_000), (1, 5, 1_000_003), (2, 3, 700_001), (7, (1 << 34) - 1234, 555_555)])
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_vanilla_many_trips_vs_oracle(small, po, X, blocks, first, n, anti):
    eng = small[blocks]
    eng.set_antithetic(anti)
    try:
        e = eng.vanilla(VAN, n, SEED, first, X)
    finally:
        eng.set_antithetic(False)
    _, o = po.dev_vanilla(X, VAN, SEED, first, n, want_paths=False, antithetic=anti)
    check(e, o, X, n)


# n_assets -> the kernel family that prices it: f32 4, 12 basket_f32_kernel (SGPR / LDS-staged constants),
# 16 basket_tiled_f32_kernel, 40 basket_dyn_f32_kernel; f64 4 basket_kernel, 16 basket_tiled_kernel, 40 basket_dyn_kernel
@pytest.mark.parametrize("anti", [False, True])
@pytest.mark.parametrize("blocks,first,n", [(1, 0, 300_000), (2, 7, 300_001), (7, (1 << 32) - 100_000, 250_007)])
@pytest.mark.parametrize("n_assets", [4, 12, 16, 40])
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_basket_many_trips_vs_oracle(mc, small, po, X, n_assets, blocks, first, n, anti):
    if n_assets == 40:
        n //= 4   # keeps the CPU oracle's share of the test short
    b = basket_inputs(mc, n_assets, X)
    eng = small[blocks]
    eng.set_antithetic(anti)
    try:
        e = eng.basket(b, n, SEED, first, X)
    finally:
        eng.set_antithetic(False)
    _, o = po.dev_basket(X, b, SEED, first, n, want_paths=False, antithetic=anti)
    check(e, o, X, n)


@pytest.mark.parametrize("blocks", [1, 2])
@pytest.mark.parametrize("n_assets", [4, 16])
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_basket_control_variate_many_trips_vs_oracle(mc, small, po, X, n_assets, blocks):
    b = basket_inputs(mc, n_assets, X)
    eng = small[blocks]
    n = 200_003
    eng.set_control_variate(True)
    try:
        e = eng.basket(b, n, SEED, 1, X)
    finally:
        eng.set_control_variate(False)
    _, o = po.dev_basket(X, b, SEED, 1, n, want_paths=False, control=True)
    # the simulated quantity is payoff - control: a difference of two nearly equal numbers, so the per-path
    # rounding error is the payoffs' (1e-5 of ~10 in f32) on values of ~0.1: 100 x the relative tolerance
    assert e.sum == pytest.approx(o["sum"], rel=TOL[X]["rel"] * 100)
    assert e.sum2 == pytest.approx(o["sum2"], rel=TOL[X]["rel"] * 100)


@pytest.mark.parametrize("anti", [False, True])
@pytest.mark.parametrize("blocks,n_grid,first,n", [(1, 50, 0, 200_000), (2, 25, 11, 300_001), (7, 75, 3, 100_003)])
@pytest.mark.parametrize("X", ["f32", "f64"])
def test_cva_many_trips_vs_oracle(small, po, X, blocks, n_grid, first, n, anti):
    c = dict(CVA0, n_grid=n_grid)
    eng = small[blocks]
    eng.set_antithetic(anti)
    try:
        e = eng.cva(c, n, SEED, first, X)
    finally:
        eng.set_antithetic(False)
    _, o = po.dev_cva(X, c, SEED, first, n, want_paths=False, antithetic=anti)
    check(e, o, X, n, rel_scale=2.0)


def test_f32_flush_period_is_exercised(small):
    """Guard on the test itself: one workgroup pricing 1e6 fp32 vanilla paths runs > 100 flushes per lane."""
    units = 1_000_000 // 4
    trips = units // (small[1].blocks * 256)
    assert small[1].blocks == 1 and trips // 8 > 100


# ---- the in-kernel final reduction (last workgroup to arrive) ---------------------------------------
def _all_products(mc):
    return [("vanilla", "f32", VAN), ("vanilla", "f64", VAN), ("basket", "f32", basket_inputs(mc, 4, "f32")),
            ("basket", "f64", basket_inputs(mc, 4, "f64")), ("basket", "f32", basket_inputs(mc, 16, "f32")),
            ("basket", "f64", basket_inputs(mc, 16, "f64")), ("basket", "f32", basket_inputs(mc, 40, "f32")),
            ("cva", "f32", dict(CVA0, n_grid=50)), ("cva", "f64", dict(CVA0, n_grid=50))]


def test_fused_finish_equals_two_launch_form_bitwise(mc):
    """One launch per call (the last workgroup to arrive adds the pairs) against the two-launch form
    (finish_kernel): same pairs, same fixed order of additions -> identical bits, whichever workgroup
    happens to be last; repeated calls stay identical (the tickets return to zero).  Ranges: aligned, with
    edge launches (vanilla), straddling 2^32 units (two segments), tiny (one workgroup), > 32 shards."""
    ranges = [(0, 1), (3, 2), (5, 1000), (1, 100_003), (0, 3_000_000), ((1 << 32) - 70_000, 200_001), ((1 << 34) - 3, 40_000_007)]
    with mc.Engine(0) as fused, mc.Engine(0) as two, mc.Engine(0, blocks=5) as small:
        two.set_finish(False)
        for prod, X, inp in _all_products(mc):
            for first, n in ranges:
                if prod != "vanilla" and n > 3_000_000:
                    n = 700_001
                a = getattr(fused, prod)(inp, n, SEED, first, X)
                b = getattr(two, prod)(inp, n, SEED, first, X)
                assert (a.sum, a.sum2, a.n) == (b.sum, b.sum2, b.n), (prod, X, first, n)
                for _ in range(3):
                    again = getattr(fused, prod)(inp, n, SEED, first, X)
                    assert (again.sum, again.sum2) == (a.sum, a.sum2), (prod, X, first, n)
            s1 = getattr(small, prod)(inp, 50_001, SEED, 9, X)
            small.set_finish(False)
            s2 = getattr(small, prod)(inp, 50_001, SEED, 9, X)
            small.set_finish(True)
            assert (s1.sum, s1.sum2) == (s2.sum, s2.sum2)
        gf, gt = fused.vanilla_greeks(VAN, 1_000_003, SEED, 5, "f64"), two.vanilla_greeks(VAN, 1_000_003, SEED, 5, "f64")
        assert [(g.sum, g.sum2) for g in gf] == [(g.sum, g.sum2) for g in gt]
        gf, gt = fused.vanilla_greeks(VAN, 1_000_003, SEED, 5, "f32"), two.vanilla_greeks(VAN, 1_000_003, SEED, 5, "f32")
        assert [(g.sum, g.sum2) for g in gf] == [(g.sum, g.sum2) for g in gt]


def test_fused_finish_under_uneven_load_and_back_to_back_calls(mc, po):
    """Hand-offs between workgroups must not depend on timing or placement (cdna_hip_programming.md Guideline 16:
    test under UNEVEN load with the consumer's caches warm).  400 calls back to back on one context, sizes chosen
    so that the last arriver is sometimes an edge launch, sometimes a workgroup with one trip less than its
    neighbours, while a second context keeps the chip busy on another stream; every triple must equal the
    two-launch form's."""
    torch = pytest.importorskip("torch")
    sizes = [100_000 + 4099 * i + (i % 4) for i in range(100)]
    with mc.Engine(0) as a, mc.Engine(0) as noise, mc.Engine(0) as ref:
        ref.set_finish(False)
        want = [ref.vanilla(VAN, n, SEED, 11 * i, "f32") for i, n in enumerate(sizes)]
        st, st2 = torch.cuda.Stream(), torch.cuda.Stream()
        out = torch.zeros((4 * len(sizes), 3), dtype=torch.float64, device="cuda")
        junk = torch.zeros((4 * len(sizes), 3), dtype=torch.float64, device="cuda")
        opt = a.prepared("vanilla", "f32", VAN)[0]
        b16 = noise.prepared("basket", "f64", basket_inputs(mc, 16, "f64"))
        for rep in range(4):
            for i, n in enumerate(sizes):
                k = rep * len(sizes) + i
                a.launch("vanilla", "f32", opt, SEED, 11 * i, n, out[k].data_ptr(), st.cuda_stream)
                if i % 3 == 0:
                    noise.launch("basket", "f64", b16[0], SEED, 0, 300_000 + 977 * i, junk[k].data_ptr(), st2.cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for rep in range(4):
            for i, w in enumerate(want):
                k = rep * len(sizes) + i
                assert (got[k, 0], got[k, 1], got[k, 2]) == (w.sum, w.sum2, float(w.n)), (rep, i)


def test_one_context_on_two_streams_is_ordered(mc, po):
    """A context owns one pair buffer, one ticket block and one constant table; calls on different streams are
    ordered behind each other by the library (mc_api.hip: begin_call), so alternating streams -- and alternating
    products that share the table (CVA grids, a 20-asset basket) -- gives every call its own data."""
    torch = pytest.importorskip("torch")
    b20 = basket_inputs(mc, 20, "f64", rho=0.3)
    cvas = [dict(CVA0, n_grid=g) for g in (64, 32, 100)]
    with mc.Engine(0) as e, mc.Engine(0) as ref:
        calls = [("cva", "f64", cvas[0]), ("basket", "f64", b20), ("cva", "f64", cvas[1]), ("vanilla", "f32", VAN),
                 ("cva", "f32", cvas[2]), ("basket", "f64", b20)]
        n = 150_001
        want = [getattr(ref, p)(inp, n, SEED, 3, X) for p, X, inp in calls]
        streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
        out = torch.zeros((5 * len(calls), 3), dtype=torch.float64, device="cuda")
        keep = [e.prepared(p, X, inp) for p, X, inp in calls]
        for rep in range(5):
            for i, (p, X, _) in enumerate(calls):
                k = rep * len(calls) + i
                e.launch(p, X, keep[i][0], SEED, 3, n, out[k].data_ptr(), streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for k in range(got.shape[0]):
            w = want[k % len(calls)]
            assert (got[k, 0], got[k, 1]) == (w.sum, w.sum2), k


def test_result_carries_kernel_and_wall_time(mc):
    with mc.Engine(0) as e:
        e.vanilla(VAN, 10 ** 6, SEED, 0, "f32")
        r = e.vanilla(VAN, 10 ** 8, SEED, 0, "f32")
        assert 0.03 < r.kernel_ms < 0.2 and r.kernel_ms < r.wall_ms < 5.0


def test_timing_off_returns_the_same_bits_through_pinned_memory(mc):
    """mc_context_set_timing(ctx, 0): the last workgroup writes the triple straight into pinned host memory and the host
    polls for it (no events, no copy command, no sleeping synchronize).  Same numbers, bit for bit, for every product,
    multi-segment ranges and back-to-back calls; kernel_ms is 0; the call is not slower."""
    with mc.Engine(0) as timed, mc.Engine(0) as direct, mc.Engine(0, blocks=3) as small:
        direct.set_timing(False)
        for prod, X, inp in _all_products(mc):
            for first, n in ((0, 2), (5, 1000), (3, 2_000_003), ((1 << 32) - 70_000, 200_001)):   # (n = 1 has a NaN half-width)
                a, b = getattr(timed, prod)(inp, n, SEED, first, X), getattr(direct, prod)(inp, n, SEED, first, X)
                assert (a.sum, a.sum2, a.n, a.expected, a.confidence) == (b.sum, b.sum2, b.n, b.expected, b.confidence), (prod, X, first, n)
                assert b.kernel_ms == 0.0 and b.wall_ms > 0
        small.set_timing(False)
        first = [small.vanilla(VAN, 100_000 + i, SEED, i, "f32").sum for i in range(200)]      # 200 calls back to back
        small.set_timing(True)
        assert first == [small.vanilla(VAN, 100_000 + i, SEED, i, "f32").sum for i in range(200)]
        for e in (timed, direct):
            e.vanilla(VAN, 10 ** 6, SEED, 0, "f32")
        t = sorted(timed.vanilla(VAN, 10 ** 6, SEED, 0, "f32").wall_ms for _ in range(200))[100]
        d = sorted(direct.vanilla(VAN, 10 ** 6, SEED, 0, "f32").wall_ms for _ in range(200))[100]
        assert d < t * 1.05, (d, t)


def test_mixed_call_sequence_keeps_the_tickets_consistent(mc):
    """Every kind of call shares the context's ticket block (one-dimensional grids, the Greeks' multi-plane and
    two-dimensional grids, multi-segment ranges, XORWOW launches, timed and untimed returns).  A long mixed sequence on
    ONE context must give, call by call, the bits of the two-launch form on another context: a ticket left non-zero or
    a stale pair would show at once."""
    b5 = basket_inputs(mc, 5, "f64", rho=0.3)
    b20 = basket_inputs(mc, 20, "f32", rho=0.3)
    c = dict(CVA0, n_grid=40)

    def sequence(e):
        out = []
        for rep in range(12):
            n = 50_000 + 977 * rep
            out.append(e.vanilla(VAN, n, SEED, rep, "f32").sum)
            out += [g.sum for g in e.vanilla_greeks_lr(VAN, n, SEED, rep, "f64")]
            pr, dl, vg = e.basket_greeks(b5, n // 5, SEED, rep, "f64")
            out += [pr.sum] + [g.sum for g in dl + vg]
            out.append(e.cva(c, n // 8, SEED, (1 << 32) - 1000 * rep - 7, "f64").sum)      # two launches
            e.set_timing(rep % 2 == 0)
            out.append(e.basket(b20, n // 4, SEED, rep, "f32").sum)
            out += [g.sum for g in e.cva_greeks(c, n // 8, SEED, rep, "f32")]
            e.set_generator("xorwow", 7)
            out.append(e.vanilla(VAN, n, SEED, rep, "f64").sum)
            out.append(e.cva(c, n // 8, SEED, rep, "f32").sum)
            e.set_generator("philox")
            e.set_antithetic(rep % 3 == 0)
            out.append(e.vanilla(VAN, n, SEED, 3 * rep + 1, "f64").sum)
            e.set_antithetic(False)
        e.set_timing(True)
        return out
    with mc.Engine(0, blocks=37) as fused, mc.Engine(0, blocks=37) as two:
        two.set_finish(False)
        a, b = sequence(fused), sequence(two)
        assert len(a) == len(b) > 200 and a == b
        assert sequence(fused) == a
