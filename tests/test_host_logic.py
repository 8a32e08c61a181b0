"""Host-side logic of the product that needs no GPU: sharding, closing formulas, Cholesky."""
import numpy as np
import pytest

from conftest import fromhex, load_golden


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    return mc


def test_shard_range_partitions_exactly(mc):
    for total in (0, 1, 7, 10**8, 10**9 + 7, 2**52 - 1, 2**63 + 12345):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for rank in range(world):
                first, count = mc.shard_range(total, rank, world)
                assert first == pos == (rank * total) // world   # SURVEY 8e rule
                pos += count
            assert pos == total


def test_closing_matches_oracle_and_reference_formula(mc, po):
    for s, s2, n, disc in ((1095721.8224204443, 36197748.012070745, 100000, 0.9523),
                           (368.1204752756539, 106.37671195984049, 2000, 1.0)):
        assert mc.closing(s, s2, n, disc) == po.closing(s, s2, n, disc)
    # against the compiled reference through the golden CVA case (closing is the last step of it)
    c = [c for c in load_golden("ref_mc.json")["cases"] if c["kind"] == "cva" and c["X"] == "f64"][1]
    r = po.host_cva("f64", c["cva"], c["paths"], c["seed"])
    e, ci = mc.closing(r["sum"], r["sum2"], r["n"], 1.0)
    assert e == fromhex(c["expected"]) and ci == fromhex(c["confidence"])


def test_chol_matches_reference_golden(mc):
    for c in load_golden("ref_chol.json")["cases"]:
        m = [[fromhex(x) for x in row] for row in c["c"]]
        want = np.array([[fromhex(x) for x in row] for row in c["a"]])
        got, bad = mc.chol(m, c["X"])
        assert (got.astype(np.float64) == want).all(), (c["X"], c["n"], c["name"])
        zero_cols = int((np.diag(want) == 0).sum())
        assert bad == zero_cols


def test_chol_flags_indefinite_input(mc):
    # the reference driver's N=4 pattern is not positive definite
    m = [[1.0 if i == j else (0.5 if max(i, j) % 2 == 0 else -0.5) for j in range(4)] for i in range(4)]
    _, bad = mc.chol(m)
    assert bad > 0
    _, bad = mc.chol(np.full((4, 4), 0.5) + 0.5 * np.eye(4))
    assert bad == 0
