"""The N>1 path on CPU: 2 ranks under gloo shard the path range, all-reduce the fp64 triple and
close the estimator.  The per-shard engine here is the oracle (test infrastructure); on the GPU
box the same ``sharded_estimate`` is fed by the HIP engine (bench.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as tmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
TOTAL = 40001  # odd on purpose: shards are unequal and not multiples of the Philox block
SEED = 0x4D435F4D49333535


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, X, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import math
    from montecarlocuda_amd import distributed as D
    from oracle import pyoracle as po
    D.init_from_env("gloo")

    def compute(first, n, out):
        _, r = po.dev_vanilla(X, VAN, SEED, first, n, want_paths=False)
        out.copy_(torch.tensor([r["sum"], r["sum2"], float(r["n"])], dtype=torch.float64))

    import numpy as np
    # discount from the inputs as the engine of that precision sees them (f32: r, t rounded to float)
    r, t = (float(np.dtype(po.NP[X]).type(VAN[k])) for k in ("r", "t"))
    est = D.sharded_estimate(compute, TOTAL, math.exp(-r * t))
    q.put((rank, D.my_shard(TOTAL), est.expected, est.confidence, est.sum, est.sum2, est.n))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_two_rank_sharded_estimate_equals_single_rank(po, X):
    import math
    world = 2
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, X, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards are contiguous, disjoint and cover [0, TOTAL)
    (f0, c0), (f1, c1) = got[0][1], got[1][1]
    assert f0 == 0 and f1 == c0 and c0 + c1 == TOTAL
    # every rank holds the same reduced estimate
    assert got[0][2:] == got[1][2:]
    # and it equals the single-process estimate over the whole range (same seed, same stream)
    _, whole = po.dev_vanilla(X, VAN, SEED, 0, TOTAL, want_paths=False)
    _, e, ci, s, s2, n = got[0][1:2] + got[0][2:]
    assert n == TOTAL
    assert s == pytest.approx(whole["sum"], rel=1e-13) and s2 == pytest.approx(whole["sum2"], rel=1e-13)
    assert e == pytest.approx(whole["expected"], rel=1e-13) and ci == pytest.approx(whole["confidence"], rel=1e-12)


def test_eight_rank_sharded_estimate_equals_single_rank(po):
    """The node's own world size (8 ranks, gloo on the CPU): contiguous disjoint shards of an odd total that cover it, one reduced
    estimate on every rank, equal to the single-process estimate over the whole range -- the bookkeeping of the 8-GPU run, which no
    box here can execute on devices."""
    X, world = "f64", 8
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, X, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    nxt = 0
    for rank, (first, count), *_ in got:
        assert first == nxt and count in (TOTAL // 8, TOTAL // 8 + 1)
        nxt = first + count
    assert nxt == TOTAL and [g[0] for g in got] == list(range(8))
    assert all(g[2:] == got[0][2:] for g in got)
    _, whole = po.dev_vanilla(X, VAN, SEED, 0, TOTAL, want_paths=False)
    e, ci, s, s2, n = got[0][2:]
    assert n == TOTAL and s == pytest.approx(whole["sum"], rel=1e-13) and s2 == pytest.approx(whole["sum2"], rel=1e-13)
    assert e == pytest.approx(whole["expected"], rel=1e-13) and ci == pytest.approx(whole["confidence"], rel=1e-12)
