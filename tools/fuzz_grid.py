#!/usr/bin/env python3
"""Randomised check of the launch-geometry compatibility mode (mc_*_run_grid_*) on the GPU box: random geometries
(blocks, threads, paths per block -- threads that price nothing, T not dividing N_PATH, single-thread blocks), products,
precisions and market data.  For every case:
  * the call's (sum, sum2, n) equal BITWISE those of the from-normals hook fed the arrangement the reference's loops imply,
    rebuilt on the host from the per-thread streams (mc_grid_normals);
  * the per-path values of that arrangement against the oracle's device formulas at the bounds of tests/test_gpu_parity.py;
  * a sample of the per-thread streams against the oracle's restatement (orc_grid_normals, bit-equal to rocRAND's host
    engine) at 4e-6.
    python tools/fuzz_grid.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (the checker)

po.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
NP = {"f32": np.float32, "f64": np.float64}
TOL = {"f32": dict(pay=2e-6, cva=2e-5), "f64": dict(pay=1e-14, cva=1e-13)}
eng = mc.Engine(0)
worst, bad = {}, 0


def note(key, err, tol, what):
    global bad
    worst[key] = max(worst.get(key, 0.0), err / tol)
    if not err <= tol:
        bad += 1
        print("VIOLATION", key, what, f"err {err:.3e} > tol {tol:.3e}")


def cva_draws(t, n_grid, X):
    R = NP[X]
    t = R(t)
    dt = R(t / R(n_grid))
    d = 0
    for _ in range(n_grid):
        t = R(t - dt)
        if not t >= 0:
            break
        d += 1
    return d


for it in range(cases):
    X = "f32" if rng.random() < 0.5 else "f64"
    G = int(rng.choice([1, 2, 3, 5, 8, 13, 32, 40]))
    T = int(rng.choice([1, 2, 3, 7, 32, 64, 100, 128, 256, 1000, 1024]))
    per_block = int(rng.integers(1, 400))
    if rng.random() < 0.2:      # long threads: the fused kernels then cut a thread's stream into several pieces (grid_pieces)
        per_block = min(int(rng.integers(400, 40 * T + 2000)), 150000 // G)
    prod = str(rng.choice(["vanilla", "basket", "basket", "cva"]))
    spot = float(np.exp(rng.uniform(np.log(5), np.log(500))))
    r, t = float(rng.uniform(0.0, 0.08)), float(rng.choice([0.25, 0.5, 1.0, 2.0, 3.0]))
    what = (prod, X, G, T, per_block)
    if prod == "vanilla":
        inp = dict(s=spot, k=spot * float(rng.uniform(0.6, 1.5)), r=r, v=float(rng.uniform(0.05, 0.6)), t=t)
        draws = row = 1
    elif prod == "basket":
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 24, 33, 40]))
        rho = float(rng.uniform(-0.9 / max(n - 1, 1), 0.9)) if n > 1 else 0.0
        L, nbad = mc.chol(np.full((n, n), rho) + (1 - rho) * np.eye(n), X)
        if nbad:
            continue
        w = rng.uniform(0.5, 1.5, n)
        inp = dict(s=(spot * rng.uniform(0.7, 1.3, n)).tolist(), v=rng.uniform(0.05, 0.5, n).tolist(), p=L.tolist(),
                   d=rng.uniform(-0.02, 0.02, n).tolist(), w=(w / w.sum()).tolist(), k=spot * float(rng.uniform(0.7, 1.3)), t=t, r=r)
        draws = row = n
        what += (n,)
    else:
        n_grid = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 12, 25, 50, 64]))
        inp = dict(s=spot, k=spot * float(rng.uniform(0.7, 1.3)), r=r, v=float(rng.uniform(0.1, 0.5)), t=t, defint=float(rng.uniform(0.0, 0.1)),
                   lgd=float(rng.uniform(0.2, 1.0)), n_grid=n_grid)
        draws, row = cva_draws(t, n_grid, X), n_grid
        what += (n_grid, draws)
    streams = eng.grid_normals(G, T, po.grid_draws_per_thread(T, per_block, draws))
    z = np.zeros((G * per_block, row), dtype=NP[X])
    z[:, :draws] = po.grid_path_normals(streams, per_block, draws)
    # staged form (round 3: normals through HBM) = the checker: bitwise against the from-normals hook below; fused form
    # (round 4: the reference's launch itself) = per-path values bit-equal to the staged form's, sums within the order of additions
    eng.set_grid_form("staged")
    e = eng.run_grid(prod, inp, G, T, per_block, X)
    sv = eng.paths_grid(prod, inp, G, T, per_block, X)
    eng.set_grid_form("auto")
    f = eng.run_grid(prod, inp, G, T, per_block, X)
    fv = eng.paths_grid(prod, inp, G, T, per_block, X)
    U = np.uint32 if X == "f32" else np.uint64
    if not np.array_equal(sv.view(U), fv.view(U)):
        bad += 1
        print("VIOLATION fused per-path values differ from the staged form's", what, int((sv.view(U) != fv.view(U)).sum()))
    rel = 3e-6 if X == "f32" else 1e-12
    if f.n != e.n or abs(f.sum - e.sum) > rel * abs(e.sum) or abs(f.sum2 - e.sum2) > rel * abs(e.sum2):
        bad += 1
        print("VIOLATION fused sums differ from the staged form's", what, (f.sum, f.sum2, f.n), (e.sum, e.sum2, e.n))
    if prod == "vanilla":
        h, vals = eng.vanilla_from_normals(inp, z.reshape(-1), X)
        want, _ = po.dev_vanilla_on_normals(X, inp, z.reshape(-1))
        tol = TOL[X]["pay"] * spot * float(np.exp(r * t + 4 * inp["v"] * np.sqrt(t))) * 3
    elif prod == "basket":
        h, vals = eng.basket_from_normals(inp, z, X)
        want, _ = po.dev_basket_on_normals(X, inp, z, 0)
        tol = TOL[X]["pay"] * spot * 1.3 * float(np.exp(r * t + 4 * max(inp["v"]) * np.sqrt(t))) * 4
    else:
        h, vals = eng.cva_from_normals(inp, z, X)
        want, _ = po.dev_cva_on_normals(X, inp, z.astype(np.float64), 0)
        tol = TOL[X]["cva"] * spot * float(np.exp(r * t + 4 * inp["v"] * np.sqrt(t))) / 100.0 * 3   # as tools/fuzz_parity.py: the reachable spot
    if (e.sum, e.sum2, e.n) != (h.sum, h.sum2, G * per_block):
        bad += 1
        print("VIOLATION sums differ from the arrangement's", what, (e.sum, e.sum2, e.n), (h.sum, h.sum2, h.n))
    note((prod, X), float(np.abs(vals.astype(np.float64) - want).max()), tol, what)
    # a sample of streams against the oracle
    cnt = min(streams.shape[2], 6)
    ws = po.grid_normals(G, min(T, 4), cnt)
    note(("streams", "f32"), float(np.abs(streams[:, :min(T, 4), :cnt].astype(np.float64) - ws).max()), 4e-6, what)

print(f"{cases} cases, {bad} violations; worst error / bound per (product, precision):")
for k in sorted(worst):
    print(f"  {k}: {worst[k]:.3f}")
eng.close()
sys.exit(1 if bad else 0)
