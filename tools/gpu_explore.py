#!/usr/bin/env python3
"""First-contact script for the GPU box: parity deltas vs the oracle and raw kernel timings.

    gpurun -- python tools/gpu_explore.py > gpurun_out/explore.log
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=256)
SEED = mc.MC_DEFAULT_SEED


def basket(n, X):
    v = [0.3 if i % 2 == 0 else 0.2 for i in range(n)]
    corr = np.full((n, n), 0.5) + 0.5 * np.eye(n)
    L, bad = mc.chol(corr, X)
    assert bad == 0
    return dict(s=[100.0] * n, v=v, p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n, k=100.0, t=1.0, r=0.048790164)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def main():
    eng = mc.Engine(0)
    print(json.dumps(eng.info()))
    for X in ("f32", "f64"):
        npb = po.NPB[X]
        # normals
        g = eng.normals(SEED, 1, 5, 4096, 0, X)
        o = np.array([po.dev_normals(X, SEED, 1, 5 + u, 0) for u in range(4096)])
        print(X, "normals max abs diff", np.abs(g.astype(np.float64) - o).max(), "max|z|", np.abs(o).max())
        # vanilla per path
        n = 20000
        gp = eng.vanilla_paths(VAN, n, SEED, 3, X)
        op, orr = po.dev_vanilla(X, VAN, SEED, 3, n)
        print(X, "vanilla per-path max abs diff", np.abs(gp.astype(np.float64) - op).max(),
              "mean", np.abs(gp.astype(np.float64) - op).mean())
        e = eng.vanilla(VAN, n, SEED, 3, X)
        print(X, "vanilla sums rel", rel(e.sum, orr["sum"]), rel(e.sum2, orr["sum2"]), "E", e.expected, orr["expected"],
              "CI", e.confidence, orr["confidence"])
        for nb in (3, 4, 16):
            b = basket(nb, X)
            n = 5000
            gp = eng.basket_paths(b, n, SEED, 7, X)
            op, orr = po.dev_basket(X, b, SEED, 7, n)
            e = eng.basket(b, n, SEED, 7, X)
            print(X, f"basket n={nb} per-path max abs diff", np.abs(gp.astype(np.float64) - op).max(), "sums rel",
                  rel(e.sum, orr["sum"]), rel(e.sum2, orr["sum2"]))
        for ng in (25, 250, 256):
            c = dict(CVA, n_grid=ng)
            n = 2000
            gp = eng.cva_paths(c, n, SEED, 11, X)
            op, orr = po.dev_cva(X, c, SEED, 11, n)
            e = eng.cva(c, n, SEED, 11, X)
            d = np.abs(gp.astype(np.float64) - op)
            print(X, f"cva n_grid={ng} per-path max abs diff", d.max(), "max rel", (d / np.maximum(np.abs(op), 1e-12)).max(),
                  "sums rel", rel(e.sum, orr["sum"]), rel(e.sum2, orr["sum2"]), "E", e.expected, orr["expected"])
    # timings
    print("--- timings (kernel_ms from HIP events; second call of each) ---")
    def t(label, fn, n, unit="paths"):
        fn()
        best = min(fn().kernel_ms for _ in range(3))
        print(f"{label:40s} {best:10.3f} ms   {n / best * 1e3:.4g} {unit}/s")
    t("vanilla f32 1e8", lambda: eng.vanilla(VAN, 10**8, SEED, 0, "f32"), 1e8)
    t("vanilla f64 1e8", lambda: eng.vanilla(VAN, 10**8, SEED, 0, "f64"), 1e8)
    t("basket n=4 f32 1e8", lambda: eng.basket(basket(4, "f32"), 10**8, SEED, 0, "f32"), 1e8)
    t("basket n=16 f32 1e7", lambda: eng.basket(basket(16, "f32"), 10**7, SEED, 0, "f32"), 1e7)
    t("basket n=4 f64 1e7", lambda: eng.basket(basket(4, "f64"), 10**7, SEED, 0, "f64"), 1e7)
    t("basket n=16 f64 1e7", lambda: eng.basket(basket(16, "f64"), 10**7, SEED, 0, "f64"), 1e7)
    t("cva 256 f32 1e6", lambda: eng.cva(CVA, 10**6, SEED, 0, "f32"), 256e6, "path-steps")
    t("cva 256 f64 1e6", lambda: eng.cva(CVA, 10**6, SEED, 0, "f64"), 256e6, "path-steps")
    e = eng.vanilla(VAN, 10**8, SEED, 0, "f32")
    bs = 10.386270784322328
    print("vanilla f32 1e8: E", e.expected, "CI", e.confidence, "|E-BS|", abs(e.expected - bs))
    e = eng.vanilla(VAN, 10**8, SEED, 0, "f64")
    print("vanilla f64 1e8: E", e.expected, "CI", e.confidence, "|E-BS|", abs(e.expected - bs))
    t0 = time.time()
    e = eng.vanilla(VAN, 10**10, SEED, 0, "f32")
    print("vanilla f32 1e10: E", e.expected, "CI", e.confidence, "|E-BS|", abs(e.expected - bs), "kernel_ms", e.kernel_ms,
          "wall", time.time() - t0)


if __name__ == "__main__":
    main()
