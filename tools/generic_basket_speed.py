#!/usr/bin/env python3
"""Kernel-time throughput of the basket kernels around the static/generic boundary (n = 16 static, 17+ generic)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first, see tests/conftest.py)
import montecarlocuda_amd as mc
import bench
e = mc.Engine(0)
for X in [x for x in ("f32", "f64") if x in os.environ.get("PRECISIONS", "f32 f64")]:
    for n in [int(a) for a in sys.argv[1:]] or (16, 17, 20, 24, 32, 48, 64):
        inputs = bench.basket_inputs(mc, n, X)
        paths = int((4e9 if X == "f32" else 1.2e9) / (n * n / 8 + 4 * n))
        t = []
        for rep in range(5):
            r = e.basket(inputs, paths, mc.MC_DEFAULT_SEED, 0, X)
            if rep:
                t.append(r.kernel_ms)
        ms = statistics.median(t)
        print(f"basket n={n:2d} {X}: {paths / ms / 1e6:8.2f} Gpaths/s  ({ms:.3f} ms for {paths} paths)  price {r.expected:.5f} +- {r.confidence:.5f}")
