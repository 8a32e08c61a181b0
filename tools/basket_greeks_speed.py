#!/usr/bin/env python3
"""Wall and kernel time of the basket Greeks calls (pathwise and likelihood ratio) against the pricing call, n = 4, 8, 16, 32, 64:
one pass over the paths per 8 assets (round 5: one pass per asset).  python tools/basket_greeks_speed.py > profiles/..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import montecarlocuda_amd as mc

eng = mc.Engine(0)
print(f"{'n':>3} {'X':>4} {'paths':>10} {'price ms':>10} {'greeks ms':>10} {'LR ms':>10} {'greeks / price':>15} {'passes':>7}")
for X in ("f32", "f64"):
    for n in (4, 8, 16, 32, 64):
        b = bench.basket_inputs(mc, n, X)
        paths = 2 * 10 ** 7 if n <= 16 else 5 * 10 ** 6
        for _ in range(2):
            p = eng.basket(b, paths, mc.MC_DEFAULT_SEED, 0, X)
            g = eng.basket_greeks(b, paths, mc.MC_DEFAULT_SEED, 0, X)
            l = eng.basket_greeks(b, paths, mc.MC_DEFAULT_SEED, 0, X, lr=True)
        print(f"{n:3d} {X:>4} {paths:10d} {p.kernel_ms:10.3f} {g[0].kernel_ms:10.3f} {l[0].kernel_ms:10.3f} {g[0].kernel_ms / p.kernel_ms:15.2f} {(n + 7) // 8:7d}")
