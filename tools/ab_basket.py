#!/usr/bin/env python3
"""A/B of basket-kernel build variants IN ONE PROCESS (interleaved rounds) over basket sizes and both
precisions.  Variants are separate .so builds of mc_api.hip (tools/ab_*.so, built by
tools/build_ab.sh): constants in SGPRs from the kernel arguments vs staged in LDS."""
import ctypes as C, glob, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc
from montecarlocuda_amd import _lib
import bench

variants = sorted(glob.glob(os.path.join(ROOT, "tools", "ab_*.so")))
cases = [("f32", n, int(2.4e9 / (n + 4))) for n in ()] + \
        [("f64", n, int(6e8 / (n + 2))) for n in (4, 8, 16)]
engines = {}
for v in variants:
    L = _lib._declare(C.CDLL(v))
    _lib._LIB = L
    engines[os.path.basename(v)] = (L, mc.Engine(0))
res = {c: {v: [] for v in engines} for c in cases}
ref = {}
for rnd in range(6):
    for c in cases:
        X, n, paths = c
        inputs = bench.basket_inputs(mc, n, X)
        for v, (L, e) in engines.items():
            _lib._LIB = L
            r = e.basket(inputs, paths, mc.MC_DEFAULT_SEED, 0, X)
            ref.setdefault(c, r.expected)
            assert abs(r.expected - ref[c]) <= 1e-6 * abs(ref[c]), (c, v, r.expected, ref[c])
            if rnd >= 2:
                res[c][v].append(r.kernel_ms)
for c in cases:
    base = None
    for v in engines:
        med = statistics.median(res[c][v])
        base = base or med
        print(f"basket n={c[1]:2d} {c[0]} {v:22s} median {med:8.3f} ms  {c[2] / med / 1e6:8.2f} Gpaths/s  ({med / base:.3f}x of first)")
