#!/usr/bin/env python3
"""A/B of build variants of the engine IN ONE PROCESS on one device (interleaved rounds), as the CDNA
guide asks: timings from different gpurun boxes differ by several per cent.  Variants are separate
.so builds of mc_api.hip (tools/ab_*.so, see the hipcc lines in DESIGN.md / git log)."""
import ctypes as C, glob, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc
from montecarlocuda_amd import _lib, engine
import bench

variants = sorted(glob.glob(os.path.join(ROOT, "tools", "ab_*.so")))
work = {"vanilla_f64": 4 * 10 ** 8, "basket16_f64": 3 * 10 ** 7, "cva256_f64": 10 ** 6, "basket4_f64": 10 ** 8,
        "vanilla_f32": 2 * 10 ** 9, "basket4_f32": 5 * 10 ** 8, "cva256_f32": 4 * 10 ** 6}
if len(sys.argv) > 1:
    work = {k: v for k, v in work.items() if k in sys.argv[1:]}
W = bench.workloads(mc)
W["basket4_f64"] = ("basket", "f64", lambda: bench.basket_inputs(mc, 4, "f64"), 0, 0, "")
engines = {}
for v in variants:
    L = _lib._declare(C.CDLL(v))
    _lib._LIB = L                      # Engine() binds whatever lib() returns at construction
    engines[os.path.basename(v)] = (L, mc.Engine(0))
res = {k: {v: [] for v in engines} for k in work}
for rnd in range(7):
    for name, n in work.items():
        prod, X, inputs, *_ = W[name]
        if callable(inputs): inputs = inputs()
        for v, (L, e) in engines.items():
            _lib._LIB = L
            t = getattr(e, prod)(inputs, n, mc.MC_DEFAULT_SEED, 0, X).kernel_ms
            if rnd >= 2:
                res[name][v].append(t)
for name in work:
    base = None
    for v in engines:
        med = statistics.median(res[name][v])
        base = base or med
        print(f"{name:14s} {v:24s} median {med:8.3f} ms  min {min(res[name][v]):8.3f}  ({med/base:.3f}x of first)")
