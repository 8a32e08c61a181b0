#!/usr/bin/env python3
"""Per-path CVA values of every tools/ab_*.so variant against the oracle (fp64, several grids): max |difference|.
Used next to tools/ab_f64.py when a variant changes the arithmetic of the exposure (bound: 1e-13, tests/test_gpu_parity.py)."""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc
from montecarlocuda_amd import _lib
from oracle import pyoracle as po
po.build()
SEED = 0x4D435F4D49333535
for v in sorted(glob.glob(os.path.join(ROOT, "tools", "ab_*.so"))):
    _lib._LIB = _lib._declare(C.CDLL(v))
    e = mc.Engine(0)
    worst = 0.0
    for n_grid, s, k, vol, t in ((256, 100.0, 100.0, 0.2, 1.0), (250, 100.0, 100.0, 0.2, 1.0), (64, 100.0, 60.0, 0.4, 2.0), (64, 100.0, 170.0, 0.1, 0.25),
                                 (7, 20.0, 20.0, 0.5, 3.0), (2, 100.0, 100.0, 0.2, 1.0)):
        c = dict(s=s, k=k, r=0.05, v=vol, t=t, defint=0.03, lgd=0.6, n_grid=n_grid)
        for anti in (False, True):
            e.set_antithetic(anti)
            got = e.cva_paths(c, 3000, SEED, 11, "f64")
            want, _ = po.dev_cva("f64", c, SEED, 11, 3000, antithetic=anti)
            worst = max(worst, float(np.abs(got - want).max()) / (s / 100.0))
    print(f"{os.path.basename(v):28s} max |cva_gpu - cva_oracle| (per 100 of spot) = {worst:.3e}")
    e.close()
