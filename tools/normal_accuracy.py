#!/usr/bin/env python3
"""Max |z_gpu - z_oracle| of the generator's normals (oracle = glibc libm on the same Philox words), and the
per-path payoff error of the fp64 vanilla kernel.  Needs the GPU; the oracle is the checker here."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc
from oracle import pyoracle as po
po.build()
e = mc.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
for X in ("f64", "f32"):
    g = e.normals(mc.MC_DEFAULT_SEED, 1, 0, n, 0, X)
    ref = np.stack([po.dev_normals(X, mc.MC_DEFAULT_SEED, 1, u, 0) for u in range(n)])
    err = np.abs(g.astype(np.float64) - ref.astype(np.float64))
    print(f"{X}: {g.size} normals, max |dz| = {err.max():.3e}, mean = {err.mean():.3e}, max |z| = {np.abs(ref).max():.3f}")
