#!/usr/bin/env python3
"""kernel_ms vs path count for one workload: fits T = a + n / rate (fixed cost per call and asymptotic rate)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, montecarlocuda_amd as mc
name = sys.argv[1] if len(sys.argv) > 1 else "vanilla_f32"
eng = mc.Engine(0)
prod, X, inputs, _, _, _ = bench.workloads(mc)[name]
if callable(inputs): inputs = inputs()
sizes = [int(float(x)) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1e6,1e7,2e7,5e7,1e8,2e8,5e8,1e9".split(","))]
ts = []
for n in sizes:
    run = lambda: getattr(eng, prod)(inputs, n, mc.MC_DEFAULT_SEED, 0, X).kernel_ms
    run(); t = min(run() for _ in range(7)); ts.append(t)
    print(f"{name} n={n:.3g} kernel_ms={t:.4f} ({n/t*1e3:.4g}/s)")
A = np.vstack([np.ones(len(sizes)), np.array(sizes, dtype=float)]).T
a, b = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
print(f"fit: fixed = {a*1e3:.2f} us per call, rate = {1/b*1e3:.4g} paths/s")
