#!/usr/bin/env python3
"""kernel_ms of one workload for several grid sizes (workgroups): Engine(blocks=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, montecarlocuda_amd as mc
name = sys.argv[1] if len(sys.argv) > 1 else "vanilla_f32"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 8
prod, X, inputs, _, _, _ = bench.workloads(mc)[name]
if callable(inputs): inputs = inputs()
for blocks in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192, 16384):
    with mc.Engine(0, blocks=blocks) as e:
        run = lambda: getattr(e, prod)(inputs, n, mc.MC_DEFAULT_SEED, 0, X).kernel_ms
        run(); run()
        t = sorted(run() for _ in range(9))[2]
        print(f"{name} n={n:.3g} blocks={blocks:6d} kernel_ms={t:.4f} rate={n/t*1e3:.4g}/s")
