#!/usr/bin/env python3
"""kernel_ms of every bench workload at its BASELINE size for grids of 4 ... 16 workgroups per CU, rounds interleaved in
one process (one engine per grid size): is the default of 8 per CU still right for kernels whose register use admits
fewer than 8 resident workgroups?   python tools/grid_sweep_all.py [workload ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, montecarlocuda_amd as mc
W = bench.workloads(mc)
# extra rows: basket<n>_<X> for any n (BASELINE's recipe), e.g. basket8_f64
import re
for a in sys.argv[1:]:
    m = re.fullmatch(r"basket(\d+)_(f32|f64)", a)
    if m and a not in W:
        n_, X_ = int(m.group(1)), m.group(2)
        W[a] = ("basket", X_, (lambda n_=n_, X_=X_: bench.basket_inputs(mc, n_, X_)), int(2e9 / (n_ * n_ + 12.5 * n_ + 6) / (1 if X_ == "f32" else 3)), 0, "")
names = sys.argv[1:] or ["vanilla_f32", "vanilla_f64", "basket4_f32", "basket16_f32", "basket16_f64", "cva256_f64", "cva256_f32"]
grids = [int(x) for x in os.environ.get("GRIDS", "1024,1280,1536,1792,2048,2560,3072,4096").split(",")]
engines = {g: mc.Engine(0, blocks=g) for g in grids}   # NOTE: blocks = the 1x grid; heavier kernels launch a multiple of it (mc_api.hip grid_for)
for name in names:
    prod, X, inputs, n, _, _ = W[name]
    if callable(inputs): inputs = inputs()
    res = {g: [] for g in grids}
    for rnd in range(9):
        for g in grids:
            t = getattr(engines[g], prod)(inputs, n, mc.MC_DEFAULT_SEED, 0, X).kernel_ms
            if rnd >= 2: res[g].append(t)
    base = statistics.median(res[2048 if 2048 in res else grids[0]])
    print(name, " ".join(f"{g}:{statistics.median(res[g]) * 1e3:.1f}us({statistics.median(res[g]) / base:.3f})" for g in grids), flush=True)
