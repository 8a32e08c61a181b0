#!/usr/bin/env python3
"""Wall time of the reference CVA driver's own call -- 131 072 paths (dp/cvaOpt.cu:12-15), grids 25 ... 500 (:70-75) and BASELINE's 256 --
as ONE synchronous mc_cva_run_* (what dev_cvaEquityOption does), timing off (pinned-slot read-back), back to back, median of 300 after
30 warm-ups: one lane per path (mc_context_set_cva_date_lanes(ctx, 1): round 5's only form) against the default rule, which prices a
small call date-parallel (fp64: up to 7/4 wave-trips; csrc/mc_launch_shape.hpp: cva_plan).  Also the same for C5's shard of 8 (1 250 000 paths).

    python tools/cva_call_latency.py > profiles/r06_cva_call_latency.log      # on the GPU box
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import montecarlocuda_amd as mc

eng = mc.Engine(0)
eng.set_timing(False)
print(eng.describe())
print(f"{'paths':>9} {'dates':>6} {'X':>4} | {'one lane/path us':>17} {'default us':>11} {'ratio':>6} | {'workgroups 1 / default':>22} | estimates agree to")
for paths, grids in ((131072, (25, 50, 75, 250, 256, 500)), (1250000, (256,))):
    for X in ("f64", "f32"):
        for n_grid in grids:
            c = dict(bench.CVA, n_grid=n_grid)
            row = {}
            for lanes in (1, 0):
                eng.set_cva_date_lanes(lanes)
                t = []
                for i in range(-30, 300):
                    t0 = time.perf_counter()
                    e = eng.cva(c, paths, mc.MC_DEFAULT_SEED, 0, X)
                    if i >= 0:
                        t.append(time.perf_counter() - t0)
                row[lanes] = (statistics.median(t) * 1e6, eng.last_launch()[0], e.sum)
            rel = abs(row[0][2] - row[1][2]) / abs(row[1][2])
            print(f"{paths:9d} {n_grid:6d} {X:>4} | {row[1][0]:17.1f} {row[0][0]:11.1f} {row[0][0] / row[1][0]:6.3f} | {row[1][1]:10d} / {row[0][1]:<9d} | {rel:.1e}")

print()
print("forced lane counts (kernel time of one call from HIP events, timing on; median of 60): where the date-parallel form wins")
print(f"{'paths':>9} {'dates':>6} {'X':>4} | " + " ".join(f"{'L=' + str(l):>8}" for l in (1, 2, 4, 8, 16, 32)) + " | auto")
eng.set_timing(True)
for X in ("f64", "f32"):
    for n_grid in (25, 256):
        for paths in (4096, 16384, 49152, 65536, 98304, 114688, 131072, 196608, 262144):
            c = dict(bench.CVA, n_grid=n_grid)
            cells = []
            for lanes in (1, 2, 4, 8, 16, 32, 0):
                eng.set_cva_date_lanes(lanes)
                t = sorted(eng.cva(c, paths, mc.MC_DEFAULT_SEED, 0, X).kernel_ms for _ in range(60))
                cells.append(t[len(t) // 2] * 1e3)
            print(f"{paths:9d} {n_grid:6d} {X:>4} | " + " ".join(f"{v:8.1f}" for v in cells[:-1]) + f" | {cells[-1]:.1f}")
