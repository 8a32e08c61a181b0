#!/usr/bin/env python3
"""Soak of the in-kernel final reduction (mc_reduce.hpp: write-through pairs, sharded tickets, one acquire by the last
arriver): hundreds of thousands of short calls, sizes cycling so that the grid, the last arriver and the number of
ticket shards in play keep changing, a second context keeping the chip unevenly busy on another stream -- every
result compared bit for bit with the value the two-launch form gave for that size.  A stale pair or a lost ticket
shows as a mismatch (or a hang: run under a timeout).
    python tools/soak_tail.py [calls]"""
import os
import sys
import time

import torch  # before libmc_mi355x.so: torch must find its own HIP runtime first

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlocuda_amd as mc  # noqa: E402

VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=16)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
sizes = [4 * (1 + 37 * i) + (i % 4) for i in range(1, 400)] + [100_000 + 4099 * i for i in range(60)] + [2_000_000, 5_000_003]
seed = mc.MC_DEFAULT_SEED
with mc.Engine(0) as ref, mc.Engine(0) as fused, mc.Engine(0) as noise:
    ref.set_finish(False)
    want = {n: ref.vanilla(VAN, n, seed, 3 * n, "f32") for n in sizes}
    want_cva = {n: ref.cva(CVA, max(1, n // 16), seed, n, "f64") for n in sizes[:50]}
    fused.set_timing(False)            # the short way back: result polled from pinned memory
    side = torch.cuda.Stream()
    junk = torch.zeros(3, dtype=torch.float64, device="cuda")
    nstruct = noise.prepared("vanilla", "f64", VAN)[0]
    bad = 0
    t0 = time.time()
    for i in range(calls):
        n = sizes[(i * 7919) % len(sizes)]
        if i % 50 == 0:                # uneven background load: a long fp64 launch on another context and stream
            noise.launch("vanilla", "f64", nstruct, seed, 0, 3_000_000 + (i % 7) * 1_000_003, junk.data_ptr(), side.cuda_stream)
        if i % 11 == 0 and n in want_cva:
            got, w = fused.cva(CVA, max(1, n // 16), seed, n, "f64"), want_cva[n]
        else:
            got, w = fused.vanilla(VAN, n, seed, 3 * n, "f32"), want[n]
        if (got.sum, got.sum2, got.n) != (w.sum, w.sum2, w.n):
            bad += 1
            if bad < 10:
                print("MISMATCH call", i, "n", n, got.sum, w.sum)
        if i % 100_000 == 0 and i:
            print(f"{i} calls, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
    torch.cuda.synchronize()
    print(f"{calls} calls ({len(sizes)} sizes, grids of 1 ... 2048 workgroups), {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
