#!/usr/bin/env python3
"""ONE big pricing call, as a drop-in caller sees it (dp/vanillaOpt.cu:77-83 times exactly that): 1e8 fp32 vanilla paths, first launch ->
result on the host, timing off (pinned-slot read-back), back to back, median of 200.  A/B of VERDICT r05 #7: the call's range as ONE
launch against TWO launches of unequal size on two streams (two contexts here: same device-side picture as two launches sharing one
ticket set, no event between the streams -- the host adds the two triples), so that the second launch's ramp fills the first one's
tail.  The stepped region of bench.py hides ramp and tail by overlapping SUCCESSIVE calls; a lone call has nothing to overlap with.

    python tools/single_call_ab.py > profiles/r06_single_call_ab.log      # on the GPU box
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import montecarlocuda_amd as mc

N = 10 ** 8
a, b = mc.Engine(0), mc.Engine(0)
for e in (a, b):
    e.set_timing(False)
sa, sb = a.prepared("vanilla", "f32", bench.VAN)[0], b.prepared("vanilla", "f32", bench.VAN)[0]
out = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
seed = mc.MC_DEFAULT_SEED


def one():
    return a.vanilla(bench.VAN, N, seed, 0, "f32").sum


def two(first_share):
    n1 = int(N * first_share) // 4 * 4
    s1, s2 = a.arm_direct(), b.arm_direct()
    a.launch("vanilla", "f32", sa, seed, 0, n1, out[0].data_ptr(), a.stream)
    b.launch("vanilla", "f32", sb, seed, n1, N - n1, out[1].data_ptr(), b.stream)
    r1, r2 = a.wait_slot(s1), b.wait_slot(s2)
    return r1[0] + r2[0]


def timed(fn, reps=200):
    for _ in range(30):
        v = fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        v = fn()
        t.append(time.perf_counter() - t0)
    return statistics.median(t) * 1e6, min(t) * 1e6, v


print(f"{'form':44s} {'median us':>10} {'min us':>8} {'paths/s':>11}   sum")
ref = None
for rnd in range(2):
    for label, fn in (("one launch (mc_vanilla_run_f32)", one), ("two launches, 50 / 50, two streams", lambda: two(0.5)), ("two launches, 60 / 40", lambda: two(0.6)),
                      ("two launches, 70 / 30", lambda: two(0.7)), ("two launches, 85 / 15", lambda: two(0.85))):
        med, mn, v = timed(fn)
        ref = v if ref is None else ref
        print(f"{label:44s} {med:10.2f} {mn:8.2f} {N / med * 1e6:11.4g}   {v:.9g}  (rel. to one launch {abs(v - ref) / ref:.1e})")
