"""Where a process's first pricing call spends its time: HIP runtime start-up (hipInit through mc_device_count), mc_context_create
(stream, events, pinned slots, scratch in HBM), the first call of each kernel family (code-object load), a repeated call.
Measured on the GPU box, second process of a pair (warm file cache): hipInit 110 ms, context 27 ms, first call 10 ms,
repeat 36 us -- the reference allocates, seeds 65 536 XORWOW states and frees on EVERY call (dp/MonteCarloKernel.cu:296-363).
    python tools/context_time.py"""
import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t0 = time.perf_counter()
import montecarlocuda_amd as mc
t1 = time.perf_counter()
n = mc._lib.lib().mc_device_count()
t2 = time.perf_counter()
eng = mc.Engine(0)
t3 = time.perf_counter()
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
eng.vanilla(VAN, 1000, precision="f32")
t4 = time.perf_counter()
eng.vanilla(VAN, 1000, precision="f64")
t5 = time.perf_counter()
eng.cva(dict(VAN, defint=0.03, lgd=0.6, n_grid=50), 1000, precision="f64")
t6 = time.perf_counter()
eng.vanilla(VAN, 1000, precision="f32")
t7 = time.perf_counter()
print(f"import {1e3*(t1-t0):.1f} ms, device_count (hipInit) {1e3*(t2-t1):.1f}, context_create {1e3*(t3-t2):.1f}, first f32 call {1e3*(t4-t3):.2f}, first f64 call {1e3*(t5-t4):.2f}, first cva call {1e3*(t6-t5):.2f}, repeat {1e3*(t7-t6):.3f}")
