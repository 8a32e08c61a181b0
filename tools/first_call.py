"""First calls of a context under the stage breakdown (mc_call_stats): what creating the context costs and what the first pricing calls cost
after it -- python tools/first_call.py on the GPU box (profiles/r05_first_call_stages.log)."""
import sys, time
sys.path.insert(0, '.')
import torch
import montecarlocuda_amd as mc
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
t0 = time.perf_counter()
e = mc.Engine(0)
t1 = time.perf_counter()
r = e.vanilla(VAN, 10**6, precision="f32"); k1 = e.last_call_stats()
r = e.vanilla(VAN, 10**6, precision="f32"); k2 = e.last_call_stats()
r = e.cva(dict(VAN, r=0.05, defint=0.03, lgd=0.6, n_grid=64), 10**4, precision="f64"); k3 = e.last_call_stats()
print("create (python wall) %.1f ms; stats: create %.1f ms" % ((t1 - t0) * 1e3, k1["context_create_ms"]))
print("first vanilla call:", {a: round(b, 3) for a, b in k1.items()})
print("second vanilla call:", {a: round(b, 3) for a, b in k2.items()})
print("first CVA call:", {a: round(b, 3) for a, b in k3.items()})
e2 = mc.Engine(0)
e2.vanilla(VAN, 10**6, precision="f32"); print("second context: create %.1f ms, first call launch %.3f ms" % (e2.last_call_stats()["context_create_ms"], e2.last_call_stats()["launch_ms"]))
