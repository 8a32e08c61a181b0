#!/usr/bin/env python3
"""Per-call latency of the drop-in entry point at the reference's own call sizes (kernel time AND
whole-call wall time, context creation reported once, separately).

The reference driver prices k x 131 072 paths per dev_vanillaOpt call and times the whole call, allocation and RNG
set-up included (double_precision/vanillaOpt.cu:51-53,77-83).  Here: dev_vanillaOpt of libmcgpu_f32/_f64.so (the
legacy symbol, by-value OptionValue) for k = 8, 80, 800, and the native mc_vanilla_run_* underneath it, each with the
in-kernel final reduction (default) and with the two-launch form (MC_FINISH=kernel) -- one child process per form,
alternated.  Wall = host clock around the call (launch + wait + 24-byte read-back), median of 300 calls after 20
warm-ups; kernel = HIP events around the call's kernels (mc_result.kernel_ms).

    python tools/call_latency.py            # on the GPU box; prints the table
"""
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)


def child():
    import montecarlocuda_amd as mc
    out = {}
    t0 = time.perf_counter()
    eng = mc.Engine(0)
    eng.vanilla(VAN, 1000, precision="f32")
    out["context_create_and_first_call_ms"] = (time.perf_counter() - t0) * 1e3
    for X in ("f32", "f64"):
        R = C.c_float if X == "f32" else C.c_double

        class OptionData(C.Structure):
            _fields_ = [(k, R) for k in "skrvt"]

        class OptionValue(C.Structure):
            _fields_ = [("Expected", R), ("Confidence", R)]
        L = C.CDLL(mc._lib.LEGACY[X])
        L.dev_vanillaOpt.argtypes = [C.POINTER(OptionData), C.c_int, C.c_int, C.c_int]
        L.dev_vanillaOpt.restype = OptionValue
        o = OptionData(*[VAN[k] for k in "skrvt"])
        for k in (8, 80, 800):
            sims = k * 131072
            for _ in range(20):
                L.dev_vanillaOpt(C.byref(o), 512, 128, sims)
                eng.vanilla(VAN, sims, precision=X)
            legacy, native_wall, native_kernel, direct_wall = [], [], [], []
            for _ in range(300):
                t0 = time.perf_counter()
                v = L.dev_vanillaOpt(C.byref(o), 512, 128, sims)
                legacy.append((time.perf_counter() - t0) * 1e6)
                e = eng.vanilla(VAN, sims, precision=X)
                native_wall.append(e.wall_ms * 1e3)
                native_kernel.append(e.kernel_ms * 1e3)
                eng.set_timing(False)
                direct_wall.append(eng.vanilla(VAN, sims, precision=X).wall_ms * 1e3)
                eng.set_timing(True)
            out[f"{X} k={k}"] = {"paths": sims, "dev_vanillaOpt_wall_us": statistics.median(legacy),
                                 "mc_vanilla_run_wall_us": statistics.median(native_wall),
                                 "mc_vanilla_run_untimed_wall_us": statistics.median(direct_wall),
                                 "kernel_us": statistics.median(native_kernel), "price": float(v.Expected)}
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    res = {}
    for rep in range(2):
        for form, env in (("fused", {}), ("two-launch", {"MC_FINISH": "kernel"})):
            p = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                sys.exit(p.stderr[-2000:])
            res.setdefault(form, []).append(json.loads(line[-1]))
    print("# tools/call_latency.py: dev_vanillaOpt(&opt, 512, 128, k * 131072), one MI355X; medians of 300 calls, two passes per form")
    print("# context creation + first call (once per process): " +
          ", ".join(f"{form} {statistics.mean(r['context_create_and_first_call_ms'] for r in runs):.0f} ms" for form, runs in res.items()))
    print("# dev_vanillaOpt (legacy symbol) and the 'untimed' column run with mc_context_set_timing(ctx, 0): no HIP events, the result")
    print("# written by the last workgroup into pinned host memory and polled (fused form only; the two-launch form copies and synchronises).")
    print(f"{'call':12s} {'paths':>11s} | {'form':10s} {'dev_vanillaOpt wall us':>23s} {'mc_vanilla_run wall us':>23s} {'untimed wall us':>16s} {'kernel us':>10s}")
    for key in [k for k in res["fused"][0] if "k=" in k]:
        for form, runs in res.items():
            f = lambda name: min(r[key][name] for r in runs)   # noqa: E731  (best of the two passes)
            print(f"{key:12s} {runs[0][key]['paths']:11d} | {form:10s} {f('dev_vanillaOpt_wall_us'):23.1f} {f('mc_vanilla_run_wall_us'):23.1f} "
                  f"{f('mc_vanilla_run_untimed_wall_us'):16.1f} {f('kernel_us'):10.1f}")


if __name__ == "__main__":
    main()
