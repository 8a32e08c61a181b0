#!/usr/bin/env python3
"""Speed of the CPU twin (libmchost_*: host_vanillaOpt / host_basketOpt (N = 3) / host_cvaEquityOption) on this host.

    MC_HOST_THREADS=16 MC_HOST_ISA=avx512 python tools/host_scale.py [f32|f64] [vanilla paths]
    MC_HOST_SCALAR=1 ... the scalar loops
"""
import ctypes as C, os, sys, time
X = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("f32", "f64") else "f32"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 400000000
R = C.c_float if X == "f32" else C.c_double
class OD(C.Structure): _fields_ = [(k, R) for k in "skrvt"]
class OV(C.Structure): _fields_ = [("Expected", R), ("Confidence", R)]
class MO(C.Structure): _fields_ = [("s", R * 3), ("v", R * 3), ("p", (R * 3) * 3), ("d", R * 3), ("w", R * 3), ("k", R), ("t", R), ("r", R)]
class CVA(C.Structure): _fields_ = [("defInt", R), ("lgd", R), ("ns", C.c_int), ("option", OD), ("n", C.c_int)]
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(f"{root}/montecarlocuda_amd/csrc/libmchost_{X}.so")
for f in ("host_vanillaOpt", "host_basketOpt", "host_cvaEquityOption"):
    getattr(L, f).restype = OV
L.host_vanillaOpt.argtypes = [OD, C.c_int]
L.mc_host_threads.restype = C.c_int
m = MO()
Lf = [[1, 0, 0], [0.5, 0.8660254037844386, 0], [0.5, 0.2886751345948129, 0.816496580927726]]
for i in range(3):
    m.s[i], m.v[i], m.d[i], m.w[i] = 100.0, (0.2, 0.3, 0.2)[i], 0.0, 1 / 3
    for j in range(3):
        m.p[i][j] = Lf[i][j]
m.k, m.t, m.r = 100.0, 1.0, 0.048790164
c = CVA(0.03, 0.6, 0, OD(100, 100, 0.05, 0.2, 1), 256)
L.host_vanillaOpt(OD(100, 100, 0.048790, 0.2, 1), 2000000)
tag = f"{X} threads {L.mc_host_threads()} isa {os.environ.get('MC_HOST_ISA', 'auto')}{' scalar' if os.environ.get('MC_HOST_SCALAR') else ''}"
for name, n, call in (("vanilla", nv, lambda n: L.host_vanillaOpt(OD(100, 100, 0.048790, 0.2, 1), n)),
                      ("basket3", nv // 8, lambda n: L.host_basketOpt(C.byref(m), n)),
                      ("cva256", max(3000, nv // 2000), lambda n: L.host_cvaEquityOption(C.byref(c), n))):
    best = 1e30
    for _ in range(2):
        t = time.perf_counter(); v = call(n); best = min(best, time.perf_counter() - t)
    print(f"{tag}  {name:8s} {n / best / 1e6:10.3f} Mpaths/s  ({n} paths, {best:.3f} s)  value {float(v.Expected):.6f}", flush=True)
