import ctypes as C, time, sys, os
R=C.c_float
class OptionData(C.Structure): _fields_=[(k,R) for k in "skrvt"]
class OptionValue(C.Structure): _fields_=[("Expected",R),("Confidence",R)]
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
L=C.CDLL(f'{root}/montecarlocuda_amd/csrc/libmchost_f32.so')
L.host_vanillaOpt.argtypes=[OptionData,C.c_int]; L.host_vanillaOpt.restype=OptionValue
o=OptionData(100,100,0.048790,0.2,1.0)
n=int(sys.argv[1])
L.host_vanillaOpt(o,2000000)
for _ in range(2):
    t=time.perf_counter(); v=L.host_vanillaOpt(o,n); dt=time.perf_counter()-t
    print("threads", os.environ.get("MC_HOST_THREADS","all"), n/dt/1e6,"Mpaths/s", round(dt,3), "s", flush=True)
