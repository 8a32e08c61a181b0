"""libmchost_f64/_f32 (montecarlocuda_amd/csrc/host_path.c): the reference's HOST entry points as a
many-core CPU twin of the GPU estimator.  Checked here without a GPU against the reference's own
golden outputs (host_bsCall, Chol: bit for bit) and against the oracle's device-formula family on
the same Philox stream (host_vanillaOpt / host_basketOpt / host_cvaEquityOption)."""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import fromhex, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlocuda_amd", "csrc")
SEED = 0x4D435F4D49333535
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)


def load(po, X, n=3):
    path = os.path.join(CSRC, f"libmchost_{X}.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", CSRC, "all"], stdout=subprocess.DEVNULL)
    L = C.CDLL(path)
    OptionData, MultiOptionData, OptionValue, CVA = po.ref_types(X, n)
    R = po.CT[X]
    L.host_bsCall.argtypes = [OptionData]
    L.host_bsCall.restype = R
    L.host_vanillaOpt.argtypes = [OptionData, C.c_int]
    L.host_vanillaOpt.restype = OptionValue
    L.host_basketOpt.argtypes = [C.POINTER(MultiOptionData), C.c_int]
    L.host_basketOpt.restype = OptionValue
    L.host_cvaEquityOption.argtypes = [C.POINTER(CVA), C.c_int]
    L.host_cvaEquityOption.restype = OptionValue
    L.Chol.argtypes = [C.POINTER((R * n) * n), C.POINTER((R * n) * n)]
    return L, OptionData, MultiOptionData, OptionValue, CVA


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_bs_call_bitwise_vs_reference_golden(po, X):
    L, OptionData, *_ = load(po, X)
    n = 0
    for c in load_golden("ref_bs_call.json")["cases"]:
        if c["X"] != X:
            continue
        got = L.host_bsCall(OptionData(c["s"], c["k"], c["r"], c["v"], c["t"]))
        assert float(got) == fromhex(c["out"]), c
        n += 1
    assert n > 250


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_chol_bitwise_vs_reference_golden(po, X):
    L, *_ = load(po, X)
    R = po.CT[X]
    for c in load_golden("ref_chol.json")["cases"]:
        if c["X"] != X or c["n"] != 3:
            continue
        cc, a = ((R * 3) * 3)(), ((R * 3) * 3)()
        for i in range(3):
            for j in range(3):
                cc[i][j] = fromhex(c["c"][i][j])
        L.Chol(C.byref(cc), C.byref(a))
        assert [[float(a[i][j]) for j in range(3)] for i in range(3)] == [[fromhex(x) for x in row] for row in c["a"]]


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_host_monte_carlo_matches_oracle_on_the_engine_stream(po, X):
    L, OptionData, MultiOptionData, OptionValue, CVA = load(po, X)
    tol = 1e-12 if X == "f64" else 2e-6      # chunked fp64 summation order; f32 values round to float on return
    n = 200001
    v = L.host_vanillaOpt(OptionData(*[VAN[k] for k in "skrvt"]), n)
    _, o = po.dev_vanilla(X, VAN, SEED, 0, n, want_paths=False)
    assert float(v.Expected) == pytest.approx(o["expected"], rel=tol)
    assert float(v.Confidence) == pytest.approx(o["confidence"], rel=tol)
    # basket, the reference driver's N=3 data (singular correlation, zero-pivot factor)
    Lf = po.chol(X, [[1, -.5, -.5], [-.5, 1, -.5], [-.5, -.5, 1]])
    b = dict(s=[100.0] * 3, v=[0.2, 0.3, 0.2], p=Lf.tolist(), d=[0.0, 0.01, -0.01], w=[1 / 3] * 3, k=100.0, t=1.0, r=0.048790164)
    m = MultiOptionData()
    for i in range(3):
        m.s[i], m.v[i], m.d[i], m.w[i] = b["s"][i], b["v"][i], b["d"][i], b["w"][i]
        for j in range(3):
            m.p[i][j] = b["p"][i][j]
    m.k, m.t, m.r = b["k"], b["t"], b["r"]
    v = L.host_basketOpt(C.byref(m), 70001)
    _, o = po.dev_basket(X, b, SEED, 0, 70001, want_paths=False)
    assert float(v.Expected) == pytest.approx(o["expected"], rel=tol)
    assert float(v.Confidence) == pytest.approx(o["confidence"], rel=tol)
    # CVA, device ordering; 250 dates ends with a negative residual maturity in f64, 256 with exactly 0
    for n_grid in (25, 250, 256):
        c = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=n_grid)
        s = CVA(c["defint"], c["lgd"], 0, OptionData(*[c[k] for k in "skrvt"]), n_grid)
        v = L.host_cvaEquityOption(C.byref(s), 3001)
        _, o = po.dev_cva(X, c, SEED, 0, 3001, want_paths=False)
        assert float(v.Expected) == pytest.approx(o["expected"], rel=tol)
        assert float(v.Confidence) == pytest.approx(o["confidence"], rel=10 * tol)


def test_f64_twin_on_fp32_normals_matches_oracle_mode(po, monkeypatch):
    """MC_F64_NORMALS=f32: the CPU twin's fp64 entry points draw four fp32 normals per Philox block, widened -- the GPU
    engine's MC_NORMALS_F32 mode (the reference's dp arithmetic, dp/MonteCarloKernel.cu:68,78,250).  Same libm on both
    sides here, so the agreement with the oracle's mode is at the fp64 summation level."""
    monkeypatch.setenv("MC_F64_NORMALS", "f32")
    L, OptionData, MultiOptionData, OptionValue, CVA = load(po, "f64")
    n = 70001
    v = L.host_vanillaOpt(OptionData(*[VAN[k] for k in "skrvt"]), n)
    with po.normals_f32():
        _, o = po.dev_vanilla("f64", VAN, SEED, 0, n, want_paths=False)
    _, native = po.dev_vanilla("f64", VAN, SEED, 0, n, want_paths=False)
    assert float(v.Expected) == pytest.approx(o["expected"], rel=1e-12) and float(v.Confidence) == pytest.approx(o["confidence"], rel=1e-12)
    assert abs(float(v.Expected) - native["expected"]) > 1e-6        # a different stream than the native one
    Lf = po.chol("f64", [[1, .5, .5], [.5, 1, .5], [.5, .5, 1]])
    b = dict(s=[100.0] * 3, v=[0.2, 0.3, 0.2], p=Lf.tolist(), d=[0.0, 0.01, -0.01], w=[1 / 3] * 3, k=100.0, t=1.0, r=0.048790164)
    m = MultiOptionData()
    for i in range(3):
        m.s[i], m.v[i], m.d[i], m.w[i] = b["s"][i], b["v"][i], b["d"][i], b["w"][i]
        for j in range(3):
            m.p[i][j] = b["p"][i][j]
    m.k, m.t, m.r = b["k"], b["t"], b["r"]
    v = L.host_basketOpt(C.byref(m), 30001)
    with po.normals_f32():
        _, o = po.dev_basket("f64", b, SEED, 0, 30001, want_paths=False)
    assert float(v.Expected) == pytest.approx(o["expected"], rel=1e-12) and float(v.Confidence) == pytest.approx(o["confidence"], rel=1e-12)
    for n_grid in (3, 250, 256):
        c = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=n_grid)
        s = CVA(c["defint"], c["lgd"], 0, OptionData(*[c[k] for k in "skrvt"]), n_grid)
        v = L.host_cvaEquityOption(C.byref(s), 1501)
        with po.normals_f32():
            _, o = po.dev_cva("f64", c, SEED, 0, 1501, want_paths=False)
        assert float(v.Expected) == pytest.approx(o["expected"], rel=1e-12) and float(v.Confidence) == pytest.approx(o["confidence"], rel=1e-11)


def test_result_does_not_depend_on_thread_count(po):
    """Fixed 65536-path chunks added in index order: 1 thread and all threads give the same bits."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from oracle import pyoracle as po\n"
            "OD = po.ref_types('f64', 3)[0]; OV = po.ref_types('f64', 3)[2]\n"
            "L = C.CDLL(%r); L.host_vanillaOpt.argtypes=[OD, C.c_int]; L.host_vanillaOpt.restype = OV\n"
            "v = L.host_vanillaOpt(OD(100, 100, 0.04879, 0.2, 1), 1000003); print(float(v.Expected).hex(), float(v.Confidence).hex())\n"
            % (ROOT, os.path.join(CSRC, "libmchost_f64.so")))
    outs = []
    for threads in ("1", "3", "8"):
        env = dict(os.environ, MC_HOST_THREADS=threads)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip())
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_vectorised_loops_match_the_scalar_forms(po, X):
    """host_simd.c (whole batches of 256, compiled for x86-64 / AVX2 / AVX-512, glibc's vector math) against the scalar
    loops of host_path.c (MC_HOST_SCALAR=1) and against the oracle, for the three products: same stream, same formulas,
    values within a few ulp -- plain and antithetic (basket: control variate too), path counts that leave a scalar
    remainder."""
    import json
    import subprocess
    import sys
    lib = os.path.join(ROOT, "montecarlocuda_amd", "csrc", "libmchost_%s.so" % X)
    ct = "c_double" if X == "f64" else "c_float"
    code = ("import ctypes as C, json\n"
            "R = C.%s\n"
            "class OD(C.Structure): _fields_ = [(k, R) for k in 'skrvt']\n"
            "class OV(C.Structure): _fields_ = [('Expected', R), ('Confidence', R)]\n"
            "class MO(C.Structure): _fields_ = [('s', R * 3), ('v', R * 3), ('p', (R * 3) * 3), ('d', R * 3), ('w', R * 3), ('k', R), ('t', R), ('r', R)]\n"
            "class CVA(C.Structure): _fields_ = [('defInt', R), ('lgd', R), ('ns', C.c_int), ('option', OD), ('n', C.c_int)]\n"
            "L = C.CDLL(%r)\n"
            "for f in ('host_vanillaOpt', 'host_basketOpt', 'host_cvaEquityOption'): getattr(L, f).restype = OV\n"
            "L.host_vanillaOpt.argtypes = [OD, C.c_int]\n"
            "out = []\n"
            "v = L.host_vanillaOpt(OD(100, 100, 0.04879, 0.2, 1), 1000003); out += [float(v.Expected), float(v.Confidence)]\n"
            "m = MO(); P = %r\n"
            "for i in range(3):\n"
            "    m.s[i], m.v[i], m.d[i], m.w[i] = 100.0, (0.2, 0.3, 0.2)[i], (0.0, 0.01, -0.01)[i], (0.3, 0.3, 0.4)[i]\n"
            "    for j in range(3): m.p[i][j] = P[i][j]\n"
            "m.k, m.t, m.r = 100.0, 1.0, 0.048790164\n"
            "v = L.host_basketOpt(C.byref(m), 70001); out += [float(v.Expected), float(v.Confidence)]\n"
            "c = CVA(0.03, 0.6, 0, OD(100, 100, 0.05, 0.2, 1), 250)\n"
            "v = L.host_cvaEquityOption(C.byref(c), 3001); out += [float(v.Expected), float(v.Confidence)]\n"
            "print(json.dumps(out))\n")
    Lf = po.chol(X, [[1, .5, .5], [.5, 1, .5], [.5, .5, 1]]).tolist()
    code = code % (ct, lib, Lf)
    base = {k: v for k, v in os.environ.items() if not k.startswith("MC_")}
    run = lambda env: json.loads(subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True,
                                                env=dict(base, **env)).stdout)
    tol = 1e-12 if X == "f64" else 2e-6
    b = dict(s=[100.0] * 3, v=[0.2, 0.3, 0.2], p=Lf, d=[0.0, 0.01, -0.01], w=[0.3, 0.3, 0.4], k=100.0, t=1.0, r=0.048790164)
    c = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=250)
    for est in ({}, {"MC_ANTITHETIC": "1"}, {"MC_CONTROL_VARIATE": "1"}, {"MC_ANTITHETIC": "1", "MC_CONTROL_VARIATE": "1"}):
        anti, cv = "MC_ANTITHETIC" in est, "MC_CONTROL_VARIATE" in est
        scalar = run(dict(est, MC_HOST_SCALAR="1"))
        want = []
        _, o = po.dev_vanilla(X, dict(s=100.0, k=100.0, r=0.04879, v=0.2, t=1.0), SEED, 0, 1000003, want_paths=False, antithetic=anti)
        want += [o["expected"], o["confidence"]]
        _, o = po.dev_basket(X, b, SEED, 0, 70001, want_paths=False, antithetic=anti, control=cv)
        want += [o["expected"], o["confidence"]]       # with the control variate: its closed-form mean already added back
        _, o = po.dev_cva(X, c, SEED, 0, 3001, want_paths=False, antithetic=anti)
        want += [o["expected"], o["confidence"]]
        for isa in ("base", "avx2", "avx512", ""):      # "" = whatever the CPU supports; an unsupported request falls back
            got = run(dict(est, MC_HOST_ISA=isa) if isa else est)
            for k in range(6):
                # confidences of the control-variate estimator are differences of nearly equal numbers: looser
                t_k = tol * (50 if (cv and k in (2, 3)) or k == 5 else 1)
                assert got[k] == pytest.approx(scalar[k], rel=t_k), (isa, est, k)
                assert got[k] == pytest.approx(want[k], rel=t_k), (isa, est, k)


def test_thread_count_follows_the_override_and_the_cpu_quota(po):
    """mc_host_threads(): MC_HOST_THREADS if set, else OpenMP's processors capped by the cgroup CPU quota (a pod that
    sees 256 hardware threads but is granted 16 CPUs runs half as fast on 256 threads as on 16)."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "montecarlocuda_amd", "csrc", "libmchost_f64.so")
    code = "import ctypes as C; print(C.CDLL(%r).mc_host_threads())" % lib
    run = lambda env: int(subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True,
                                         env=dict({k: v for k, v in os.environ.items() if k != "MC_HOST_THREADS"}, **env)).stdout)
    assert run({"MC_HOST_THREADS": "3"}) == 3
    n = run({})
    limit = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        limit = None if q == "max" else -(-int(q) // int(p))
    except OSError:
        pass
    assert 1 <= n <= len(os.sched_getaffinity(0)) or limit is not None
    if limit is not None:
        assert n <= max(1, limit)


def test_drivers_build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")], stdout=subprocess.DEVNULL)
    for b in ("vanillaOpt", "basketOpt", "cvaOpt"):
        for X in ("f64", "f32"):
            assert os.path.exists(os.path.join(ROOT, "drivers", f"{b}_{X}"))


def test_antithetic_cpu_twin_matches_oracle(po):
    """MC_ANTITHETIC=1 switches the CPU twin to the antithetic estimator (as it does the legacy GPU symbols)."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from oracle import pyoracle as po\n"
            "OD, MOD, OV, CVA = po.ref_types('f64', 3)\n"
            "L = C.CDLL(%r); L.host_vanillaOpt.argtypes=[OD, C.c_int]; L.host_vanillaOpt.restype = OV\n"
            "L.host_cvaEquityOption.argtypes=[C.POINTER(CVA), C.c_int]; L.host_cvaEquityOption.restype = OV\n"
            "v = L.host_vanillaOpt(OD(100, 100, 0.04879, 0.2, 1), 100003)\n"
            "c = CVA(0.03, 0.6, 0, OD(100, 100, 0.05, 0.2, 1), 25); w = L.host_cvaEquityOption(C.byref(c), 3001)\n"
            "print(float(v.Expected).hex(), float(v.Confidence).hex(), float(w.Expected).hex(), float(w.Confidence).hex())\n"
            % (ROOT, os.path.join(CSRC, "libmchost_f64.so")))
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, MC_ANTITHETIC="1")).decode().split()
    e, ci, ce, cci = (float.fromhex(x) for x in out)
    _, o = po.dev_vanilla("f64", dict(VAN, r=0.04879), SEED, 0, 100003, want_paths=False, antithetic=True)
    assert e == pytest.approx(o["expected"], rel=1e-12) and ci == pytest.approx(o["confidence"], rel=1e-12)
    c = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=25)
    _, o = po.dev_cva("f64", c, SEED, 0, 3001, want_paths=False, antithetic=True)
    assert ce == pytest.approx(o["expected"], rel=1e-12) and cci == pytest.approx(o["confidence"], rel=1e-11)


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_small_host_helpers(po, X):
    """prodMat / randMinMax, which the reference's host file also exports (MonteCarloHost.c:67,111)."""
    L, R = load(po, X)[0], (C.c_double if X == "f64" else C.c_float)
    a = np.arange(6, dtype=R).reshape(2, 3) + 1
    b = (np.arange(12, dtype=R).reshape(3, 4) - 5) / 3
    out = np.zeros((2, 4), dtype=R)
    ptr = lambda m: m.ctypes.data_as(C.POINTER(R))   # noqa: E731
    L.prodMat.argtypes = [C.POINTER(R)] * 3 + [C.c_int] * 3
    L.prodMat.restype = None
    L.prodMat(ptr(a), ptr(b), ptr(out), 2, 3, 4)
    np.testing.assert_allclose(out, a.astype(np.float64) @ b.astype(np.float64), rtol=1e-6 if X == "f32" else 1e-14)
    L.randMinMax.argtypes = [R, R]
    L.randMinMax.restype = R
    draws = [L.randMinMax(-2.0, 3.0) for _ in range(200)]
    assert all(-2.0 <= d <= 3.0 for d in draws) and len(set(draws)) > 150


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_explicit_seed_variants(po, X, monkeypatch):
    """host_*_ex(..., seed): the reference's API has no seed parameter; the added variants take one and equal the
    plain calls under MC_SEED."""
    L, OptionData, MultiOptionData, OptionValue, CVA = load(po, X)
    L.host_vanillaOpt_ex.argtypes = [OptionData, C.c_int, C.c_uint64]
    L.host_vanillaOpt_ex.restype = OptionValue
    L.host_cvaEquityOption_ex.argtypes = [C.POINTER(CVA), C.c_int, C.c_uint64]
    L.host_cvaEquityOption_ex.restype = OptionValue
    L.host_basketOpt_ex.argtypes = [C.POINTER(MultiOptionData), C.c_int, C.c_uint64]
    L.host_basketOpt_ex.restype = OptionValue
    o = OptionData(*[VAN[k] for k in "skrvt"])
    a = L.host_vanillaOpt_ex(o, 50001, 777)
    _, want = po.dev_vanilla(X, VAN, 777, 0, 50001, want_paths=False)
    assert float(a.Expected) == pytest.approx(want["expected"], rel=1e-12 if X == "f64" else 2e-6)
    monkeypatch.setenv("MC_SEED", "777")
    b = L.host_vanillaOpt(o, 50001)
    assert (a.Expected, a.Confidence) == (b.Expected, b.Confidence)
    monkeypatch.setenv("MC_SEED", "778")
    assert L.host_vanillaOpt(o, 50001).Expected != a.Expected
    a2 = L.host_vanillaOpt_ex(o, 50001, 777)          # the explicit seed wins over the environment, for its own call only
    assert (a2.Expected, a2.Confidence) == (a.Expected, a.Confidence)
    assert L.host_vanillaOpt(o, 50001).Expected != a.Expected
    c = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=25)
    s = CVA(c["defint"], c["lgd"], 0, OptionData(*[c[k] for k in "skrvt"]), 25)
    v = L.host_cvaEquityOption_ex(C.byref(s), 3001, 5)
    _, want = po.dev_cva(X, c, 5, 0, 3001, want_paths=False)
    assert float(v.Expected) == pytest.approx(want["expected"], rel=1e-12 if X == "f64" else 2e-6)
    m = MultiOptionData()
    for i in range(3):
        m.s[i], m.v[i], m.d[i], m.w[i] = 100.0, 0.2, 0.0, 1 / 3
        m.p[i][i] = 1.0
    m.k, m.t, m.r = 100.0, 1.0, 0.05
    assert L.host_basketOpt_ex(C.byref(m), 20001, 9).Expected != L.host_basketOpt_ex(C.byref(m), 20001, 10).Expected
