"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package ``montecarlocuda_amd`` never does.

Two libraries are bound here:

* ``oracle/liboracle.so`` -- our CPU restatement (``mc_oracle.c``), see ``mc_oracle.h``.
* ``oracle/_ref/libref_{f32,f64}_n{3,4,16}.so`` -- the UNMODIFIED reference CPU path
  (``/root/reference/{single,double}_precision/MonteCarloHost.c``) compiled by
  ``oracle/Makefile``; present only where that recipe ran (dev container) or where the
  prebuilt files travelled (GPU box).  Struct layouts mirror reference ``MonteCarlo.h:32-65``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from functools import lru_cache

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")

DOMAIN_VANILLA, DOMAIN_BASKET, DOMAIN_CVA = 1, 2, 3
NPB = {"f32": 4, "f64": 8}   # normals per block of the stream (f64: three Philox blocks, stream version 2)
CT = {"f32": C.c_float, "f64": C.c_double}
NP = {"f32": np.float32, "f64": np.float64}


class OrcResult(C.Structure):
    _fields_ = [("expected", C.c_double), ("confidence", C.c_double), ("sum", C.c_double),
                ("sum2", C.c_double), ("n", C.c_longlong)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(HERE, f))
                                             for f in ("mc_oracle.c", "mc_oracle_impl.h", "mc_oracle.h")):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/double_precision"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


@lru_cache(maxsize=None)
def lib() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    u32p = C.POINTER(C.c_uint32)
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_philox4x32_10.restype = None
    L.orc_closing.argtypes = [C.c_double, C.c_double, C.c_longlong, C.c_double,
                              C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_closing.restype = None
    for X in ("f32", "f64"):
        R = CT[X]
        RP = C.POINTER(R)
        res = C.POINTER(OrcResult)

        def f(name):
            return getattr(L, f"{name}_{X}")
        f("orc_cnd").argtypes = [R]
        f("orc_cnd").restype = R
        f("orc_bs_call").argtypes = [R] * 5
        f("orc_bs_call").restype = R
        f("orc_chol").argtypes = [C.c_int, RP, RP]
        f("orc_chol").restype = None
        for nm in ("orc_host_uniforms", "orc_host_gaussians"):
            f(nm).argtypes = [C.c_uint, C.c_int, RP]
            f(nm).restype = None
        f("orc_host_vanilla").argtypes = [R] * 5 + [C.c_int, C.c_uint, res]
        f("orc_host_basket").argtypes = [C.c_int, RP, RP, RP, RP, RP, R, R, R, C.c_int, C.c_uint,
                                         C.c_int, res]
        f("orc_host_cva").argtypes = [R] * 7 + [C.c_int, C.c_int, C.c_uint, res]
        f("orc_dev_normals").argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, RP]
        f("orc_dev_vanilla").argtypes = [R] * 5 + [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, RP, res]
        f("orc_dev_basket").argtypes = [C.c_int, RP, RP, RP, RP, RP, R, R, R, C.c_uint64,
                                        C.c_uint64, C.c_uint64, C.c_int, RP, res]
        f("orc_dev_cva").argtypes = [R] * 7 + [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, RP, res]
        f("orc_dev_vanilla_greeks").argtypes = [R] * 5 + [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcResult * 3)]
        f("orc_dev_vanilla_greeks").restype = None
        f("orc_basket_control_mean").argtypes = [C.c_int, RP, RP, RP, RP, RP, R, R, R]
        f("orc_basket_control_mean").restype = C.c_double
        for nm in ("orc_host_vanilla", "orc_host_basket", "orc_host_cva", "orc_dev_normals",
                   "orc_dev_vanilla", "orc_dev_basket", "orc_dev_cva"):
            f(nm).restype = None
        # per-path taps of the host family, the reference's accumulation, the device formulas on a given normal stream
        f("orc_host_vanilla_paths").argtypes = [R] * 5 + [C.c_int, C.c_uint, RP, res]
        f("orc_host_basket_paths").argtypes = [C.c_int, RP, RP, RP, RP, RP, R, R, R, C.c_int, C.c_uint, C.c_int, RP, res]
        f("orc_host_cva_paths").argtypes = [R] * 7 + [C.c_int, C.c_int, C.c_uint, RP, res]
        f("orc_ref_close").argtypes = [RP, C.c_int, C.c_int, R, R, res]
        f("orc_dev_vanilla_on_normals").argtypes = [R] * 5 + [RP, C.c_uint64, C.c_int, RP, res]
        f("orc_dev_basket_on_normals").argtypes = [C.c_int, RP, RP, RP, RP, RP, R, R, R, RP, C.c_uint64, C.c_int, RP, res]
        f("orc_dev_cva_on_normals").argtypes = [R] * 7 + [C.c_int, RP, C.c_uint64, C.c_int, C.c_int, RP, res]
        for nm in ("orc_host_vanilla_paths", "orc_host_basket_paths", "orc_host_cva_paths", "orc_ref_close",
                   "orc_dev_vanilla_on_normals", "orc_dev_basket_on_normals", "orc_dev_cva_on_normals"):
            f(nm).restype = None
        f("orc_dev_npb").argtypes = []
        f("orc_dev_npb").restype = C.c_int
    L.orc_set_normals_f32.argtypes = [C.c_int]
    L.orc_set_normals_f32.restype = None
    return L


def _arr(a, X):
    a = np.ascontiguousarray(a, dtype=NP[X])
    return a, a.ctypes.data_as(C.POINTER(CT[X]))


# ------------------------------------------------------------------------------------------
# thin pythonic wrappers
# ------------------------------------------------------------------------------------------
def counter(unit, block, domain):
    """The engine's Philox counter for a 64-bit unit index: {unit_hi, unit_lo, block, domain}
    (montecarlocuda_amd/csrc/mc_rng.hpp: philox_unit; oracle/mc_oracle_impl.h: orc_dev_normals)."""
    return [(unit >> 32) & 0xFFFFFFFF, unit & 0xFFFFFFFF, block, domain]


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def xorwow_words(seed, subsequence, count):
    """First `count` 32-bit outputs of XORWOW(seed, subsequence) -- rocRAND's seeding and subsequence layout."""
    L = lib()
    L.orc_xorwow_init.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    L.orc_xorwow_init.restype = None
    L.orc_xorwow_next.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_xorwow_next.restype = C.c_uint32
    st = (C.c_uint32 * 6)()
    L.orc_xorwow_init(seed, subsequence, st)
    return [int(L.orc_xorwow_next(st)) for _ in range(count)]


def grid_normals(num_blocks, num_threads, count):
    """The reference's launch geometry (dp/MonteCarloKernel.cu:285-290): the first `count` normals of every thread's own
    XORWOW stream, shape (num_blocks, num_threads, count), float32 (orc_grid_normals)."""
    L = lib()
    L.orc_grid_normals.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
    L.orc_grid_normals.restype = None
    out = np.empty((num_blocks, num_threads, count), dtype=np.float32)
    row = (C.c_float * count)()
    for b in range(num_blocks):
        for t in range(num_threads):
            L.orc_grid_normals(num_blocks, b, t, count, row)
            out[b, t] = np.frombuffer(row, dtype=np.float32)
    return out


def grid_path_normals(streams, paths_per_block, draws):
    """Which normals each path of a grid-geometry call gets: thread t of a block prices paths t, t + T, ... < paths_per_block
    (dp/MonteCarloKernel.cu:146,191,240) and draws `draws` normals for each, one after the other, from its stream
    (`streams`: (num_blocks, num_threads, >= needed), from grid_normals or the device).  Returns
    (num_blocks * paths_per_block, draws), path p = b * paths_per_block + i."""
    G, T, have = streams.shape
    out = np.empty((G, paths_per_block, draws), dtype=streams.dtype)
    for t in range(min(T, paths_per_block)):
        k = len(range(t, paths_per_block, T))
        assert k * draws <= have
        out[:, t::T, :] = streams[:, t, :k * draws].reshape(G, k, draws)
    return out.reshape(G * paths_per_block, draws)


def grid_draws_per_thread(num_threads, paths_per_block, draws):
    """Normals the busiest thread (t = 0) of a block draws."""
    return len(range(0, paths_per_block, num_threads)) * draws


class xorwow_mode:
    """Context manager: inside it the dev_* functions draw their normals from one XORWOW sequence per lane
    (lane l = subsequence base + l prices units unit0 + l, unit0 + l + lanes, ...), like the product's XORWOW mode."""

    def __init__(self, seed, subsequence_base, lanes, unit0):
        self.args = (seed, subsequence_base, lanes, unit0)

    def __enter__(self):
        L = lib()
        L.orc_xorwow_begin.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_xorwow_begin.restype = None
        L.orc_xorwow_end.restype = None
        L.orc_xorwow_begin(*self.args)
        return self

    def __exit__(self, *a):
        lib().orc_xorwow_end()


def closing(s, s2, n, discount=1.0):
    e, c = C.c_double(), C.c_double()
    lib().orc_closing(s, s2, n, discount, C.byref(e), C.byref(c))
    return e.value, c.value


def cnd(X, d):
    return getattr(lib(), f"orc_cnd_{X}")(d)


def bs_call(X, s, k, r, v, t):
    return getattr(lib(), f"orc_bs_call_{X}")(s, k, r, v, t)


def chol(X, c):
    c = np.asarray(c, dtype=NP[X])
    n = c.shape[0]
    cc, cp = _arr(c, X)
    out = np.zeros((n, n), dtype=NP[X])
    getattr(lib(), f"orc_chol_{X}")(n, cp, out.ctypes.data_as(C.POINTER(CT[X])))
    return out


def factor_from_cov(X, cov):
    """(v, corr, factor, bad_pivots): the covariance twin of chol (oracle/mc_oracle_impl.h: orc_factor_from_cov)."""
    cov = np.ascontiguousarray(cov, dtype=NP[X])
    n = cov.shape[0]
    v, corr, a = np.zeros(n, dtype=NP[X]), np.zeros((n, n), dtype=NP[X]), np.zeros((n, n), dtype=NP[X])
    P = C.POINTER(CT[X])
    f = getattr(lib(), f"orc_factor_from_cov_{X}")
    f.argtypes = [C.c_int, P, P, P, P]
    f.restype = C.c_int
    bad = f(n, cov.ctypes.data_as(P), v.ctypes.data_as(P), corr.ctypes.data_as(P), a.ctypes.data_as(P))
    return v, corr, a, bad


def host_uniforms(X, seed, count):
    out = np.zeros(count, dtype=NP[X])
    getattr(lib(), f"orc_host_uniforms_{X}")(seed, count, out.ctypes.data_as(C.POINTER(CT[X])))
    return out


def host_gaussians(X, seed, count):
    out = np.zeros(count, dtype=NP[X])
    getattr(lib(), f"orc_host_gaussians_{X}")(seed, count, out.ctypes.data_as(C.POINTER(CT[X])))
    return out


def host_vanilla(X, opt, paths, seed):
    r = OrcResult()
    getattr(lib(), f"orc_host_vanilla_{X}")(opt["s"], opt["k"], opt["r"], opt["v"], opt["t"],
                                            paths, seed, C.byref(r))
    return r.as_dict()


def host_basket(X, b, paths, seed, vol_in_diffusion=None):
    """b: dict(s, v, p (factor, n x n), d, w, k, t, r).  vol_in_diffusion defaults to what the
    reference host of that precision does (f64: 0 = dp bug, f32: 1)."""
    if vol_in_diffusion is None:
        vol_in_diffusion = 1 if X == "f32" else 0
    n = len(b["s"])
    keep = [_arr(b[k], X) for k in ("s", "v", "p", "d", "w")]
    r = OrcResult()
    getattr(lib(), f"orc_host_basket_{X}")(n, *[p for _, p in keep], b["k"], b["t"], b["r"],
                                           paths, seed, vol_in_diffusion, C.byref(r))
    return r.as_dict()


def host_cva(X, c, paths, seed):
    """c: dict(s,k,r,v,t, defint, lgd, n_grid)."""
    r = OrcResult()
    getattr(lib(), f"orc_host_cva_{X}")(c["s"], c["k"], c["r"], c["v"], c["t"], c["defint"],
                                        c["lgd"], c["n_grid"], paths, seed, C.byref(r))
    return r.as_dict()


def dev_npb(X):
    """Normals per block the orc_dev_* family of precision X currently draws (4; 8 in native fp64)."""
    return int(getattr(lib(), f"orc_dev_npb_{X}")())


def dev_normals(X, seed, domain, unit, block):
    out = np.zeros(8, dtype=NP[X])
    getattr(lib(), f"orc_dev_normals_{X}")(seed, domain, unit, block,
                                           out.ctypes.data_as(C.POINTER(CT[X])))
    return out[:dev_npb(X)].copy()


class normals_f32:
    """Context manager: inside it the fp64 dev_* family draws FOUR fp32 normals per block, widened to double -- the
    reference's dp arithmetic (``double z = curand_normal(...)``), twin of the product's MC_NORMALS_F32 mode."""

    def __enter__(self):
        lib().orc_set_normals_f32(1)
        return self

    def __exit__(self, *a):
        lib().orc_set_normals_f32(0)


CVA_HOST_ORDER, CVA_REF_DP, CVA_REF_T0 = 1, 2, 4     # flags of dev_cva_on_normals (mc_oracle.h)
BASKET_NO_VOL = 4                                     # mode bit of dev_basket_on_normals: the dp CPU path's diffusion


def host_vanilla_paths(X, opt, paths, seed):
    """(per-path payoffs, result) of the reference CPU algorithm on its own stream (the E/CI are bit-pinned)."""
    out = np.zeros(paths, dtype=NP[X])
    r = OrcResult()
    getattr(lib(), f"orc_host_vanilla_paths_{X}")(opt["s"], opt["k"], opt["r"], opt["v"], opt["t"], paths, seed,
                                                  out.ctypes.data_as(C.POINTER(CT[X])), C.byref(r))
    return out, r.as_dict()


def host_basket_paths(X, b, paths, seed, vol_in_diffusion=None):
    if vol_in_diffusion is None:
        vol_in_diffusion = 1 if X == "f32" else 0
    n = len(b["s"])
    keep = [_arr(b[k], X) for k in ("s", "v", "p", "d", "w")]
    out = np.zeros(paths, dtype=NP[X])
    r = OrcResult()
    getattr(lib(), f"orc_host_basket_paths_{X}")(n, *[p for _, p in keep], b["k"], b["t"], b["r"], paths, seed,
                                                 vol_in_diffusion, out.ctypes.data_as(C.POINTER(CT[X])), C.byref(r))
    return out, r.as_dict()


def host_cva_paths(X, c, paths, seed):
    out = np.zeros(paths, dtype=NP[X])
    r = OrcResult()
    getattr(lib(), f"orc_host_cva_paths_{X}")(c["s"], c["k"], c["r"], c["v"], c["t"], c["defint"], c["lgd"], c["n_grid"],
                                              paths, seed, out.ctypes.data_as(C.POINTER(CT[X])), C.byref(r))
    return out, r.as_dict()


def ref_close(X, values, discounted, r, t):
    """The reference's sequential accumulation in REAL + its closing formulas on given per-path values."""
    v, vp = _arr(values, X)
    res = OrcResult()
    getattr(lib(), f"orc_ref_close_{X}")(vp, len(v), int(discounted), r, t, C.byref(res))
    return res.as_dict()


def dev_vanilla_on_normals(X, opt, z, antithetic=False):
    """The DEVICE vanilla formula with z[i] as path i's normal: (payoffs, result with fp64 sums)."""
    z, zp = _arr(z, X)
    out = np.zeros(len(z), dtype=NP[X])
    r = OrcResult()
    getattr(lib(), f"orc_dev_vanilla_on_normals_{X}")(opt["s"], opt["k"], opt["r"], opt["v"], opt["t"], zp, len(z),
                                                      int(antithetic), out.ctypes.data_as(C.POINTER(CT[X])), C.byref(r))
    return out, r.as_dict()


def dev_basket_on_normals(X, b, g, mode=0):
    """The DEVICE basket formulas with g[i, a] as path i's normal for asset a (drawing order)."""
    nn = len(b["s"])
    g, gp = _arr(np.asarray(g).reshape(-1), X)
    n_paths = len(g) // nn
    keep = [_arr(b[k], X) for k in ("s", "v", "p", "d", "w")]
    out = np.zeros(n_paths, dtype=NP[X])
    r = OrcResult()
    getattr(lib(), f"orc_dev_basket_on_normals_{X}")(nn, *[p for _, p in keep], b["k"], b["t"], b["r"], gp, n_paths,
                                                     mode, out.ctypes.data_as(C.POINTER(CT[X])), C.byref(r))
    return out, r.as_dict()


def dev_cva_on_normals(X, c, z, flags=0, antithetic=False):
    """The DEVICE CVA loop with z[i, j-1] as path i's normal at date j; flags: CVA_HOST_ORDER | CVA_REF_DP | CVA_REF_T0."""
    z, zp = _arr(np.asarray(z).reshape(-1), X)
    n_paths = len(z) // c["n_grid"]
    out = np.zeros(n_paths, dtype=NP[X])
    r = OrcResult()
    getattr(lib(), f"orc_dev_cva_on_normals_{X}")(c["s"], c["k"], c["r"], c["v"], c["t"], c["defint"], c["lgd"], c["n_grid"],
                                                  zp, n_paths, int(antithetic), flags, out.ctypes.data_as(C.POINTER(CT[X])),
                                                  C.byref(r))
    return out, r.as_dict()


def dev_vanilla(X, opt, seed, first, n, want_paths=True, antithetic=False):
    out = np.zeros(n if want_paths else 0, dtype=NP[X])
    ptr = out.ctypes.data_as(C.POINTER(CT[X])) if want_paths else None
    r = OrcResult()
    getattr(lib(), f"orc_dev_vanilla_{X}")(opt["s"], opt["k"], opt["r"], opt["v"], opt["t"], seed,
                                           first, n, int(antithetic), ptr, C.byref(r))
    return out, r.as_dict()


def dev_basket(X, b, seed, first, n, want_paths=True, antithetic=False, control=False):
    nn = len(b["s"])
    keep = [_arr(b[k], X) for k in ("s", "v", "p", "d", "w")]
    out = np.zeros(n if want_paths else 0, dtype=NP[X])
    ptr = out.ctypes.data_as(C.POINTER(CT[X])) if want_paths else None
    r = OrcResult()
    getattr(lib(), f"orc_dev_basket_{X}")(nn, *[p for _, p in keep], b["k"], b["t"], b["r"], seed,
                                          first, n, int(antithetic) | (int(control) << 1), ptr, C.byref(r))
    return out, r.as_dict()


def dev_vanilla_greeks(X, opt, seed, first, n):
    """(price, delta, vega) result dicts of the pathwise-Greeks twin."""
    r = (OrcResult * 3)()
    getattr(lib(), f"orc_dev_vanilla_greeks_{X}")(opt["s"], opt["k"], opt["r"], opt["v"], opt["t"], seed, first, n, C.byref(r))
    return [x.as_dict() for x in r]


def dev_vanilla_greeks_lr(X, opt, seed, first, n):
    """(price, delta, vega) result dicts of the likelihood-ratio twin."""
    r = (OrcResult * 3)()
    f = getattr(lib(), f"orc_dev_vanilla_greeks_lr_{X}")
    f.argtypes = [CT[X]] * 5 + [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcResult * 3)]
    f.restype = None
    f(opt["s"], opt["k"], opt["r"], opt["v"], opt["t"], seed, first, n, C.byref(r))
    return [x.as_dict() for x in r]


def dev_basket_greeks(X, b, seed, first, n, lr=False):
    """(price, [delta_a], [vega_a]) result dicts of the basket-Greeks twin: pathwise, or (lr=True) likelihood ratio."""
    nn = len(b["s"])
    keep = [_arr(b[k], X) for k in ("s", "v", "p", "d", "w")]
    r = (OrcResult * (1 + 2 * nn))()
    f = getattr(lib(), f"orc_dev_basket_greeks{'_lr' if lr else ''}_{X}")
    RP = C.POINTER(CT[X])
    f.argtypes = [C.c_int, RP, RP, RP, RP, RP, CT[X], CT[X], CT[X], C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    f.restype = C.c_int if lr else None
    rc = f(nn, *[p for _, p in keep], b["k"], b["t"], b["r"], seed, first, n, C.cast(r, C.c_void_p))
    if lr and rc != 0:
        raise ValueError("singular factor")
    out = [x.as_dict() for x in r]
    return out[0], out[1:1 + nn], out[1 + nn:]


def dev_cva_greeks(X, c, seed, first, n, lr=False):
    """(cva, delta, vega) result dicts of the CVA-Greeks twin: pathwise, or (lr=True) likelihood ratio."""
    r = (OrcResult * 3)()
    f = getattr(lib(), f"orc_dev_cva_greeks{'_lr' if lr else ''}_{X}")
    f.argtypes = [CT[X]] * 7 + [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcResult * 3)]
    f.restype = None
    f(c["s"], c["k"], c["r"], c["v"], c["t"], c["defint"], c["lgd"], c["n_grid"], seed, first, n, C.byref(r))
    return [x.as_dict() for x in r]


def basket_control_mean(X, b):
    keep = [_arr(b[k], X) for k in ("s", "v", "p", "d", "w")]
    return getattr(lib(), f"orc_basket_control_mean_{X}")(len(b["s"]), *[p for _, p in keep], b["k"], b["t"], b["r"])


def dev_cva(X, c, seed, first, n, want_paths=True, antithetic=False):
    out = np.zeros(n if want_paths else 0, dtype=NP[X])
    ptr = out.ctypes.data_as(C.POINTER(CT[X])) if want_paths else None
    r = OrcResult()
    getattr(lib(), f"orc_dev_cva_{X}")(c["s"], c["k"], c["r"], c["v"], c["t"], c["defint"],
                                       c["lgd"], c["n_grid"], seed, first, n, int(antithetic), ptr, C.byref(r))
    return out, r.as_dict()


# ------------------------------------------------------------------------------------------
# compiled reference (oracle/_ref)
# ------------------------------------------------------------------------------------------
def ref_available(X="f64", n=3, opt="") -> bool:
    return os.path.exists(os.path.join(REF_DIR, f"libref_{X}_n{n}{opt}.so"))


def ref_types(X, n):
    """ctypes mirrors of reference MonteCarlo.h:32-65 for precision X and asset count n."""
    R = CT[X]

    class OptionData(C.Structure):
        _fields_ = [("s", R), ("k", R), ("r", R), ("v", R), ("t", R)]

    class MultiOptionData(C.Structure):
        _fields_ = [("s", R * n), ("v", R * n), ("p", (R * n) * n), ("d", R * n), ("w", R * n),
                    ("k", R), ("t", R), ("r", R)]

    class OptionValue(C.Structure):
        _fields_ = [("Expected", R), ("Confidence", R)]

    class CVA(C.Structure):
        _fields_ = [("defInt", R), ("lgd", R), ("ns", C.c_int), ("option", OptionData),
                    ("n", C.c_int)]

    return OptionData, MultiOptionData, OptionValue, CVA


class Ref:
    """The unmodified reference host object for one (precision, N), with time() pinned."""

    def __init__(self, X="f64", n=3, opt=""):
        """opt = "" (the -O2 build) or "_O0" (the reference Makefile's own optimisation level; N = 3 only)."""
        self.X, self.n = X, n
        self.L = C.CDLL(os.path.join(REF_DIR, f"libref_{X}_n{n}{opt}.so"))
        self.OptionData, self.MultiOptionData, self.OptionValue, self.CVA = ref_types(X, n)
        R = CT[X]
        self.L.mcref_set_seed.argtypes = [C.c_uint]
        self.L.host_bsCall.argtypes = [self.OptionData]
        self.L.host_bsCall.restype = R
        self.L.host_vanillaOpt.argtypes = [self.OptionData, C.c_int]
        self.L.host_vanillaOpt.restype = self.OptionValue
        self.L.host_basketOpt.argtypes = [C.POINTER(self.MultiOptionData), C.c_int]
        self.L.host_basketOpt.restype = self.OptionValue
        self.L.host_cvaEquityOption.argtypes = [C.POINTER(self.CVA), C.c_int]
        self.L.host_cvaEquityOption.restype = self.OptionValue
        self.L.Chol.argtypes = [C.POINTER((R * n) * n), C.POINTER((R * n) * n)]
        self.L.Chol.restype = None
        self.L.randMinMax.argtypes = [R, R]
        self.L.randMinMax.restype = R
        self.libc = C.CDLL(None)

    def opt(self, o):
        return self.OptionData(o["s"], o["k"], o["r"], o["v"], o["t"])

    def bs_call(self, o):
        return float(self.L.host_bsCall(self.opt(o)))

    def vanilla(self, o, paths, seed):
        self.L.mcref_set_seed(seed)
        v = self.L.host_vanillaOpt(self.opt(o), paths)
        return float(v.Expected), float(v.Confidence)

    def multi(self, b):
        m = self.MultiOptionData()
        n = self.n
        for i in range(n):
            m.s[i], m.v[i], m.d[i], m.w[i] = b["s"][i], b["v"][i], b["d"][i], b["w"][i]
            for j in range(n):
                m.p[i][j] = b["p"][i][j]
        m.k, m.t, m.r = b["k"], b["t"], b["r"]
        return m

    def basket(self, b, paths, seed):
        self.L.mcref_set_seed(seed)
        m = self.multi(b)
        v = self.L.host_basketOpt(C.byref(m), paths)
        return float(v.Expected), float(v.Confidence)

    def cva(self, c, paths, seed):
        self.L.mcref_set_seed(seed)
        s = self.CVA(c["defint"], c["lgd"], 0, self.opt(c), c["n_grid"])
        v = self.L.host_cvaEquityOption(C.byref(s), paths)
        return float(v.Expected), float(v.Confidence)

    def chol(self, c):
        n = self.n
        R = CT[self.X]
        a = ((R * n) * n)()
        cc = ((R * n) * n)()
        for i in range(n):
            for j in range(n):
                cc[i][j] = c[i][j]
        self.L.Chol(C.byref(cc), C.byref(a))
        return np.array([[a[i][j] for j in range(n)] for i in range(n)], dtype=NP[self.X])

    def uniforms(self, seed, count):
        self.libc.srand(C.c_uint(seed))
        return np.array([self.L.randMinMax(0, 1) for _ in range(count)], dtype=NP[self.X])
