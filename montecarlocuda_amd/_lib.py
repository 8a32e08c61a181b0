"""ctypes declarations for libmc_mi355x.so (include/mc_mi355x.h, and the test hooks of include/mc_mi355x_test.h).

Loading is strict: if the shared library is missing the import of this module raises -- there
is no Python or CPU fallback for the simulation path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libmc_mi355x.so")
LEGACY = {"f64": os.path.join(CSRC, "libmcgpu_f64.so"), "f32": os.path.join(CSRC, "libmcgpu_f32.so")}

MC_OK = 0
MC_DEFAULT_SEED = 0x4D435F4D49333535
DOMAIN_VANILLA, DOMAIN_BASKET, DOMAIN_CVA = 1, 2, 3
MAX_ASSETS = 16          # register-resident basket kernels
MAX_ASSETS_GENERIC = 64  # LDS-staged generic kernel beyond that
NPB = {"f32": 4, "f64": 8}   # normals per block of the stream (include/mc_mi355x.h: MC_STREAM_VERSION 2)
FROM_NORMALS_NO_VOL, FROM_NORMALS_HOST_ORDER = 1, 2   # flags of the mc_*_from_normals_* test hooks
CT = {"f32": C.c_float, "f64": C.c_double}


def build(verbose: bool = False) -> None:
    """Compile every HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "all"], stdout=out)


class OptionF32(C.Structure):
    _fields_ = [("s", C.c_float), ("k", C.c_float), ("r", C.c_float), ("v", C.c_float), ("t", C.c_float)]


class OptionF64(C.Structure):
    _fields_ = [("s", C.c_double), ("k", C.c_double), ("r", C.c_double), ("v", C.c_double), ("t", C.c_double)]


def _basket(R):
    P = C.POINTER(R)

    class Basket(C.Structure):
        _fields_ = [("n", C.c_int), ("s", P), ("v", P), ("p", P), ("d", P), ("w", P), ("k", R), ("t", R), ("r", R)]
    return Basket


BasketF32, BasketF64 = _basket(C.c_float), _basket(C.c_double)


class CvaF32(C.Structure):
    _fields_ = [("defint", C.c_float), ("lgd", C.c_float), ("option", OptionF32), ("n_grid", C.c_int)]


class CvaF64(C.Structure):
    _fields_ = [("defint", C.c_double), ("lgd", C.c_double), ("option", OptionF64), ("n_grid", C.c_int)]


class Result(C.Structure):
    _fields_ = [("expected", C.c_double), ("confidence", C.c_double), ("sum", C.c_double), ("sum2", C.c_double),
                ("n", C.c_uint64), ("kernel_ms", C.c_float), ("wall_ms", C.c_float)]


class CallStats(C.Structure):   # mc_call_stats: where the last synchronous call's time went
    _fields_ = [(k, C.c_float) for k in ("setup_ms", "table_upload_ms", "launch_ms", "kernel_ms", "readback_ms", "closing_ms", "wall_ms",
                                         "context_create_ms")] + [("first_call", C.c_int)]


class Greeks(C.Structure):
    _fields_ = [("price", Result), ("delta", Result), ("vega", Result)]


class CvaGreeks(C.Structure):
    _fields_ = [("cva", Result), ("delta", Result), ("vega", Result)]


OPTION = {"f32": OptionF32, "f64": OptionF64}
BASKET = {"f32": BasketF32, "f64": BasketF64}
CVA = {"f32": CvaF32, "f64": CvaF64}

# every symbol include/mc_mi355x.h declares (the drop-in surface), then the test hooks of include/mc_mi355x_test.h;
# tests/test_abi.py checks that the .so exports each of them and that each is declared in exactly one of the two headers
EXPORTS = ["mc_last_error", "mc_device_count", "mc_device_pci_bus_id", "mc_context_create", "mc_context_destroy", "mc_context_device",
           "mc_context_blocks", "mc_context_stream", "mc_context_info", "mc_context_last_launch", "mc_context_last_call_stats", "mc_context_describe", "mc_context_profile", "mc_context_profile_read",
           "mc_context_set_antithetic", "mc_context_set_control_variate", "mc_context_set_finish", "mc_context_set_timing",
           "mc_context_order", "mc_context_idle", "mc_context_arm_direct", "mc_context_publish", "mc_context_set_generator",
           "mc_context_set_normals", "mc_context_set_cva_date_lanes", "mc_basket_control_mean_f32", "mc_basket_control_mean_f64", "mc_closing", "mc_shard_range",
           "mc_chol_f32", "mc_chol_f64", "mc_factor_from_cov_f32", "mc_factor_from_cov_f64"]
TEST_EXPORTS = ["mc_xorwow_words", "mc_grid_normals", "mc_context_set_grid_form"]
for _x in ("f32", "f64"):
    for _p in ("vanilla", "basket", "cva"):
        EXPORTS += [f"mc_{_p}_launch_{_x}", f"mc_{_p}_run_{_x}", f"mc_{_p}_paths_{_x}"]
    EXPORTS += [f"mc_{_p}_run_grid_{_x}" for _p in ("vanilla", "basket", "cva")]       # the reference's launch geometry
    EXPORTS += [f"mc_vanilla_greeks_run_{_x}", f"mc_vanilla_greeks_lr_run_{_x}", f"mc_basket_greeks_run_{_x}", f"mc_cva_greeks_run_{_x}",
                f"mc_basket_greeks_lr_run_{_x}", f"mc_cva_greeks_lr_run_{_x}"]
    TEST_EXPORTS.append(f"mc_normals_{_x}")
    TEST_EXPORTS += [f"mc_{_p}_from_normals_{_x}" for _p in ("vanilla", "basket", "cva")]
    TEST_EXPORTS += [f"mc_{_p}_paths_grid_{_x}" for _p in ("vanilla", "basket", "cva")]
GRID_FORM = {"auto": 0, "staged": 1, "fused": 2}


def _declare(L: C.CDLL) -> C.CDLL:
    ctx = C.c_void_p
    u64 = C.c_uint64
    L.mc_last_error.restype = C.c_char_p
    L.mc_last_error.argtypes = []
    L.mc_device_count.restype = C.c_int
    L.mc_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.mc_context_create.argtypes = [C.c_int, C.c_int, C.POINTER(ctx)]
    L.mc_context_destroy.argtypes = [ctx]
    L.mc_context_destroy.restype = None
    L.mc_context_device.argtypes = [ctx]
    L.mc_context_blocks.argtypes = [ctx]
    L.mc_context_stream.argtypes = [ctx]
    L.mc_context_stream.restype = C.c_void_p
    L.mc_context_info.argtypes = [ctx, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mc_context_last_launch.argtypes = [ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mc_context_last_call_stats.argtypes = [ctx, C.POINTER(CallStats)]
    L.mc_context_describe.argtypes = [ctx, C.c_char_p, C.c_int]
    L.mc_context_set_antithetic.argtypes = [ctx, C.c_int]
    L.mc_context_set_control_variate.argtypes = [ctx, C.c_int]
    L.mc_context_set_finish.argtypes = [ctx, C.c_int]
    L.mc_context_set_timing.argtypes = [ctx, C.c_int]
    L.mc_context_order.argtypes = [ctx, C.c_void_p]
    L.mc_context_idle.argtypes = [ctx]
    L.mc_context_arm_direct.argtypes = [ctx, C.POINTER(C.POINTER(C.c_double))]
    L.mc_context_publish.argtypes = [ctx, C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(C.c_double))]
    L.mc_context_set_generator.argtypes = [ctx, C.c_int, C.c_uint64]
    L.mc_context_set_normals.argtypes = [ctx, C.c_int]
    L.mc_context_set_cva_date_lanes.argtypes = [ctx, C.c_int]
    L.mc_grid_normals.argtypes = [ctx, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_float)]
    L.mc_context_set_grid_form.argtypes = [ctx, C.c_int]
    L.mc_xorwow_words.argtypes = [ctx, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    for X in ("f32", "f64"):
        getattr(L, f"mc_basket_control_mean_{X}").argtypes = [C.POINTER(BASKET[X]), C.POINTER(C.c_double)]
    L.mc_context_profile.argtypes = [ctx, C.c_int]
    L.mc_context_profile_read.argtypes = [ctx, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mc_closing.argtypes = [C.c_double, C.c_double, u64, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mc_closing.restype = None
    L.mc_shard_range.argtypes = [u64, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64)]
    L.mc_shard_range.restype = None
    for X in ("f32", "f64"):
        R = CT[X]
        RP = C.POINTER(R)
        getattr(L, f"mc_chol_{X}").argtypes = [C.c_int, RP, RP]
        getattr(L, f"mc_factor_from_cov_{X}").argtypes = [C.c_int, RP, RP, RP]
        for prod, S in (("vanilla", OPTION[X]), ("basket", BASKET[X]), ("cva", CVA[X])):
            getattr(L, f"mc_{prod}_launch_{X}").argtypes = [ctx, C.POINTER(S), u64, u64, u64, C.c_void_p, C.c_void_p]
            getattr(L, f"mc_{prod}_run_{X}").argtypes = [ctx, C.POINTER(S), u64, u64, u64, C.POINTER(Result)]
            getattr(L, f"mc_{prod}_paths_{X}").argtypes = [ctx, C.POINTER(S), u64, u64, u64, RP]
            getattr(L, f"mc_{prod}_run_grid_{X}").argtypes = [ctx, C.POINTER(S), C.c_int, C.c_int, u64, C.POINTER(Result)]
            getattr(L, f"mc_{prod}_paths_grid_{X}").argtypes = [ctx, C.POINTER(S), C.c_int, C.c_int, u64, RP]
        getattr(L, f"mc_normals_{X}").argtypes = [ctx, u64, C.c_uint32, u64, u64, C.c_uint32, RP]
        getattr(L, f"mc_vanilla_from_normals_{X}").argtypes = [ctx, C.POINTER(OPTION[X]), RP, u64, RP, C.POINTER(Result)]
        getattr(L, f"mc_basket_from_normals_{X}").argtypes = [ctx, C.POINTER(BASKET[X]), RP, u64, C.c_int, RP, C.POINTER(Result)]
        getattr(L, f"mc_cva_from_normals_{X}").argtypes = [ctx, C.POINTER(CVA[X]), RP, u64, C.c_int, RP, C.POINTER(Result)]
        getattr(L, f"mc_vanilla_greeks_run_{X}").argtypes = [ctx, C.POINTER(OPTION[X]), u64, u64, u64, C.POINTER(Greeks)]
        getattr(L, f"mc_vanilla_greeks_lr_run_{X}").argtypes = [ctx, C.POINTER(OPTION[X]), u64, u64, u64, C.POINTER(Greeks)]
        getattr(L, f"mc_basket_greeks_run_{X}").argtypes = [ctx, C.POINTER(BASKET[X]), u64, u64, u64, C.POINTER(Result), C.POINTER(Result),
                                                            C.POINTER(Result)]
        getattr(L, f"mc_cva_greeks_run_{X}").argtypes = [ctx, C.POINTER(CVA[X]), u64, u64, u64, C.POINTER(CvaGreeks)]
        getattr(L, f"mc_basket_greeks_lr_run_{X}").argtypes = [ctx, C.POINTER(BASKET[X]), u64, u64, u64, C.POINTER(Result), C.POINTER(Result),
                                                               C.POINTER(Result)]
        getattr(L, f"mc_cva_greeks_lr_run_{X}").argtypes = [ctx, C.POINTER(CVA[X]), u64, u64, u64, C.POINTER(CvaGreeks)]
    return L


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP engine has no fallback)")
        _LIB = _declare(C.CDLL(LIB_PATH))
    return _LIB


class McError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != MC_OK:
        raise McError(f"mc error {rc}: {lib().mc_last_error().decode()}")
