"""The C-ABI boundary, checked without a GPU: the library loads, exports every symbol
include/mc_mi355x.h declares, the legacy libraries export the reference's three entry points,
and include/MonteCarlo.h has the reference's struct layouts (measured with gcc on
the reference header: dp 40/192/16/72, sp 20/96/8/36 for N=3)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


@pytest.fixture(scope="module")
def mc():
    import montecarlocuda_amd as mc
    if not os.path.exists(mc._lib.LIB_PATH):
        mc.build()
    return mc


def _declared(header):
    text = open(os.path.join(INC, header)).read()
    return set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", text)) - {"mc_context"}


def test_library_exports_every_declared_symbol(mc):
    """The product header declares the drop-in surface, the test header the hooks; the .so exports every one of them."""
    L = mc._lib.lib()
    product, hooks = _declared("mc_mi355x.h"), _declared("mc_mi355x_test.h")
    assert product == set(mc._lib.EXPORTS), product ^ set(mc._lib.EXPORTS)
    assert hooks == set(mc._lib.TEST_EXPORTS), hooks ^ set(mc._lib.TEST_EXPORTS)
    for name in sorted(product | hooks):
        assert hasattr(L, name), name


def test_every_exported_symbol_is_declared_in_exactly_one_header(mc):
    """nm on the shipping library: each exported mc_* function is declared by the product header or by the test-hook header,
    never by both, never by neither -- and the product header advertises no test hook (the switches that
    reproduce the reference's CPU-path bugs are not part of the public surface)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", mc._lib.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("mc_")}
    exported -= {"mc_internal_fail"}      # the error text shared with the CPU twin's object (mc_hostmath.c), not an entry point
    product, hooks = _declared("mc_mi355x.h"), _declared("mc_mi355x_test.h")
    assert not (product & hooks), product & hooks
    assert exported == product | hooks, exported ^ (product | hooks)
    text = open(os.path.join(INC, "mc_mi355x.h")).read()
    for word in ("from_normals", "MC_FROM_NORMALS", "mc_grid_normals", "mc_normals_f", "NO_VOL", "HOST_ORDER"):
        assert word not in text, word
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()      # lists the product header only
    for word in ("from_normals", "mc_mi355x_test.h", "mc_grid_normals", "NO_VOL"):
        assert word not in integration, word


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_legacy_library_exports_reference_entry_points(mc, X):
    path = mc._lib.LEGACY[X]
    assert os.path.exists(path)
    L = C.CDLL(path)
    for name in ("dev_vanillaOpt", "dev_basketOpt", "dev_cvaEquityOption"):
        assert hasattr(L, name)
        assert hasattr(L, name + "_ex")     # explicit-seed variants (SURVEY 8b "RNG contract"; include/MonteCarlo.h)
    # the host entry points the reference's drivers link (MonteCarloHost.c:139,282,292,302,90,42,51) + their _ex variants
    H = C.CDLL(os.path.join(os.path.dirname(path), f"libmchost_{X}.so"))
    for name in ("host_bsCall", "host_vanillaOpt", "host_basketOpt", "host_cvaEquityOption", "Chol", "printOption", "printMultiOpt",
                 "printVect", "printMat", "prodMat", "randMinMax", "host_vanillaOpt_ex", "host_basketOpt_ex", "host_cvaEquityOption_ex"):
        assert hasattr(H, name), name


LAYOUT_C = r"""
#include <stddef.h>
#include "MonteCarlo.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(OptionData), sizeof(MultiOptionData), sizeof(OptionValue),
         sizeof(CVA), offsetof(CVA, option), offsetof(CVA, n), sizeof(MonteCarloData));
  return 0; }
"""

# (precision flag, N) -> sizes measured on the REFERENCE header (SURVEY 8a "Types")
EXPECTED = {("", 3): (40, 192, 16, 72, 24, 64, 256), ("-DMC_SINGLE_PRECISION", 3): (20, 96, 8, 36, 12, 32, 132),
            ("", 4): (40, 280, 16, 72, 24, 64, None), ("-DMC_SINGLE_PRECISION", 4): (20, 140, 8, 36, 12, 32, None),
            ("", 16): (40, 2584, 16, 72, 24, 64, None), ("-DMC_SINGLE_PRECISION", 16): (20, 1292, 8, 36, 12, 32, None)}


@pytest.mark.parametrize("key", sorted(EXPECTED))
def test_dropin_header_layout(tmp_path, key):
    flag, n = key
    src = tmp_path / "layout.c"
    src.write_text(LAYOUT_C)
    exe = tmp_path / "layout"
    cmd = ["gcc", "-std=c11", f"-I{INC}", f"-DN={n}", str(src), "-o", str(exe)] + ([flag] if flag else [])
    subprocess.check_call(cmd)
    got = tuple(int(x) for x in subprocess.check_output([str(exe)]).split())
    want = EXPECTED[key]
    for g, w in zip(got, want):
        if w is not None:
            assert g == w, (key, got, want)


def test_layout_matches_live_reference_header(tmp_path):
    """Where the reference is mounted, compile the same probe against ITS header and compare."""
    ref = "/root/reference/double_precision"
    if not os.path.isdir(ref):
        pytest.skip("reference not mounted")
    src = tmp_path / "layout.c"
    src.write_text(LAYOUT_C)
    outs = []
    for inc, flags in ((ref, []), (INC, [])), (("/root/reference/single_precision", []), (INC, ["-DMC_SINGLE_PRECISION"])):
        pair = []
        for i, f in (inc, flags):
            exe = tmp_path / "p"
            subprocess.check_call(["gcc", "-w", f"-I{i}", str(src), "-o", str(exe)] + f)
            pair.append(subprocess.check_output([str(exe)]))
        outs.append(pair)
    for a, b in outs:
        assert a == b


def test_no_gpu_is_a_loud_error(mc):
    """Without a device the engine refuses to run -- there is no CPU fallback to fall into."""
    if mc._lib.lib().mc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(mc.McError, match="no HIP device"):
        mc.Engine(0)


MULTI_LIB = os.path.join(ROOT, "montecarlocuda_amd", "csrc", "libmc_multi.so")


def test_multi_library_exports_every_declared_symbol(mc):
    """libmc_multi.so (several GPUs from one process, RCCL) loads without a GPU -- it links librccl and
    libamdhip64 directly -- and exports every symbol include/mc_multi.h declares."""
    assert os.path.exists(MULTI_LIB)
    L = C.CDLL(MULTI_LIB)
    header = open(os.path.join(INC, "mc_multi.h")).read()
    declared = set(re.findall(r"\b(mc_multi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) == 25, sorted(declared)      # round 6: + mc_multi_last_collective_us, mc_multi_last_device_us
    for name in sorted(declared):
        assert hasattr(L, name), name
    needed = subprocess.check_output(["readelf", "-d", MULTI_LIB], text=True)
    assert "librccl.so" in needed and "libmc_mi355x.so" in needed      # RCCL's C API, linked directly
    assert "torch" not in needed


def test_multi_without_a_gpu_is_a_loud_error(mc):
    if mc._lib.lib().mc_device_count() > 0:
        pytest.skip("a GPU is visible")
    L = C.CDLL(MULTI_LIB)
    L.mc_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mc_multi_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert L.mc_multi_create(None, 0, 0, C.byref(h)) == 3      # MC_ERR_NO_DEVICE
    assert b"no HIP device" in L.mc_multi_last_error() and not h.value


def test_legacy_library_does_not_pull_in_rccl(mc):
    """The single-GPU drop-in libraries stay light: RCCL (a 570 MB library) is loaded only when MC_DEVICES asks
    for several GPUs (legacy_abi.c dlopens libmc_multi.so then)."""
    for X in ("f32", "f64"):
        needed = subprocess.check_output(["readelf", "-d", mc._lib.LEGACY[X]], text=True)
        assert "librccl" not in needed and "libmc_multi" not in needed and "libmc_mi355x.so" in needed


def test_every_environment_variable_the_libraries_read_is_in_the_table():
    """INTEGRATION.md section 1 holds ONE table of the knobs: every name the product sources pass to getenv() or to
    the env_int() helper must have a row there, and the table must not list names nobody reads.  The test-only switch of
    libmc_multi is compiled in under -DMC_MULTI_TEST_HOOKS only and named below the table, not in it."""
    import glob
    import re
    csrc = os.path.join(ROOT, "montecarlocuda_amd", "csrc")
    read = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.[ch]*")) + glob.glob(os.path.join(ROOT, "drivers", "*.[ch]"))):
        if path.endswith((".so", ".o", ".inc")):
            continue
        text = open(path, errors="replace").read()
        for name in re.findall(r'(?:getenv|env_int)\("(MC_[A-Z0-9_]+)"', text):
            read.setdefault(name, set()).add(os.path.basename(path))
    assert len(read) >= 20, read
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rows = set()
    for line in doc.splitlines():
        if line.startswith("| `MC_"):
            rows |= set(re.findall(r"`(MC_[A-Z0-9_]+)`", line.split("|")[1]))
    test_only = {"MC_MULTI_ALLOW_REPEATED_DEVICES"}
    missing = sorted(set(read) - rows - test_only)
    assert not missing, f"read by the sources but not in INTEGRATION.md's table: { {m: sorted(read[m]) for m in missing} }"
    stale = sorted(rows - set(read))
    assert not stale, f"in the table but read by nobody: {stale}"
    # the test-only switch: named in the text, guarded by the macro in the source, absent from the shipped binary's strings
    assert "MC_MULTI_ALLOW_REPEATED_DEVICES" in doc and read["MC_MULTI_ALLOW_REPEATED_DEVICES"] == {"mc_multi.cpp"}
    src = open(os.path.join(csrc, "mc_multi.cpp")).read()
    guard = src.index("#ifdef MC_MULTI_TEST_HOOKS")
    assert guard < src.index('getenv("MC_MULTI_ALLOW_REPEATED_DEVICES")') < src.index("#endif", guard)
    assert b"MC_MULTI_ALLOW_REPEATED_DEVICES" not in open(os.path.join(csrc, "libmc_multi.so"), "rb").read()
