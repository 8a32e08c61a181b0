"""The library's Makefile rule must list every local file the translation unit includes: a header missing from the
prerequisites means `make` keeps a stale libmc_mi355x.so after an edit to it (that happened once with
mc_math_f64.hpp, and a fix was "tested" against the old binary)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlocuda_amd", "csrc")


def local_includes(path, seen):
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        full = os.path.join(CSRC, name)
        if os.path.exists(full) and name not in seen:
            seen.add(name)
            local_includes(full, seen)
    return seen


def test_makefile_lists_every_included_file():
    needed = local_includes(os.path.join(CSRC, "mc_api.hip"), set())
    assert {"mc_kernels.hpp", "mc_rng.hpp", "mc_math_f64.hpp", "mc_tables_f64.inc", "mc_reduce.hpp"} <= needed
    text = open(os.path.join(CSRC, "Makefile")).read()
    inc = re.search(r"^INC\s*:?=\s*(\S+)", text, flags=re.M).group(1)
    rule = re.search(r"^libmc_mi355x\.so:(.*)$", text, flags=re.M).group(1).replace("$(INC)", inc).split()
    missing = sorted(n for n in needed if n not in rule)
    assert not missing, f"Makefile rule for libmc_mi355x.so does not depend on {missing}"


def test_cpu_twin_needs_no_rocm_runtime():
    """libmchost_* is the no-GPU path, so it must load where ROCm is absent -- its dynamic section names
    neither the GPU engine nor the HIP runtime (the shared host math is linked in as an object: mc_hostmath.c)."""
    import subprocess
    for X in ("f32", "f64"):
        lib = os.path.join(CSRC, f"libmchost_{X}.so")
        assert os.path.exists(lib), f"{lib} not built"
        dyn = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
        needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", dyn)
        assert needed, dyn
        bad = [n for n in needed if "mc_mi355x" in n or "amdhip" in n or "hsa" in n or "rccl" in n]
        assert not bad, f"libmchost_{X}.so depends on {bad}"
