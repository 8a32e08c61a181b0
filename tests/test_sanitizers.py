"""AddressSanitizer + UndefinedBehaviorSanitizer over all CPU-side C of the repo (oracle, host_path.c,
the host helpers of the C ABI are exercised through host_path's mc_chol/mc_closing calls).  GPU ASan
is not available on the pool, so kernels are covered by the parity tests instead."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlocuda_amd", "csrc")


@pytest.mark.parametrize("flag,n", [("", 3), ("-DMC_SINGLE_PRECISION", 3), ("", 16)])
def test_cpu_code_is_clean_under_asan_ubsan(tmp_path, flag, n):
    if not os.path.exists(os.path.join(CSRC, "libmc_mi355x.so")):
        subprocess.check_call(["make", "-C", CSRC, "all"], stdout=subprocess.DEVNULL)
    exe = tmp_path / "sanitize_cpu"
    san = ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    # host_simd.c, the vectorised vanilla loop: its three per-ISA copies, sanitised too (with the product's own -O3
    # -ffast-math, which is what makes it vector code; the dispatcher only ever calls a copy the CPU can run)
    simd = []
    for name, march in (("base", ["-march=x86-64"]), ("avx2", ["-march=haswell"]),
                        ("avx512", ["-march=skylake-avx512", "-mprefer-vector-width=512"])):
        obj = tmp_path / f"host_simd_{name}.o"
        subprocess.check_call(["gcc", "-std=gnu11", "-O3", "-ffast-math", "-fopenmp-simd", *san, *march, *([flag] if flag else []),
                               f"-DMC_SIMD_SUFFIX={name}", f"-DN={n}", f"-I{ROOT}/include", "-c", "-o", str(obj),
                               os.path.join(CSRC, "host_simd.c")])
        simd.append(str(obj))
    cmd = ["gcc", "-std=gnu11", "-O1", *san, "-fopenmp",
           f"-DN={n}", f"-I{ROOT}/include", f"-I{ROOT}/oracle",
           os.path.join(ROOT, "tests", "c", "sanitize_cpu.c"), os.path.join(ROOT, "oracle", "mc_oracle.c"),
           os.path.join(CSRC, "host_path.c"), *simd, "-o", str(exe), f"-L{CSRC}", "-lmc_mi355x", f"-Wl,-rpath,{CSRC}", "-lmvec", "-lm"]
    if flag:
        cmd.insert(1, flag)
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               MC_HOST_THREADS="4")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    assert "twin cva" in out.stdout
