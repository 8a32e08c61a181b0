"""The plain-C drivers (drivers/*.c) end to end on a real MI355X: host code in C calls
dev_* (libmcgpu) and host_* (libmchost); CPU and GPU legs share the Philox stream, so for the same
paths they agree to rounding -- a stronger check than the reference's driver could print."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 10.386270784322328


def run(name, *args):
    exe = os.path.join(ROOT, "drivers", name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    return out.stdout


def floats_after(text, marker, count):
    tail = text[text.index(marker) + len(marker):]
    return [float(x) for x in re.findall(r"^-?\d+\.\d+", tail, flags=re.M)[:count]]


@pytest.mark.parametrize("X,tol", [("f64", 1e-9), ("f32", 5e-5)])
def test_vanilla_driver(X, tol):
    out = run(f"vanillaOpt_{X}", "64")           # 64 x 131072 = 8.4e6 paths
    assert "Black & Scholes price: 10.3862" in out
    cpu = floats_after(out, "time [s]\n", 4)      # price, CI, |price-BS|, time
    gpu = floats_after(out, "Speedup :\n", 4)      # (the threads line is an int) price, CI, diff, time
    price_gpu, ci_gpu, diff_gpu = gpu[0], gpu[1], gpu[2]
    assert abs(price_gpu - BS) < 3.5 / 1.96 * ci_gpu + 1e-5
    assert abs(diff_gpu - abs(price_gpu - 10.386262)) < 2e-5     # printed diff is vs the Hastings-CDF closed form
    assert abs(cpu[0] - price_gpu) <= max(tol * price_gpu, 2e-6)  # same stream: CPU == GPU to rounding / print precision
    assert abs(cpu[1] - ci_gpu) <= 2e-6


def test_vanilla_driver_under_the_reference_launch_geometry():
    """MC_RNG=xorwow_grid: the unchanged C driver, through dev_vanillaOpt(&option, 512, 128, sims), prices the sample the
    reference's 512 x 128 launch draws (mc_vanilla_run_grid_f64 from Python gives the same number)."""
    import montecarlocuda_amd as mc
    exe = os.path.join(ROOT, "drivers", "vanillaOpt_f64")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "8", "--no-cpu"], capture_output=True, text=True, timeout=600, env=dict(os.environ, MC_RNG="xorwow_grid"))
    assert out.returncode == 0, out.stderr
    gpu = floats_after(out.stdout, "Speedup :\n", 4)
    with mc.Engine(0) as eng:
        e = eng.run_grid("vanilla", dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0), 512, 128, 8 * 131072 // 512, "f64")
    assert abs(gpu[0] - e.expected) < 1e-6 and abs(gpu[1] - e.confidence) < 1e-6
    assert abs(gpu[0] - BS) < 3.5 / 1.96 * gpu[1]


def test_basket_driver_reference_data():
    out = run("basketOpt_f64", "16")
    assert "zero pivot" in out                   # the reference's N=3 correlation matrix is singular
    cpu = floats_after(out, "Expected price, I.C., time [s]\n", 3)
    gpu = floats_after(out, "Speedup :\n", 4)
    assert abs(cpu[0] - gpu[0]) < 2e-6 and abs(cpu[1] - gpu[1]) < 2e-6
    assert 4.5 < gpu[0] < 5.0                    # reference sp host at N=3: 4.76


def test_cva_driver_all_grids():
    out = run("cvaOpt_f64", "1", "--no-cpu")
    assert out.count("--- exposure dates:") == 5
    vals = [float(x) for x in re.findall(r"^(0\.1[89]\d+) $", out, flags=re.M)]
    assert len(vals) >= 20 and all(0.17 < v < 0.21 for v in vals)


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_basket_driver_with_sixteen_assets(X, tmp_path):
    """The reference's "edit #define N" as a build variable: legacy and host libraries plus the basket driver built
    for N = 16 in a scratch directory (the in-tree N = 3 builds stay untouched), linked against the in-tree engine.
    The GPU leg then runs the tiled kernels; CPU twin and GPU agree to rounding on the shared stream."""
    csrc, inc = os.path.join(ROOT, "montecarlocuda_amd", "csrc"), os.path.join(ROOT, "include")
    prec = [] if X == "f64" else ["-DMC_SINGLE_PRECISION"]
    common = ["-DN=16", f"-I{inc}", *prec, f"-L{csrc}", "-lmc_mi355x", f"-Wl,-rpath,{csrc}"]
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", str(tmp_path / f"libmcgpu_{X}.so"),
                           os.path.join(csrc, "legacy_abi.c"), *common])
    simd = []   # host_simd.c: one copy per vector ISA, as the Makefile builds them
    for name, march in (("base", ["-march=x86-64"]), ("avx2", ["-march=haswell"]),
                        ("avx512", ["-march=skylake-avx512", "-mprefer-vector-width=512"])):
        obj = tmp_path / f"host_simd_{name}.o"
        subprocess.check_call(["gcc", "-O3", "-ffast-math", "-fopenmp-simd", "-std=gnu11", "-fPIC", *march, *prec, "-DN=16",
                               f"-DMC_SIMD_SUFFIX={name}", f"-I{inc}", "-c", "-o", str(obj), os.path.join(csrc, "host_simd.c")])
        simd.append(str(obj))
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-o",
                           str(tmp_path / f"libmchost_{X}.so"), os.path.join(csrc, "host_path.c"), *simd, *common, "-lmvec", "-lm"])
    exe = tmp_path / f"basketOpt16_{X}"
    # the scratch directory comes first on the link line and in the run path: csrc holds N = 3 builds of the same names
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-DN=16", f"-I{inc}", *prec, f"-I{os.path.join(ROOT, 'drivers')}", "-o", str(exe),
                           os.path.join(ROOT, "drivers", "basketOpt.c"), f"-L{tmp_path}", f"-Wl,-rpath,{tmp_path}", f"-lmcgpu_{X}",
                           f"-lmchost_{X}", f"-L{csrc}", f"-Wl,-rpath,{csrc}", "-lmc_mi355x", "-lm"])
    out = subprocess.run([str(exe), "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "Basket Option with 16 underlyings" in out.stdout and "zero pivot" not in out.stdout
    cpu = floats_after(out.stdout, "Expected price, I.C., time [s]\n", 3)
    gpu = floats_after(out.stdout, "Speedup :\n", 4)
    tol = 2e-6 if X == "f64" else 2e-4
    assert abs(cpu[0] - gpu[0]) < tol and abs(cpu[1] - gpu[1]) < tol
    assert 0.0 < gpu[0] < 100.0 and gpu[1] > 0
