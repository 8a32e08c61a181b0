"""The engine's generator IS rocRAND's: Philox4x32-10 words from rocRAND's own (host-callable) engine
for this repo's counter layout equal the oracle's, which the HIP kernels are tested against on the GPU.
(The oracle is also pinned by the Random123 known-answer vectors: test_oracle_golden.py.)"""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_philox_equals_rocrand_engine(po, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_philox4x32_10.h"):
        pytest.skip("rocRAND headers / hipcc not available")
    exe = tmp_path / "rocrand_xcheck"
    subprocess.check_call([hipcc, "-O1", "-w", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "rocrand_xcheck.cpp"),
                           "-o", str(exe)])
    rng = np.random.default_rng(7)
    # (seed, unit, block, domain); rocRAND's offset is 4 * (counter words 1:0), which must fit 64 bits: the
    # engine keeps a unit's LOW word in counter word 1 (po.counter), so the low word stays below 2^30 here
    cases = [(0, 0, 0, 0), (0x4D435F4D49333535, 0, 0, 1), (0x4D435F4D49333535, 24999999, 0, 1), (1, (1 << 30) - 1, 3, 2),
             (2 ** 64 - 1, (0xFFFFFFFF << 32) + 77, 63, 3), (12345, (7 << 32) + 5, 0, 1)]
    for _ in range(300):
        cases.append((int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2)),
                      (int(rng.integers(0, 2 ** 32)) << 32) | int(rng.integers(0, 2 ** 30)),
                      int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31))))
    ctrs = [po.counter(u, b, d) for _, u, b, d in cases]
    text = "".join(f"{s} {(c[1] << 32) | c[0]} {c[2]} {c[3]}\n" for (s, _, _, _), c in zip(cases, ctrs))
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    for (s, u, b, d), c, line in zip(cases, ctrs, out):
        want = [int(x, 16) for x in line.split()]
        assert len(want) == 4, (line, s, u, b, d)
        assert po.philox(c, [s & 0xFFFFFFFF, s >> 32]) == want, (s, u, b, d)
        # the oracle's normals are built from exactly these words (f32: word k -> uniform k)
        z = po.dev_normals("f32", s, d, u, b)
        ua = np.float32(np.float32(want[0]) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))
        if ua < 1:
            r2 = float(z[0]) ** 2 + float(z[1]) ** 2
            assert r2 == pytest.approx(-2.0 * np.log(float(ua)), rel=2e-5), (s, u, b, d)


def test_oracle_xorwow_equals_rocrand_engine(po, tmp_path):
    """XORWOW, the reference's generator (curand_init / curand_normal, dp/MonteCarloKernel.cu:285-290,68): the oracle's
    words -- seeding by rocRAND's rule, subsequence jump by GF(2) matrices the oracle computes itself -- equal those of
    rocRAND's own host-callable engine, rocrand_init(seed, subsequence, 0) + rocrand(), for small and huge subsequence
    numbers.  The HIP engine's XORWOW is compared with the oracle on the GPU (tests/test_gpu_xorwow.py)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_xorwow.h"):
        pytest.skip("rocRAND headers / hipcc not available")
    exe = tmp_path / "rocrand_xorwow_xcheck"
    subprocess.check_call([hipcc, "-O1", "-w", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "rocrand_xorwow_xcheck.cpp"),
                           "-o", str(exe)])
    rng = np.random.default_rng(11)
    cases = [(0, 0, 8), (0x4D435F4D49333535, 0, 8), (0x4D435F4D49333535, 1, 8), (1, 255, 4), (1, 256, 4), (12345, 524287, 4),
             (2 ** 64 - 1, 2 ** 32 - 1, 4), (7, 2 ** 48 - 1, 4)]
    for _ in range(200):
        cases.append((int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 2 ** 47)), 3))
    text = "".join(f"{s} {sub} {c}\n" for s, sub, c in cases)
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(cases)
    for (s, sub, c), line in zip(cases, out):
        assert po.xorwow_words(s, sub, c) == [int(x, 16) for x in line.split()], (s, sub)


def test_oracle_grid_normals_equal_rocrand_normal(po, tmp_path):
    """The reference's per-thread normal stream under its launch geometry (dp/MonteCarloKernel.cu:285-290,68: curand_init(
    blockIdx.x + gridDim.x, threadIdx.x, 0) + curand_normal): the oracle's restatement (orc_grid_normals) against
    rocRAND's own rocrand_init + rocrand_normal -- what a HIP build of the reference calls -- bit for bit, odd counts
    (the kept second member of a Box-Muller pair) included."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_normal.h"):
        pytest.skip("rocRAND headers / hipcc not available")
    exe = tmp_path / "rocrand_normal_xcheck"
    subprocess.check_call([hipcc, "-O1", "-w", "-ffp-contract=off", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tests", "cpp", "rocrand_normal_xcheck.cpp"), "-o", str(exe)])
    count = 9
    geoms = [(1, 1), (2, 3), (7, 64), (256, 5)]
    cases = [(G, T, b, t) for G, T in geoms for b in sorted({0, G // 2, G - 1}) for t in sorted({0, T // 2, T - 1})]
    text = "".join(f"{b + G} {t} {count}\n" for G, T, b, t in cases)
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(cases)
    streams = {(G, T): po.grid_normals(G, T, count) for G, T in geoms}
    for (G, T, b, t), line in zip(cases, out):
        want = np.array([int(x, 16) for x in line.split()], dtype=np.uint32)
        got = streams[(G, T)][b, t].view(np.uint32)
        assert (got == want).all(), (G, T, b, t)
    # the arrangement helper: thread t of a block takes paths t, t + T, ...; a path's draws are consecutive
    s = streams[(2, 3)]
    z = po.grid_path_normals(s, 4, 2)     # 4 paths per block, 2 draws each: thread 0 prices paths 0 and 3
    assert z.shape == (8, 2)
    assert (z[0] == s[0, 0, 0:2]).all() and (z[3] == s[0, 0, 2:4]).all() and (z[1] == s[0, 1, 0:2]).all()
    assert (z[4 + 2] == s[1, 2, 0:2]).all()
    assert po.grid_draws_per_thread(3, 4, 2) == 4
