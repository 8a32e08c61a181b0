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
    cases = [(0, 0, 0, 0), (0x4D435F4D49333535, 0, 0, 1), (0x4D435F4D49333535, 24999999, 0, 1), (1, (1 << 32) - 1, 3, 2),
             (2 ** 64 - 1, (1 << 32), 63, 3), (12345, (7 << 32) + 5, 0, 1)]
    for _ in range(300):
        cases.append((int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 2 ** 61)),
                      int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31))))
    text = "".join(f"{s} {u} {b} {d}\n" for s, u, b, d in cases)
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    for (s, u, b, d), line in zip(cases, out):
        want = [int(x, 16) for x in line.split()]
        got = po.philox([u & 0xFFFFFFFF, u >> 32, b, d], [s & 0xFFFFFFFF, s >> 32])
        assert got == want, (s, u, b, d)
