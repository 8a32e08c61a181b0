#!/usr/bin/env python3
"""Distribution-level checks of the engine's normals, independent of the oracle (which mirrors the stream definition):
moments, a Kolmogorov-Smirnov test against N(0,1), the correlation matrix of the normals of one block (fp64 stream
version 2 builds four Box-Muller pairs from twelve Philox words and splits the middle word of each triple between
radius and angle: any shared bit would show up here), radius-angle independence inside a pair, lag correlations
across consecutive units.  Also for the fp32 stream, the fp32-normals mode, XORWOW, and the launch-geometry streams.
    python tools/normal_stats.py [units]"""
import os
import sys

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc  # noqa: E402

units = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
SEED = 0x4D435F4D49333535
bad = 0


def report(name, z):
    """z: (units, npb) float64"""
    global bad
    n = z.size
    flat = z.reshape(-1)
    m, v = flat.mean(), flat.var()
    sk, ku = stats.skew(flat), stats.kurtosis(flat)
    ks = stats.kstest(flat[: min(n, 4_000_000)], "norm")
    c = np.corrcoef(z.T) if z.shape[1] > 1 else np.ones((1, 1))
    off = np.abs(c - np.eye(c.shape[0])).max() if c.shape[0] > 1 else 0.0
    lag = float(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1])
    se = 1.0 / np.sqrt(z.shape[0])
    # inside a pair: radius^2 ~ chi2(2) and angle ~ U(-pi, pi), independent
    r2 = z[:, 0] ** 2 + z[:, 1] ** 2 if z.shape[1] > 1 else None
    line = (f"{name:28s} n={n:.2e} mean {m:+.2e} var-1 {v - 1:+.2e} skew {sk:+.2e} exkurt {ku:+.2e} KS p={ks.pvalue:.3f} "
            f"max|corr| in block {off:.2e} lag-1 {lag:+.2e} (1 sigma = {se:.1e})")
    ok = abs(m) < 5 / np.sqrt(n) and abs(v - 1) < 5 * np.sqrt(2 / n) and abs(sk) < 5 * np.sqrt(6 / n) and abs(ku) < 5 * np.sqrt(24 / n) \
        and ks.pvalue > 1e-4 and off < 5.5 * se and abs(lag) < 5 * se
    if r2 is not None:
        ang = np.arctan2(z[:, 1], z[:, 0])
        rc = float(np.corrcoef(r2, ang)[0, 1])
        rc2 = float(np.corrcoef(r2, np.cos(4 * ang))[0, 1])
        ksr = stats.kstest(r2[:2_000_000], "chi2", args=(2,))
        ksa = stats.kstest(ang[:2_000_000], "uniform", args=(-np.pi, 2 * np.pi))
        line += f" | pair: corr(r2, angle) {rc:+.2e}, corr(r2, cos 4 angle) {rc2:+.2e}, KS r2~chi2(2) p={ksr.pvalue:.3f}, KS angle~U p={ksa.pvalue:.3f}"
        ok = ok and abs(rc) < 5 * se and abs(rc2) < 5 * se and ksr.pvalue > 1e-4 and ksa.pvalue > 1e-4
    print(line + ("" if ok else "   <-- OUTSIDE 5 sigma"))
    bad += 0 if ok else 1


eng = mc.Engine(0)
chunk = 500_000


def gather(e, domain, X, block):
    return np.concatenate([e.normals(SEED, domain, u0, min(chunk, units - u0), block, X) for u0 in range(0, units, chunk)]).astype(np.float64)


for block in (0, 5):
    report(f"philox f64 block {block}", gather(eng, 2, "f64", block))
    report(f"philox f32 block {block}", gather(eng, 2, "f32", block))
eng.set_normals("f32")
report("philox f64 on f32 normals", gather(eng, 3, "f64", 1))
eng.set_normals("native")
# launch-geometry streams: 512 x 128 threads, 32 consecutive normals each -> rows of one thread
g = eng.grid_normals(512, 128, 32).reshape(-1, 32).astype(np.float64)
report("xorwow grid 512x128 streams", g)
eng.close()
print("violations:", bad)
sys.exit(1 if bad else 0)
