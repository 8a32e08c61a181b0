#!/usr/bin/env python3
"""How far does fp32 simulation bias the price?  1e12 paths (CI ~ 3e-5) in both precisions against exact
Black-Scholes.  DESIGN.md section 5 bounds the fp32 constant-rounding bias at ~5e-6."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, montecarlocuda_amd as mc
BS = bench.BS_EXACT
eng = mc.Engine(0)
for X, total, chunk, normals in (("f32", 10 ** 12, 5 * 10 ** 10, "native"), ("f64", 2 * 10 ** 11, 2 * 10 ** 10, "native"),
                                 ("f64", 4 * 10 ** 11, 2 * 10 ** 10, "f32")):   # last: the reference's dp arithmetic (MC_NORMALS_F32)
    eng.set_normals(normals)
    for anti in (False, True):
        eng.set_antithetic(anti)
        s = s2 = 0.0
        n = 0
        ms = 0.0
        t0 = time.time()
        for i in range(total // chunk):
            e = eng.vanilla(bench.VAN, chunk, mc.MC_DEFAULT_SEED + 17, i * chunk, X)
            s, s2, n, ms = s + e.sum, s2 + e.sum2, n + e.n, ms + e.kernel_ms
        r, t = (float(__import__("numpy").float32(bench.VAN[k])) if X == "f32" else bench.VAN[k] for k in ("r", "t"))
        price, ci = mc.closing(s, s2, n, math.exp(-r * t))
        print(f"{X}{'/n32' if normals == 'f32' else '    '} {'antithetic' if anti else 'plain':10s} {n:.3g} samples in {ms:.1f} ms GPU ({time.time()-t0:.2f} s wall): price {price:.7f} +- {ci:.7f}"
              f"   price-BS = {price-BS:+.2e}  ({(price-BS)/ci*1.96:+.2f} sigma)")
