#!/usr/bin/env python3
"""Check + time the matrix-core basket kernel (MC_BASKET_MFMA=1, fp64, 13..16 assets) against the default family.

Child processes because the family switches are read once per process.  Per-path payoffs must agree to a few ulp
(the order of the additions differs), sums to 1e-13; then both price the same 3e7-path call a few times.
"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, time, numpy as np
sys.path.insert(0, %r)
import torch
import montecarlocuda_amd as mc
import bench
n, anti, cv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = bench.basket_inputs(mc, n, "f64")
with mc.Engine(0) as e:
    e.set_antithetic(anti); e.set_control_variate(cv)
    p = e.basket_paths(b, 5003, 12345, 77, "f64")
    r = e.basket(b, 200001, 12345, 5, "f64")
    ts = []
    for _ in range(7):
        t = e.basket(b, 30000000, 12345, 0, "f64")
        ts.append(t.kernel_ms)
    np.save(sys.argv[4], p)
    print(json.dumps({"sum": r.sum, "sum2": r.sum2, "ms": sorted(ts)[len(ts) // 2], "price": t.expected}))
''' % ROOT
for n in (16, 13):
    for anti, cv in ((0, 0), (1, 0), (0, 1), (1, 1)):
        res = {}
        with tempfile.TemporaryDirectory() as d:
            for fam in ("0", "1"):
                out = subprocess.run([sys.executable, "-c", CODE, str(n), str(anti), str(cv), os.path.join(d, fam + ".npy")],
                                     env=dict(os.environ, MC_BASKET_MFMA=fam), capture_output=True, text=True, timeout=600)
                if out.returncode:
                    print(out.stderr[-2000:]); sys.exit(1)
                res[fam] = (json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]), np.load(os.path.join(d, fam + ".npy")))
        a, b = res["0"], res["1"]
        err = np.max(np.abs(a[1] - b[1]) / np.maximum(1.0, np.abs(a[1])))
        print(f"n={n} anti={anti} cv={cv}: per-path max diff {err:.2e}; sum rel {abs(a[0]['sum'] - b[0]['sum']) / abs(a[0]['sum']):.2e}; "
              f"tiled {a[0]['ms']:.3f} ms  mfma {b[0]['ms']:.3f} ms  ({b[0]['ms'] / a[0]['ms']:.3f}x)  price {a[0]['price']:.6f} {b[0]['price']:.6f}", flush=True)
