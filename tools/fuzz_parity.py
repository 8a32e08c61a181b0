#!/usr/bin/env python3
"""One-off randomized parity sweep on the GPU box: random market data, sizes, ranges and estimator switches for
all three products, HIP engine vs the oracle twin per path (the tolerances of tests/test_gpu_parity.py, scaled by
the spot level).  Prints the worst normalised error per product/precision; exits non-zero on any violation.
    python tools/fuzz_parity.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import montecarlocuda_amd as mc
from oracle import pyoracle as po     # the checker
po.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
EXTREME = len(sys.argv) > 3 and sys.argv[3] == "extreme"   # tiny / large vols and maturities, far strikes, odd rates
TOL = {"f32": dict(pay=2e-6, cva=2e-5), "f64": dict(pay=1e-14, cva=1e-13)}
worst, bad = {}, 0
eng = {(a, c): mc.Engine(0) for a in (False, True) for c in (False, True)}
for (a, c), e in eng.items():
    e.set_antithetic(a)
    e.set_control_variate(c)


def note(key, err, tol, what):
    global bad
    worst[key] = max(worst.get(key, 0.0), err / tol)
    if not err <= tol:
        bad += 1
        print("VIOLATION", key, what, f"err {err:.3e} > tol {tol:.3e}")


refused = 0
for it in range(cases):
  try:
      X = "f32" if rng.random() < 0.5 else "f64"
      anti = bool(rng.random() < 0.4)
      seed = int(rng.integers(0, 2 ** 63))
      first = int(rng.integers(0, 2 ** 40)) if rng.random() < 0.7 else (1 << 32) * int(rng.integers(1, 100)) - int(rng.integers(1, 300))
      n_paths = int(rng.integers(1, 700))
      prod = rng.choice(["vanilla", "basket", "basket", "cva", "cva"])
      spot = float(np.exp(rng.uniform(np.log(5), np.log(500))))
      r, t = float(rng.uniform(0.0, 0.08)), float(rng.choice([0.25, 0.5, 1.0, 2.0, 3.0]))
      vol_lo, vol_hi, k_lo, k_hi = 0.05, 0.6, 0.6, 1.5
      if EXTREME:
          r, t = float(rng.uniform(-0.05, 0.4)), float(np.exp(rng.uniform(np.log(1e-4), np.log(30.0))))
          vol_lo, vol_hi, k_lo, k_hi = 0.002, 1.5, 0.05, 20.0
      if prod == "vanilla":
          o = dict(s=spot, k=spot * float(rng.uniform(k_lo, k_hi)), r=r, v=float(rng.uniform(vol_lo, vol_hi)), t=t)
          got = eng[(anti, False)].vanilla_paths(o, n_paths, seed, first, X).astype(np.float64)
          want, _ = po.dev_vanilla(X, o, seed, first, n_paths, antithetic=anti)
          level = spot * float(np.exp(max(0.0, r) * t + 4 * o["v"] * np.sqrt(t)))   # payoffs (and their rounding) scale with the reachable spot
          note((prod, X), float(np.abs(got - want.astype(np.float64)).max()), TOL[X]["pay"] * level * 3, (o, anti, first, n_paths))
      elif prod == "basket":
          n = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 23, 28, 32, 33, 40, 64]))
          cv = bool(rng.random() < 0.3)
          rho = float(rng.uniform(-0.9 / max(n - 1, 1), 0.9)) if n > 1 else 0.0
          corr = np.full((n, n), rho) + (1 - rho) * np.eye(n)
          L, nbad = mc.chol(corr, X)
          if nbad:
              continue
          w = rng.uniform(0.5, 1.5, n)
          w = (w / w.sum() * float(rng.uniform(0.8, 1.2))).tolist()
          s = (spot * rng.uniform(0.7, 1.3, n)).tolist()
          b = dict(s=s, v=rng.uniform(vol_lo, min(vol_hi, 0.9), n).tolist(), p=L.tolist(), d=(rng.uniform(-0.02, 0.02, n) if rng.random() < 0.3 else np.zeros(n)).tolist(),
                   w=w, k=float(np.dot(w, s) * rng.uniform(max(k_lo, 0.3), min(k_hi, 3.0))), t=t, r=r)
          got = eng[(anti, cv)].basket_paths(b, n_paths, seed, first, X).astype(np.float64)
          want, _ = po.dev_basket(X, b, seed, first, n_paths, antithetic=anti, control=cv)
          level = float(np.dot(w, s)) * float(np.exp(max(0.0, r) * t + 4 * max(b["v"]) * np.sqrt(t)))   # reachable basket level
          note((prod, X), float(np.abs(got - want.astype(np.float64)).max()), TOL[X]["pay"] * level * 4 * (1 + n / 16), (n, anti, cv, first, n_paths))
      else:
          c = dict(s=spot, k=spot * float(rng.uniform(max(k_lo, 0.2), min(k_hi, 5.0))), r=r, v=float(rng.uniform(max(vol_lo, 0.02), min(vol_hi, 1.0))), t=t,
                   defint=float(rng.uniform(0.005, 0.1)), lgd=float(rng.uniform(0.2, 0.9)), n_grid=int(rng.integers(1, 1500 if EXTREME else 300)))
          lanes = int(rng.choice([0, 0, 1, 2, 4, 8, 16, 32, 64]))    # 0 = the call-size rule (small calls: date-parallel), else forced
          eng[(anti, False)].set_cva_date_lanes(lanes)
          got = eng[(anti, False)].cva_paths(c, min(n_paths, 200), seed, first, X).astype(np.float64)
          want, _ = po.dev_cva(X, c, seed, first, min(n_paths, 200), antithetic=anti)
          level = spot * float(np.exp(max(0.0, r) * t + 4 * c["v"] * np.sqrt(t)))
          note((prod, X), float(np.abs(got - want.astype(np.float64)).max()), TOL[X]["cva"] * level / 100 * 3 * (1 + c["n_grid"] / 256), (c, anti, first, lanes))
  except mc.McError as ex:
      # inputs whose exponent leaves the range of the precision are REFUSED by the engine (tests/test_gpu_parity.py
      # test_invalid_arguments_are_errors_not_crashes); anything else is a failure of the sweep
      if "range" not in str(ex):
          raise
      refused += 1
if not all(np.isfinite(v) for v in worst.values()):
    bad += 1
for k in sorted(worst):
    print(f"{k[0]:8s} {k[1]}: worst error / bound = {worst[k]:.3f}")
print(f"{cases} cases ({refused} refused by the range guards), {bad} violations")
sys.exit(1 if bad else 0)
