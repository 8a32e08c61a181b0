#!/usr/bin/env python3
"""Golden vectors for the covariance-input helper (mc_factor_from_cov_*) -> tests/golden/ref_cov.json.

Run in the dev container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle && python tests/golden/gen_golden_cov.py

A caller of the reference who holds a covariance matrix must normalise it to volatilities + correlation by hand and
then call the reference's Chol (double_precision/basketOpt.cu:96-99 -> MonteCarloHost.c:90-105).  This script does
exactly that: the normalisation in numpy scalars of the target precision (IEEE operations, one rounding each:
v_a = sqrt(cov_aa); corr_ab = cov_ab / (v_a * v_b), unit diagonal, lower triangle mirrored), the factorisation by the
UNMODIFIED reference Chol compiled by oracle/Makefile.  Written: inputs and outputs only, as C99 hex floats.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from gen_golden import hx, meta  # noqa: E402


def cov_cases(n):
    rng = np.random.default_rng(1000 + n)
    out = []
    vols = np.array([0.3 if i % 2 == 0 else 0.2 for i in range(n)])
    corr = np.full((n, n), 0.5) + 0.5 * np.eye(n)
    out.append(("equi0.5_alternating_vols", corr * vols[:, None] * vols[None, :]))          # BASELINE C3/C4 inputs as a covariance
    a = rng.standard_normal((n, n + 4)) * 0.15
    out.append(("random_pd", a @ a.T))
    out.append(("daily_scale", (a @ a.T) / 252.0))
    if n == 3:   # the reference driver's own singular correlation (all -0.5) with its vols 0.2, 0.3, 0.2
        v = np.array([0.2, 0.3, 0.2])
        out.append(("ref3_singular", (np.full((3, 3), -0.5) + 1.5 * np.eye(3)) * v[:, None] * v[None, :]))
    a = rng.standard_normal((n, max(1, n - 2)))
    out.append(("rank_deficient", a @ a.T + (1e-3 * np.eye(n) if n < 3 else 0)))           # not positive definite for n >= 3
    return out


def main():
    if not po.ref_available("f64", 3):
        sys.exit("oracle/_ref missing: run `make -C oracle` in the dev container first")
    cases = []
    for n in (3, 4, 16):
        for name, cov in cov_cases(n):
            for X in ("f64", "f32"):
                T = np.dtype(po.NP[X]).type
                c = np.array(cov, dtype=po.NP[X])
                if not all(c[i, i] > 0 for i in range(n)):
                    continue
                v = np.array([np.sqrt(c[i, i]) for i in range(n)], dtype=po.NP[X])
                corr = np.zeros((n, n), dtype=po.NP[X])
                for i in range(n):
                    corr[i, i] = T(1)
                    for j in range(i):
                        corr[i, j] = corr[j, i] = c[i, j] / (v[i] * v[j])
                L = po.Ref(X, n).chol(corr.tolist())
                cases.append({"X": X, "n": n, "name": name, "cov": [[hx(x) for x in row] for row in c],
                              "v": [hx(x) for x in v], "corr": [[hx(x) for x in row] for row in corr],
                              "a": [[hx(x) for x in row] for row in L],
                              "bad_pivots": int(sum(1 for i in range(n) if not L[i][i] > 0))})
    m = meta()
    m["generator"] = "tests/golden/gen_golden_cov.py"
    json.dump({"meta": m, "cases": cases}, open(os.path.join(OUT, "ref_cov.json"), "w"), indent=0)
    print(len(cases), "covariance cases written")


if __name__ == "__main__":
    main()
