#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference CPU path.

Run in the dev container only (needs /root/reference to build oracle/_ref):

    make -C oracle && python tests/golden/gen_golden.py

What is executed: the reference's own ``MonteCarloHost.c`` (double_precision/ and
single_precision/), compiled as it lies by ``oracle/Makefile`` (gcc -O2 -ffp-contract=off,
glibc 2.35), with the wall-clock seed of ``MonteCarloHost.c:189,237`` pinned through the
hidden ``time()`` in ``oracle/ref_shim.c``.  What is written: inputs and outputs only
(numbers), as C99 hex floats so they round-trip bit for bit.  No reference source or binary
is written anywhere.

Files written
    ref_bs_call.json   host_bsCall on a grid (MonteCarloHost.c:139-143)
    ref_chol.json      Chol on the driver's N=3 matrix and PD matrices, N=3,4,16 (:90-105)
    ref_uniforms.json  first uniforms of the glibc stream through randMinMax (:111-114)
    ref_mc.json        seed-pinned (Expected, Confidence) of host_vanillaOpt / host_basketOpt /
                       host_cvaEquityOption (:282-311)
``philox_kat.json`` is NOT generated here: it holds the published Random123 known-answer
vectors for Philox4x32-10 and is committed by hand.
"""
import json
import os
import platform
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hx(v):
    return float(v).hex()


def meta():
    return {"generator": "tests/golden/gen_golden.py",
            "source": "reference MonteCarloHost.c compiled unmodified (oracle/Makefile)",
            "cc": "gcc -O2 -ffp-contract=off", "libc": " ".join(platform.libc_ver()),
            "float_format": "C99 hex (float.hex()); f32 values are widened exactly"}


# inputs ---------------------------------------------------------------------------------
VANILLA = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)           # dp/vanillaOpt.cu:22-26
CVA = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6)  # dp/cvaOpt.cu:22-34
REF3 = [[1, -.5, -.5], [-.5, 1, -.5], [-.5, -.5, 1]]                  # dp/basketOpt.cu:46-54


def equicorr(n, rho=0.5):
    return [[1.0 if i == j else rho for j in range(n)] for i in range(n)]


def randpd(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, n + 3))
    c = a @ a.T
    d = np.sqrt(np.diag(c))
    return (c / d[:, None] / d[None, :]).tolist()


def basket_inputs(n, corr):
    """dp/basketOpt.cu:34-68,147-158: S=100, w=1/n, K=100, r=0.048790164, T=1, d=0;
    vols 0.2,0.3,0.2 for the shipped N=3, alternating 0.3/0.2 otherwise."""
    v = [0.2, 0.3, 0.2] if n == 3 else [0.3 if i % 2 == 0 else 0.2 for i in range(n)]
    return dict(s=[100.0] * n, v=v, corr=corr, d=[0.0] * n, w=[1.0 / n] * n, k=100.0, t=1.0,
                r=0.048790164)


def main():
    if not po.ref_available("f64", 3):
        sys.exit("oracle/_ref missing: run `make -C oracle` in the dev container first")

    # ---- bsCall grid -----------------------------------------------------------------
    cases = []
    for X in ("f64", "f32"):
        ref = po.Ref(X, 3)
        for s in (60.0, 90.0, 100.0, 100.5, 110.0, 180.0):
            for v in (0.05, 0.2, 0.6):
                for t in (1e-8, 0.003, 0.25, 1.0, 5.0):
                    for r in (0.0, 0.048790, 0.12):
                        o = dict(s=s, k=100.0, r=r, v=v, t=t)
                        cases.append({"X": X, **o, "out": hx(ref.bs_call(o))})
        for s in (80.0, 120.0):  # t == 0 with s != k: d1 = +-inf -> intrinsic value
            o = dict(s=s, k=100.0, r=0.05, v=0.2, t=0.0)
            cases.append({"X": X, **o, "out": hx(ref.bs_call(o))})
    json.dump({"meta": meta(), "cases": cases}, open(os.path.join(OUT, "ref_bs_call.json"), "w"),
              indent=0)

    # ---- Chol -------------------------------------------------------------------------
    chol = []
    mats = {3: [("ref3_singular", REF3), ("equi0.5", equicorr(3)), ("randpd", randpd(3, 3))],
            4: [("equi0.5", equicorr(4)), ("randpd", randpd(4, 4)),
                ("driver_pattern_indefinite", None)],
            16: [("equi0.5", equicorr(16)), ("randpd", randpd(16, 16))]}
    for n, lst in mats.items():
        for name, m in lst:
            if m is None:  # dp/basketOpt.cu:160-177 pattern: +-0.5 by column parity (not PD)
                m = [[1.0 if i == j else (0.5 if max(i, j) % 2 == 0 else -0.5) for j in range(n)]
                     for i in range(n)]
            for X in ("f64", "f32"):
                ref = po.Ref(X, n)
                L = ref.chol(m)
                chol.append({"X": X, "n": n, "name": name,
                             "c": [[hx(np.dtype(po.NP[X]).type(x)) for x in row] for row in m],
                             "a": [[hx(x) for x in row] for row in L]})
    json.dump({"meta": meta(), "cases": chol}, open(os.path.join(OUT, "ref_chol.json"), "w"),
              indent=0)

    # ---- uniforms ---------------------------------------------------------------------
    uni = []
    for X in ("f64", "f32"):
        ref = po.Ref(X, 3)
        for seed in (1, 777, 12345):
            uni.append({"X": X, "seed": seed, "u": [hx(x) for x in ref.uniforms(seed, 32)]})
    json.dump({"meta": meta(), "cases": uni}, open(os.path.join(OUT, "ref_uniforms.json"), "w"),
              indent=0)

    # ---- Monte Carlo ------------------------------------------------------------------
    mc = []
    for X in ("f64", "f32"):
        ref = po.Ref(X, 3)
        # BASELINE.json configs[0] (C1): 10^6 fp64 paths, seeds 12345/777/1
        for paths, seeds in ((2, (1,)), (1000, (1, 777)), (100000, (12345,)),
                             (1000000, (12345, 777, 1))):
            for seed in seeds:
                e, c = ref.vanilla(VANILLA, paths, seed)
                mc.append({"kind": "vanilla", "X": X, "paths": paths, "seed": seed,
                           "opt": VANILLA, "expected": hx(e), "confidence": hx(c)})
        for n_grid, paths, seed in ((25, 100, 1), (250, 2000, 12345), (256, 500, 777),
                                    (250, 10000, 12345), (500, 200, 5)):
            c = dict(CVA, n_grid=n_grid)
            e, ci = ref.cva(c, paths, seed)
            mc.append({"kind": "cva", "X": X, "paths": paths, "seed": seed, "cva": c,
                       "expected": hx(e), "confidence": hx(ci)})
        for n, corr_name, corr in ((3, "ref3_singular", REF3), (3, "equi0.5", equicorr(3)),
                                   (4, "equi0.5", equicorr(4)), (16, "equi0.5", equicorr(16))):
            refn = po.Ref(X, n)
            b = basket_inputs(n, corr)
            L = refn.chol(corr)                      # driver does this, dp/basketOpt.cu:96-99
            b_call = dict(b, p=L.tolist())
            for paths, seed in ((1000, 1), (100000, 12345)):
                e, ci = refn.basket(b_call, paths, seed)
                mc.append({"kind": "basket", "X": X, "n": n, "corr_name": corr_name,
                           "paths": paths, "seed": seed,
                           "basket": {k: b[k] for k in ("s", "v", "d", "w", "k", "t", "r")},
                           "corr": corr, "factor": [[hx(x) for x in row] for row in L],
                           "expected": hx(e), "confidence": hx(ci)})
    json.dump({"meta": meta(), "cases": mc}, open(os.path.join(OUT, "ref_mc.json"), "w"), indent=0)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
