#!/usr/bin/env python3
"""State-leak check of one long-lived context: a random sequence of every kind of call the C ABI offers -- pricing calls in
both precisions under random estimator / generator / normals settings, per-path dumps, from-normals hooks (with their test
flags), launch-geometry calls with changing geometries, Greeks, asynchronous launches read back through armed pinned slots
-- on ONE engine, each result compared bit for bit with the same call made on a FRESH engine configured the same way.
Catches anything a call leaves behind in the context (cached tables, the external-normals switch, generator state arrays,
armed slots, ticket words).
    python tools/state_mix_check.py [steps] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401  (first: see tests/conftest.py)
except ImportError:
    torch = None
import montecarlocuda_amd as mc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA0 = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6)
NP = {"f32": np.float32, "f64": np.float64}


def basket(n, X, rho):
    L, bad = mc.chol(np.full((n, n), rho) + (1 - rho) * np.eye(n), X)
    assert bad == 0
    return dict(s=[100.0] * n, v=[0.3 if i % 2 == 0 else 0.2 for i in range(n)], p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n, k=100.0,
                t=1.0, r=0.048790164)


def configure(e, cfg):
    e.set_antithetic(cfg["anti"])
    e.set_control_variate(cfg["cv"])
    e.set_generator(cfg["gen"], cfg["base"])
    e.set_normals(cfg["normals"])
    e.set_timing(cfg["timing"])


def inputs(prod, X):
    if prod == "vanilla":
        return dict(VAN, k=float(rng.choice([90.0, 100.0, 110.0])), v=float(rng.choice([0.1, 0.2, 0.4])))
    if prod == "basket":
        return basket(int(rng.choice([2, 3, 4, 8, 12, 16, 20, 33])), X, float(rng.choice([0.0, 0.3, 0.5])))
    return dict(CVA0, n_grid=int(rng.choice([1, 3, 12, 50, 64])), v=float(rng.choice([0.2, 0.3])))


def do(e, op):
    kind = op["kind"]
    prod, X, inp = op["prod"], op["X"], op["inp"]
    if kind == "run":
        r = getattr(e, prod)(inp, op["n"], op["seed"], op["first"], X)
        return (r.sum, r.sum2, r.n)
    if kind == "paths":
        return getattr(e, prod + "_paths")(inp, op["n"], op["seed"], op["first"], X).tobytes()
    if kind == "grid":
        r = e.run_grid(prod, inp, op["G"], op["T"], op["per"], X)
        return (r.sum, r.sum2, r.n)
    if kind == "from_normals":
        z = op["z"]
        if prod == "vanilla":
            r, v = e.vanilla_from_normals(inp, z, X)
        elif prod == "basket":
            r, v = e.basket_from_normals(inp, z, X, no_vol=op["flag"])
        else:
            r, v = e.cva_from_normals(inp, z, X, host_order=op["flag"])
        return (r.sum, r.sum2, r.n, v.tobytes())
    if kind == "greeks":
        g = e.vanilla_greeks(inp, op["n"], op["seed"], op["first"], X)
        return tuple((x.sum, x.sum2) for x in g)
    if kind == "armed":
        out = torch.zeros(3, dtype=torch.float64, device="cuda")
        struct, keep = e.prepared(prod, X, inp)
        slot = e.arm_direct()
        e.launch(prod, X, struct, op["seed"], op["first"], op["n"], out.data_ptr(), e.stream)
        res = e.wait_slot(slot)
        torch.cuda.synchronize()
        return tuple(res) + tuple(out.tolist())
    raise ValueError(kind)


main = mc.Engine(0)
bad = 0
counts = {}
for it in range(steps):
    X = str(rng.choice(["f32", "f64"]))
    prod = str(rng.choice(["vanilla", "basket", "cva"]))
    kinds = ["run", "run", "paths", "grid", "from_normals", "greeks"] + (["armed"] if torch is not None else [])
    kind = str(rng.choice(kinds))
    cfg = dict(anti=False, cv=False, gen="philox", base=0, normals="native", timing=bool(rng.random() < 0.5))
    op = dict(kind=kind, prod=prod, X=X, seed=int(rng.integers(0, 2 ** 63)), first=int(rng.integers(0, 2 ** 40)),
              n=int(rng.integers(1, 20000)))
    if kind in ("run", "paths"):
        cfg["anti"] = bool(rng.random() < 0.3)
        cfg["cv"] = bool(prod == "basket" and rng.random() < 0.3)
        if rng.random() < 0.25 and not cfg["cv"]:
            cfg["gen"], cfg["base"] = "xorwow", int(rng.integers(0, 1000))
            op["first"] = 4 * int(rng.integers(0, 1000))     # XORWOW: unit-aligned ranges, one launch
            if kind == "paths":
                op["n"] = 4 * int(rng.integers(1, 500))
        elif X == "f64" and rng.random() < 0.3:
            cfg["normals"] = "f32"
    if kind == "greeks":
        op["prod"] = prod = "vanilla"
    if kind == "armed":
        cfg["timing"] = False
    op["inp"] = inputs(prod, X)
    if kind == "paths":
        op["n"] = min(op["n"], 3000)
    if kind == "grid":
        op.update(G=int(rng.choice([1, 3, 8, 40])), T=int(rng.choice([1, 32, 100, 256])), per=int(rng.integers(1, 300)))
    if kind == "from_normals":
        n = int(rng.integers(1, 500))
        per = 1 if prod == "vanilla" else (len(op["inp"]["s"]) if prod == "basket" else op["inp"]["n_grid"])
        op["z"] = rng.standard_normal((n, per)).astype(NP[X])
        op["flag"] = bool(rng.random() < 0.4)
    try:
        configure(main, cfg)
        got = do(main, op)
        fresh = mc.Engine(0)
        configure(fresh, cfg)
        want = do(fresh, op)
        fresh.close()
    except mc.McError as ex:          # a refused combination must be refused the same way by both
        fresh = mc.Engine(0)
        configure(fresh, cfg)
        try:
            do(fresh, op)
            print("VIOLATION: only the long-lived context refused", kind, prod, X, cfg, ex)
            bad += 1
        except mc.McError:
            pass
        fresh.close()
        counts[kind + " (refused)"] = counts.get(kind + " (refused)", 0) + 1
        continue
    counts[kind] = counts.get(kind, 0) + 1
    if got != want:
        bad += 1
        print("VIOLATION", it, kind, prod, X, cfg, {k: v for k, v in op.items() if k not in ("z", "inp")})
print(f"{steps} steps, {bad} violations; calls by kind: {counts}")
main.close()
sys.exit(1 if bad else 0)
