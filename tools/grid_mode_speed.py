"""What the launch-geometry compatibility mode (mc_*_run_grid_*) costs next to the engine's own streams, at the
reference drivers' shape: 512 blocks x 128 threads (vanillaOpt.cu:13-15, basketOpt.cu:13-15), 1024 x 128 for the CVA
(cvaOpt.cu:12-15), SIMS = k x 131072.  Wall time of the synchronous call, best of 5, first call (state set-up) apart;
fused = round 4's form (the reference's launch itself, streams in registers), staged = round 3's (normals through HBM).
    python tools/grid_mode_speed.py > gpurun_out/grid_mode_speed.log"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlocuda_amd as mc  # noqa: E402

VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=250)


def basket(n, X):
    L, _ = mc.chol(np.full((n, n), 0.5) + 0.5 * np.eye(n), X)
    return dict(s=[100.0] * n, v=[0.3 if i % 2 == 0 else 0.2 for i in range(n)], p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n,
                k=100.0, t=1.0, r=0.048790164)


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


def main():
    eng = mc.Engine(0)
    eng.set_timing(False)
    t0 = time.perf_counter()
    eng.run_grid("vanilla", VAN, 2, 64, 10, "f32")
    print(f"first launch-geometry call of the process (jump matrices on the host + upload + set-up kernel): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    print(f"{'call':34s} {'paths':>11s} {'fused first ms':>14s} {'fused ms':>9s} {'staged ms':>10s} {'engine ms':>10s}   price fused / engine (+- CI)")
    rows = [("vanilla", VAN, 512, 128, 763), ("vanilla", VAN, 512, 128, 8), ("basket", None, 512, 128, 8), ("basket16", None, 512, 128, 8),
            ("cva", CVA, 1024, 128, 1)]
    for X in ("f32", "f64"):
        for name, inp, G, T, k in rows:
            prod = "basket" if name.startswith("basket") else name
            if name == "basket":
                inp_ = basket(4, X)
            elif name == "basket16":
                inp_ = basket(16, X)
            else:
                inp_ = inp
            per_block = k * 131072 // G
            n = G * per_block
            t0 = time.perf_counter()
            eng.run_grid(prod, inp_, G, T, per_block, X)
            first = (time.perf_counter() - t0) * 1e3
            tg, eg = best(lambda: eng.run_grid(prod, inp_, G, T, per_block, X))
            eng.set_grid_form("staged")
            ts, _ = best(lambda: eng.run_grid(prod, inp_, G, T, per_block, X))
            eng.set_grid_form("auto")
            te, ee = best(lambda: getattr(eng, prod)(inp_, n, precision=X))
            print(f"{name + ' ' + X + f' ({G}x{T})':34s} {n:11d} {first:14.3f} {tg:9.3f} {ts:10.3f} {te:10.3f}   "
                  f"{eg.expected:.6f} / {ee.expected:.6f} (+- {eg.confidence:.2g})")
    eng.close()


if __name__ == "__main__":
    main()
