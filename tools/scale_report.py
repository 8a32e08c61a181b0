#!/usr/bin/env python3
"""What a set of bench.py lines for N = 1, 2, 4, 8 GPUs says about scaling -- for the first lease with more than one GPU (none so
far: DESIGN.md section 6).  Feed it the stdout lines (or files holding them) of

    python bench.py --gpus 1 ...        python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

    python tools/scale_report.py n1.out n2.out n4.out n8.out

WEAK (the headline: every GPU prices `paths` per step): efficiency = value(N) / (N value(1)).
STRONG (one call of C4 / C5 sharded over the N ranks): eff = T1 / (N wall(N)) with T1 from the N = 1 line; next to it what the N-rank
line says about ITSELF -- eff and eff_device_side with T1 measured on its own rank 0, t_shard_ms [min, max] over ranks, collective_ms =
wall - slowest shard, and the 24-byte all-reduce alone (allreduce_us) -- so that a number below 0.90 can be split into device side
(clock state, launch) and collective without another run."""
import json
import sys


def load(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit(f"{path}: no bench line")


def report(lines, out=sys.stdout):
    by_n = {d["n_gpus"]: d for d in lines}
    if 1 not in by_n:
        raise SystemExit("need the N = 1 line")
    one = by_n[1]
    p = lambda *a: print(*a, file=out)     # noqa: E731
    p(f"{'N':>2} {'value paths/s':>14} {'weak eff':>9} {'ms/step':>9}   backend / distinct devices")
    for n in sorted(by_n):
        d = by_n[n]
        devs = len({(r.get('host'), r.get('pci')) for r in d.get('ranks', [])}) or 1
        p(f"{n:2d} {d['value']:14.5g} {d['value'] / (n * one['value']):9.4f} {d['ms_per_step']:9.5f}   {d.get('backend')} / {devs}")
    ss1 = one.get("strong_summary") or {}
    cols = ss1.get("cols", [])
    t1 = {c: v[0][0] for c, v in ss1.items() if c not in ("cols", "src") and isinstance(v, list) and v and isinstance(v[0], list)} if cols else {}
    p()
    p(f"{'config':8s} {'N':>2} {'wall ms':>9} {'eff vs N=1 line':>16} {'eff (own T1)':>13} {'device side':>12} {'t_shard ms min..max':>22} {'collective ms':>14} {'allreduce us':>13}")
    for n in sorted(by_n):
        if n == 1:
            continue
        ss = by_n[n].get("strong_summary") or {}
        ar = (ss.get("allreduce_us") or {}).get("median")
        for c, e in ss.items():
            if not isinstance(e, dict) or "wall_ms_median" not in e:
                continue
            ts = e.get("t_shard_ms") or [float("nan")] * 2
            f = lambda x: f"{x:.4f}" if isinstance(x, (int, float)) else "-"     # noqa: E731
            p(f"{c:8s} {n:2d} {e['wall_ms_median']:9.4f} {f(t1[c] / (n * e['wall_ms_median'])) if c in t1 else '-':>16} {f(e.get('eff')):>13} "
              f"{f(e.get('eff_device_side')):>12} {ts[0]:10.4f} .. {ts[1]:8.4f} {f(e.get('collective_ms')):>14} {f(ar):>13}")
    if cols:
        p()
        p("N = 1, one GPU: T(1) / (8 T(shard 0 of 8)) -- the device side of the 8-GPU point before any collective (hot, cold):")
        for c, v in ss1.items():
            if c not in ("cols", "src") and isinstance(v, list) and len(v) >= 3 and isinstance(v[2], list):
                p(f"  {c:8s} {v[2][0]:.4f}" + (f"  cold {v[3][0]:.4f}" if len(v) > 3 and v[3][0] is not None else ""))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    report([load(a) for a in sys.argv[1:]])
