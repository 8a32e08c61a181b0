#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<workload>/ (tools/collect_pmc.sh) into per-kernel averages.

    python tools/summarize_pmc.py vanilla_f32 [round-tag]

Writes profiles/<tag>_pmc_<workload>.txt (readable) and updates profiles/pmc_traffic.json, which
bench.py reads for roofline.traffic.  HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE counts a wide coalesced read stream at half its bytes, so the
read side is reported both raw and doubled (upper bound) -- for these kernels both are ~KiB.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)   # bench.py (launch_stamp) lives at the repo root; `python tools/summarize_pmc.py` does not put it on the path
w = sys.argv[1] if len(sys.argv) > 1 else "vanilla_f32"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
base = os.path.join(ROOT, "gpurun_out", f"pmc_{w}")
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values per dispatch
grids = defaultdict(set)                       # kernel -> {(workgroups, lanes per workgroup)} seen in the passes
files = []
x2 = defaultdict(lambda: defaultdict(list))    # the same for the pass at twice the paths per launch (sq1x2)
for d in glob.glob(os.path.join(base, "*", "")):      # one directory per pass; keep its newest run only
    runs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if runs and os.path.basename(os.path.dirname(d)) == "sq1x2":
        for row in csv.DictReader(open(runs[-1])):
            if "mc::" in row["Kernel_Name"]:
                x2[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    elif runs:
        files.append(runs[-1])
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "mc::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if row.get("Grid_Size") and row.get("Workgroup_Size"):
            grids[k].add((int(row["Grid_Size"]) // int(row["Workgroup_Size"]), int(row["Workgroup_Size"])))
lines = []
summary = {}
for k in sorted(acc):
    lines.append(f"kernel {k}   launched as (workgroups x lanes): {sorted(grids[k])}")
    avg = {}
    for c in sorted(acc[k]):
        v = acc[k][c]
        avg[c] = sum(v) / len(v)
        lines.append(f"    {c:28s} dispatches={len(v):4d}  avg={avg[c]:.6g}  min={min(v):.6g}  max={max(v):.6g}")
    summary[k] = avg
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        rd, wr = avg["FETCH_SIZE"] * 1024, avg["WRITE_SIZE"] * 1024
        lines.append(f"    -> HBM bytes per launch: read {rd:.0f} (x2 correction: {2*rd:.0f}), write {wr:.0f}")
    if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg:
        lines.append(f"    -> VALU instructions per wave: {avg['SQ_INSTS_VALU']/avg['SQ_WAVES']:.1f}"
                     f", transcendental f32 per wave: {avg.get('SQ_INSTS_VALU_TRANS_F32',0)/avg['SQ_WAVES']:.1f}")
    if "SQ_ACTIVE_INST_VALU" in avg and "SQ_BUSY_CYCLES" in avg:
        lines.append(f"    -> SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {avg['SQ_ACTIVE_INST_VALU']/avg['SQ_WAVE_CYCLES']:.3f}"
                     f"; SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {avg['SQ_ACTIVE_INST_ANY']/avg['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_ACTIVE_INST_LDS" in avg and avg["SQ_ACTIVE_INST_LDS"]:
        lines.append(f"    -> LDS: bank-conflict cycles / active LDS cycles = {avg['SQ_LDS_BANK_CONFLICT']/avg['SQ_ACTIVE_INST_LDS']:.3f}"
                     + (f"; active LDS cycles / SQ busy cycles = {avg['SQ_ACTIVE_INST_LDS']/avg['SQ_BUSY_CYCLES']:.3f}" if "SQ_BUSY_CYCLES" in avg else ""))
text = "\n".join(lines)
print(text)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{w}.txt"), "w").write(
    f"# rocprofv3 --pmc passes of `python3 bench.py --workload {w} --steps 20 --warmup 2 --regions 1` (tools/collect_pmc.sh)\n" + text + "\n")
print(f"launch stamp {__import__('bench').launch_stamp()['stamp'][:16]}")
main = [k for k in summary if "finish" not in k and "masked" not in k]
if main and "FETCH_SIZE" in summary[main[0]]:
    a = summary[main[0]]
    j = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(j)) if os.path.exists(j) else {}
    keep = {k: v for k, v in d.get(w, {}).items() if k.startswith("valu_busy")}   # filled by tools/clock_probe.py runs
    paths = {"vanilla": 10 ** 8, "basket4": 10 ** 8, "basket16": 125 * 10 ** 6, "cva256": 1250000}[w.split("_")[0]]
    from bench import launch_stamp   # the stamp bench.py compares: stale counts are flagged, not used
    stamp = launch_stamp()
    # hot-loop slope: (counts at 2 x paths - counts at 1 x paths) / paths, in wave-instructions per path
    slope = {}
    # the MEDIAN dispatch: besides its K + W steps a bench run makes a launch or two of other sizes (23 dispatches with one at the
    # default path count in the basket and CVA passes), which an average would mix in
    med = lambda v: sorted(v)[len(v) // 2]
    b = {c: med(v) for c, v in x2.get(main[0], {}).items()}
    a1 = {c: med(v) for c, v in acc[main[0]].items()}
    if b and abs(b.get("SQ_WAVES", 0) - a1.get("SQ_WAVES", -1)) < 0.5:   # the same grid in both passes (medians: an average would mix in the odd dispatch)
        for c, key in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_VALU_TRANS_F32", "trans_f32"), ("SQ_INSTS_VALU_TRANS_F64", "trans_f64"),
                       ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_SMEM", "smem"), ("SQ_INSTS_LDS", "lds")):
            if c in a1 and c in b:
                slope[key + "_wave_insts_per_path"] = (b[c] - a1[c]) / paths
        print("hot-loop slope (wave-instructions per path, x 64 = per lane): " + ", ".join(f"{k.split('_wave')[0]} {v * 64:.3f}" for k, v in slope.items()))
    shapes = sorted(grids[main[0]])
    if len(shapes) != 1:
        print(f"WARNING: {main[0]} ran with several launch shapes in the passes: {shapes}")
    d[w] = {"kernel": main[0], "fetch_bytes_raw": a["FETCH_SIZE"] * 1024, "write_bytes": a["WRITE_SIZE"] * 1024,
            "hbm_bytes_per_launch": 2 * a["FETCH_SIZE"] * 1024 + a["WRITE_SIZE"] * 1024,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide reads at half); per simulation-kernel launch",
            "source": f"profiles/{tag}_pmc_{w}.txt",
            # wave-instructions per launch, for bench.py's issue-slot model (roofline.issue_frac)
            "paths_per_launch": paths, "valu_insts_per_launch": a.get("SQ_INSTS_VALU"),
            "trans_f32_per_launch": a.get("SQ_INSTS_VALU_TRANS_F32", 0.0), "trans_f64_per_launch": a.get("SQ_INSTS_VALU_TRANS_F64", 0.0),
            "waves_per_launch": a.get("SQ_WAVES"), "grid_workgroups": shapes[0][0] if shapes else None,
            "group_size": shapes[0][1] if shapes else None, "launch_stamp": stamp["stamp"], "device_code_sha256": stamp["device_code_sha256"],
            "launch_shape_sha256": stamp["launch_shape_sha256"], "hipflags": stamp["hipflags"], "hot_loop_slope": slope, **keep}
    json.dump(d, open(j, "w"), indent=1)
