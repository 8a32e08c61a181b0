#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ...` run of tools/clock_probe.py:
VALU busy = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), per kernel, last (longest) dispatch.
    python tools/clock_probe_summary.py gpurun_out/clock [log-name-for-the-source-field]
With a second argument the figures are also written into profiles/pmc_traffic.json (valu_busy_long_launch per workload,
matched by kernel name), which bench.py reports next to the issue-slot model."""
import json
import csv
import glob
import os
import sys
from collections import defaultdict

rows = defaultdict(dict)
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mc::" in k and "finish" not in k:
            rows[(k, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (k, d), c in rows.items():
    if "GRBM_GUI_ACTIVE" in c and (k not in best or c["GRBM_GUI_ACTIVE"] > best[k]["GRBM_GUI_ACTIVE"]):
        best[k] = c
for k, c in best.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:52s} VALU busy {100 * c['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:5.1f} %   cycles per VALU instruction "
          f"{cyc * 1024 / c['SQ_INSTS_VALU']:.2f}   GRBM_GUI_ACTIVE/8 = {cyc:.4g} cycles")

if len(sys.argv) > 2:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    j = os.path.join(root, "profiles", "pmc_traffic.json")
    d = json.load(open(j))
    for w, v in d.items():
        c = best.get(v.get("kernel"))
        if c:
            v["valu_busy_long_launch"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 3)
            v["valu_busy_source"] = f"profiles/{sys.argv[2]} (SQ_ACTIVE_INST_VALU*4/1024 / (GRBM_GUI_ACTIVE/8) on a ~10 ms launch)"
    json.dump(d, open(j, "w"), indent=1)
