#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage output (make -C montecarlocuda_amd/csrc asm)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "montecarlocuda_amd/csrc/mc_api.resource-usage.txt"
txt = open(path).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"^void ", "", re.sub(r"\(.*", "", dn))[:78]
    vals = []
    for k, pat in KEYS:
        m = re.search(pat + r": (\d+)", b)
        vals.append(f"{k}={m.group(1) if m else '?'}")
    print(f"{dn:80s} " + " ".join(vals))
