#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop from the ISA listing (make -C montecarlocuda_amd/csrc asm).

    python tools/loop_mix.py <mangled-name-substring> [more substrings ...]

The hot loop = the innermost backward branch whose body holds the most Philox multiplies (v_mad_u64_u32).  Classes: VALU full-rate, fp32 transcendental
(v_exp/log/sin/cos/sqrt/rcp/rsq_f32: 8 cycles), fp64 rcp/sqrt (16 cycles), LDS, SALU/SMEM, other.
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "montecarlocuda_amd", "csrc", "mc_api.gfx950.s")).read()
TRANS32 = re.compile(r"^v_(exp|log|sin|cos|sqrt|rcp|rsq)_f32")
TRANS64 = re.compile(r"^v_(rcp|sqrt|rsq)_f64")

for pat in sys.argv[1:]:
    names = [m for m in re.findall(r"^(_ZN2mc\w+):", txt, re.M) if pat in m]
    for name in names:
        body = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M).group(1).split("\n")
        labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        best = None
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                span = body[labels[m.group(1)]:i + 1]
                n = (sum("v_mad_u64_u32" in x for x in span), -len(span))
                if n[0] and (best is None or n > best[0]):
                    best = (n, labels[m.group(1)], i)
        if not best:
            print(name, ": no loop")
            continue
        ops = [l.strip().split()[0] for l in body[best[1]:best[2] + 1] if l.strip() and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        t32 = sum(v for k, v in c.items() if TRANS32.match(k))
        t64 = sum(v for k, v in c.items() if TRANS64.match(k))
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        sal = sum(v for k, v in c.items() if k.startswith("s_"))
        cyc = (valu - t32 - t64) * 4.1 + t32 * 8.1 + t64 * 16.1
        print(f"{name}\n  loop: {len(ops)} instructions, VALU {valu} (fp32 transcendental {t32}, fp64 rcp/sqrt {t64}), LDS {lds}, scalar {sal}; "
              f"issue model {cyc:.0f} cycles per wave-trip")
        print("  " + ", ".join(f"{v} {k}" for k, v in c.most_common(40)))
