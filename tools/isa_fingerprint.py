#!/usr/bin/env python3
"""Per-kernel fingerprint of the ISA listing (make -C montecarlocuda_amd/csrc asm): total instruction count, opcode histogram
hash, VGPR/SGPR counts.  `isa_fingerprint.py save F` writes it, `isa_fingerprint.py diff F` lists the kernels whose code
changed since -- how a refactor of shared device functions is shown to leave the hot kernels' code alone.
"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "montecarlocuda_amd", "csrc", "mc_api.gfx950.s")


def fingerprint():
    txt = open(S).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ops = [l.strip().split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
        c = collections.Counter(ops)
        h = hashlib.sha256(json.dumps(sorted(c.items())).encode()).hexdigest()[:16]
        seq = hashlib.sha256("\n".join(ops).encode()).hexdigest()[:16]
        meta = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", txt, re.S)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta.group(1)).group(1) if meta else "?"
        sg = re.search(r"\.amdhsa_next_free_sgpr (\d+)", meta.group(1)).group(1) if meta else "?"
        out[name] = {"insts": len(ops), "valu": sum(v for k, v in c.items() if k.startswith("v_")), "hist": h, "seq": seq, "vgpr": vg, "sgpr": sg}
    return out


def demangle(n):
    return re.sub(r"\(.*", "", subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip())[:110]


if __name__ == "__main__":
    cur = fingerprint()
    if sys.argv[1] == "save":
        json.dump(cur, open(sys.argv[2], "w"), indent=0, sort_keys=True)
        print(f"{len(cur)} kernels")
    else:
        old = json.load(open(sys.argv[2]))
        same = changed = 0
        for n in sorted(set(old) | set(cur)):
            a, b = old.get(n), cur.get(n)
            if a == b:
                same += 1
                continue
            changed += 1
            if a is None or b is None:
                print(("NEW     " if a is None else "GONE    ") + demangle(n))
            else:
                kind = "reordered" if a["hist"] == b["hist"] and a["insts"] == b["insts"] else "CHANGED  "
                print(f"{kind} {demangle(n)}: insts {a['insts']} -> {b['insts']}, VALU {a['valu']} -> {b['valu']}, vgpr {a['vgpr']} -> {b['vgpr']}, sgpr {a['sgpr']} -> {b['sgpr']}")
        print(f"{same} kernels identical, {changed} differ")
