#!/usr/bin/env python3
"""Check DESIGN.md's measured figures against the profiles/ files it names (VERDICT r03 "next" #7: an explanation must
not outlive its evidence).

A *block* of DESIGN.md is a paragraph (consecutive non-blank lines) or one table row; a table row also inherits the
files cited by the paragraph directly above its table.  In every block that cites at least one `profiles/<file>`
(or `BENCH_rNN.json`), each *measured-looking* figure --

    2.07e12, 3e-6               scientific notation
    55.4 µs, 0.110 ms, 7.7 s    a number followed by us / µs / ms / s / ns
    97.3 %                      a number followed by %
    2.38 GHz, 5.8 GB/s          a number followed by GHz / MHz / GB/s / TB/s / TFLOP/s

-- must occur in one of the cited files: some number of the file, scaled by a power of ten a unit change can
introduce (1e-9 ... 1e9 in steps of 1e3, and x100 for per cent), rounds to the figure at the precision DESIGN.md
prints it with.  Exempt: figures inside `code spans`, figures directly preceded by ≈ or ~ or < or > or ≤ or ≥ or ± (estimates
and bounds are prose, not read-outs), and figures followed by "(derived)".  Blocks without a citation are not checked:
tolerances, peaks and design constants live there.

Exit status 1 and one line per miss when a figure is in no cited file, or when a cited file does not exist.
    python tools/check_design_numbers.py [DESIGN.md] [--list]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CITE = re.compile(r"(profiles/[A-Za-z0-9_.\-]+|BENCH_r\d+\.json)")
NUM = r"\d+(?:\.\d+)?"
FIGURE = re.compile(
    r"(?P<pre>[≈~<>≤≥±]\s*)?(?<![\w.\-/])(?P<num>" + NUM + r"(?:e[+-]?\d+)?)"
    r"(?P<unit>\s*(?:µs|us|ms|ns|s|%|GHz|MHz|GB/s|TB/s|TFLOP/s)(?![\w/]))?(?P<post>\s*\(derived\))?")
FILE_NUMBER = re.compile(r"(?<![\w.])[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")
SCALES = [10.0 ** k for k in range(-9, 10, 3)] + [100.0, 0.01]


def blocks(text):
    """(first line number, text, inherited citations) for every paragraph / table row."""
    out, para, start, caption_cites, last_para_cites = [], [], 0, [], []
    lines = text.split("\n")
    in_code = False
    for i, line in enumerate(lines + [""], 1):
        if line.startswith("```"):
            in_code = not in_code
            continue
        if in_code:
            continue
        is_row = line.lstrip().startswith("|")
        if is_row:
            if para:
                last_para_cites = CITE.findall(" ".join(para))
                out.append((start, " ".join(para), []))
                para = []
            if not caption_cites:
                caption_cites = last_para_cites
            if not re.fullmatch(r"\s*\|[\s:\-|]+\|\s*", line):
                out.append((i, line, list(caption_cites)))
            continue
        caption_cites = []
        if line.strip() == "":
            if para:
                last_para_cites = CITE.findall(" ".join(para))
                out.append((start, " ".join(para), []))
                para = []
            continue
        if line.startswith("#"):
            if para:
                out.append((start, " ".join(para), []))
                para = []
            last_para_cites = []
            continue
        if not para:
            start = i
        para.append(line)
    return out


def figures(block):
    text = re.sub(r"`[^`]*`", lambda m: " " * len(m.group(0)), block)      # code spans carry names, not read-outs
    text = CITE.sub(lambda m: " " * len(m.group(0)), text)
    for m in FIGURE.finditer(text):
        num, unit = m.group("num"), (m.group("unit") or "").strip()
        sci = "e" in num
        if not sci and not unit:
            continue
        if m.group("pre") or m.group("post"):
            continue
        yield num, unit


def decimals_and_value(num):
    """The figure as (value, half a unit of its last printed digit)."""
    if "e" in num:
        mant, exp = num.split("e")
        exp = int(exp)
    else:
        mant, exp = num, 0
    dec = len(mant.split(".")[1]) if "." in mant else 0
    return float(num), 0.5 * 10.0 ** (exp - dec)


_cache = {}


def file_numbers(path):
    if path not in _cache:
        with open(path, "r", errors="replace") as f:
            vals = set()
            for tok in FILE_NUMBER.findall(f.read()):
                try:
                    v = abs(float(tok))
                except ValueError:
                    continue
                if v != 0.0 and v == v and v != float("inf"):
                    vals.add(v)
        _cache[path] = sorted(vals)
    return _cache[path]


def found(num, unit, paths):
    import bisect
    value, half = decimals_and_value(num)
    if value == 0.0:
        return True
    scales = [1.0, 1e3, 1e-3, 1e6, 1e-6, 1e9, 1e-9] if "e" in num else SCALES
    for p in paths:
        vals = file_numbers(p)
        for s in scales:
            lo, hi = (value - half) / s, (value + half) / s
            i = bisect.bisect_left(vals, lo * (1 - 1e-12))
            if i < len(vals) and vals[i] <= hi * (1 + 1e-12):
                return True
    return False


def main(argv):
    listing = "--list" in argv
    args = [a for a in argv if not a.startswith("--")]
    design = args[0] if args else os.path.join(ROOT, "DESIGN.md")
    text = open(design, encoding="utf-8").read()
    misses = checked = cited_blocks = 0
    for line_no, block, inherited in blocks(text):
        cites = CITE.findall(block) + inherited
        if not cites:
            continue
        cited_blocks += 1
        paths = []
        for c in dict.fromkeys(cites):
            p = os.path.join(ROOT, c)
            if not os.path.exists(p):
                print(f"{os.path.basename(design)}:{line_no}: cited file does not exist: {c}")
                misses += 1
            else:
                paths.append(p)
        for num, unit in figures(block):
            checked += 1
            ok = found(num, unit, paths)
            if listing:
                print(f"{line_no:5d}  {num + ' ' + unit:>14s}  {'ok' if ok else 'MISSING'}  {', '.join(os.path.basename(p) for p in paths)}")
            if not ok:
                misses += 1
                if not listing:
                    print(f"{os.path.basename(design)}:{line_no}: {num} {unit} is in none of: {', '.join(dict.fromkeys(cites))}")
    print(f"{os.path.basename(design)}: {checked} figures in {cited_blocks} cited blocks checked, {misses} not found")
    return 1 if misses else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
