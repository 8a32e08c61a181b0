#!/usr/bin/env python3
"""After an evidence run (tools/round_evidence.sh + the extra steps named there) has been merged into gpurun_out/: copy the round's files
into profiles/ and tests/golden/, and rewrite the tables and the sentences of DESIGN.md / README.md / profiles/README.md that quote them
(every figure stays checked by tools/check_design_numbers.py afterwards).  A maintenance script for the development container:
    python tools/refresh_doc_tables.py r06"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, dst)


for f in ("bench_default.log", "bench_detail_default.json", "bench_driver_style_1.log", "bench_detail_driver_style_1.json", "bench_driver_style_2.log",
          "shard_clock_AD.log", "basket_greeks_speed.log", "smoke.log", "bench_all_workloads.log", "rocprofv3_kernel_stats_bench_default.csv",
          "rocprofv3_kernel_stats_bench_streams1.csv", "cva_call_latency.log"):
    cp(os.path.join(G, f"{TAG}_{f}"), os.path.join(P, f"{TAG}_{f}"))
cp(os.path.join(G, "bench_secondary.json"), os.path.join(P, f"{TAG}_bench_secondary_estimators.json"))
cp(os.path.join(G, "bench_all.json"), os.path.join(P, f"{TAG}_bench_all_workloads.json"))
cp(os.path.join(G, "bench_all.json"), os.path.join(ROOT, "tests", "golden", f"bench_all_{TAG}.json"))
cp(os.path.join(G, f"{TAG}_bench_detail_driver_style_1.json"), os.path.join(ROOT, "tests", "golden", f"bench_detail_{TAG}.json"))


def line_of(f):
    return json.loads([x for x in open(os.path.join(P, f)).read().splitlines() if x.startswith("{")][0])


dflt, s1, s2 = line_of(f"{TAG}_bench_default.log"), line_of(f"{TAG}_bench_driver_style_1.log"), line_of(f"{TAG}_bench_driver_style_2.log")
NUM = r"[0-9][0-9.e+\-]*"
D = os.path.join(ROOT, "DESIGN.md")
d = open(D, encoding="utf-8").read()


def sub(pattern, repl, text, count=1):
    new, n = re.subn(pattern, repl, text, count=count, flags=re.S)
    assert n == count, (pattern, n)
    return new


# ---- all-workloads table
names = {"vanilla_f32": "vanilla f32, 1e8 paths (headline)", "vanilla_f64": "vanilla f64", "vanilla_f64_n32": "vanilla f64 on fp32 normals",
         "basket4_f32": "basket n=4 f32 (C3)", "basket16_f32": "basket n=16 f32, 1.25e8 paths", "basket16_f64": "basket n=16 f64 (C4's kernel), 1.25e8",
         "basket16_f64_n32": "basket n=16 f64 on fp32 normals", "cva256_f64": "CVA 256 dates f64 (C5's kernel), 1.25e6",
         "cva256_f64_n32": "CVA 256 dates f64 on fp32 normals", "cva256_f32": "CVA 256 dates f32"}
log = open(os.path.join(P, f"{TAG}_bench_all_workloads.log")).read()
tab, mx = [], 0.0
for w, nm in names.items():
    m = re.search(r"^%s\s.*?value (\S+) paths/s\s+alone \S+ paths/s \((\S+) us\)\s+flop frac (\S+)\s+issue ceiling (\S+) us frac (\S+) \(step period (\S+),"
                  r".*?sclk (\d+) MHz -> at measured clock (\S+)" % re.escape(w), log, re.M)
    v, k, f, c, i, sp, clk, ic = m.groups()
    mx = max(mx, float(ic), float(i))
    tab.append(f"| {nm} | {v} | {k} µs | {f} | {c} µs | {i} / {sp} / {ic} at {clk} MHz |")
i0, i1 = d.index("| vanilla f32, 1e8 paths (headline) |"), d.index("\n\nThe secondary estimators on the same kernels")
d = d[:i0] + "\n".join(tab) + d[i1:]
d = sub(r"No row reaches 1\.00 at either clock \(largest: [0-9.]+\)\.", f"No row reaches 1.00 at either clock (largest: {mx:.3f}).", d)
# ---- configs table
c = dflt["configs"]
rows = [("C2 vanilla 1e8 fp32", "C2", "µs", 1, "vs Black-Scholes", "paths/s"), ("C3 basket n=4 1e8 fp32", "C3", "µs", 1, "vs fp64, 1e9 paths", "(sp object)"),
        ("C4 basket n=16 1e9 fp64", "C4", "ms", 1e-3, "vs the 1e10-path run", "(sp object)"), ("C5 CVA 256 × 1e7 fp64", "C5", "ms", 1e-3, "vs the closed form", "paths/s")]
new = []
for label, k, unit, scale, vs, cpu in rows:
    e = c[k]
    new.append(f"| {label} | {e['paths_per_s']:.4e} | {e['kernel_us'] * scale:.4g} {unit} | {e['frac']} | {e['issue_frac']} / {e['issue_frac_clk']} at {e['sclk_mhz']:.0f} MHz | "
               f"{e['err']} {vs} | {e['cpu_baseline']['value']:.3e} {cpu} |")
i0, i1 = d.index("| C2 vanilla 1e8 fp32 |"), d.index("\n\nAll ten workloads, kernel alone")
d = d[:i0] + "\n".join(new) + d[i1:]
# ---- the default command's sentence
r = dflt["roofline"]
d = sub(r"\*\*" + NUM + r" paths/s\*\* \(" + NUM + r" µs per step\), lone launch " + NUM + r" µs → flop `frac` " + NUM + r", `issue_frac` " + NUM + r" \(" + NUM + r" at\s+the measured " + NUM
        + r" MHz\), \|price − BS\| " + NUM + r" over 5e11 paths; fp64 side " + NUM + r"; the driver's `--steps 20 --warmup 5`: " + NUM + r" and " + NUM,
        f"**{dflt['value']:.4g} paths/s** ({dflt['ms_per_step'] * 1e3:.2f} µs per step), lone launch {r['avg_kernel_us']:.2f} µs → flop `frac` {r['frac']:.3f}, `issue_frac` {r['issue_frac']:.3f} "
        f"({r['issue_frac_at_measured_clock']:.3f} at\nthe measured {r['sclk_mhz']:.0f} MHz), |price − BS| {dflt['price_error_vs_black_scholes']} over 5e11 paths; fp64 side {dflt['fp64']['value']:.3g}; "
        f"the driver's `--steps 20 --warmup 5`: {s1['value']:.4g} and {s2['value']:.4g}", d)
d = d.replace("e+12 paths/s**", "e12 paths/s**")
d = re.sub(r"(`--steps 20 --warmup 5`: )(\d\.\d+)e\+12 and (\d\.\d+)e\+12", r"\1\2e12 and \3e12", d)
d = re.sub(r"fp64 side (\d\.\d+)e\+11", r"fp64 side \1e11", d)
# ---- driver wall time, bytes; rocprof
wall = re.search(r"real\s+(\dm[0-9.]+s)", open(os.path.join(P, f"{TAG}_bench_driver_style_1.log")).read()).group(1)
nbytes = len([x for x in open(os.path.join(P, f"{TAG}_bench_driver_style_1.log")).read().splitlines() if x.startswith("{")][0])
d = sub(r"\dm[0-9.]+s and prints `\d+` bytes", f"{wall} and prints `{nbytes}` bytes", d)
row = [x for x in open(os.path.join(P, f"{TAG}_rocprofv3_kernel_stats_bench_streams1.csv")).read().splitlines() if "vanilla_f32_kernel" in x][0].split('",')
calls, avg = row[1].split(",")[0], row[1].split(",")[2]
d = sub(r"rocprofv3 agrees: [0-9.]+ ns average over \d+ launches", f"rocprofv3 agrees: {avg} ns average over {calls} launches", d)
# ---- scaling table, this round's cells = the driver-style run
ss = s1["strong_summary"]
for label, key, r05 in (("| C4 basket n=16 fp64, 1e9 paths |", "C4", ("27.9101, 27.9603", "3.51348", "0.992966", "0.93158")),
                        ("| C5 CVA 256 dates fp64, 1e7 paths |", "C5", ("7.9505, 7.931", "1.0354", "0.959836", "0.824935")),
                        ("| C4 on fp32 normals |", "C4_n32", ("15.9126", "2.01797", "0.985685", "0.855255")),
                        ("| C5 on fp32 normals |", "C5_n32", ("5.80672", "0.766926", "0.946427", "0.812965"))):
    cells = [f"{a} / {b[0]}, {b[1]}" for a, b in zip(r05, ss[key])]
    d = "\n".join((f"{label} " + " | ".join(cells) + " |") if ln.startswith(label) else ln for ln in d.split("\n"))
# ---- small calls (one synchronous call, 256 dates fp64; kernel times of forced lane counts)
lat = open(os.path.join(P, f"{TAG}_cva_call_latency.log")).read()
forced = lat[lat.index("forced lane counts"):]
k64 = re.search(r"^\s+4096\s+256\s+f64 \|\s+(\S+)(?:\s+\S+){5} \| (\S+)", forced, re.M)
t64 = re.search(r"^\s+65536\s+256\s+f64 \|\s+(\S+)(?:\s+\S+){5} \| (\S+)", forced, re.M)
own = re.search(r"^\s+131072\s+256\s+f64 \|\s+(\S+)\s+\S+\s+(\S+)", forced, re.M)   # L = 1 and L = 4
k32 = re.search(r"^\s+65536\s+256\s+f32 \|\s+(\S+)\s+\S+\s+(\S+)", forced, re.M)
d = sub(r"4096 paths: [0-9.]+ → [0-9.]+ µs, 65 536 paths: [0-9.]+ → [0-9.]+ µs\.",
        f"4096 paths: {k64.group(1)} → {k64.group(2)} µs, 65 536 paths: {t64.group(1)} → {t64.group(2)} µs.", d)
d = sub(r"keeps one lane per path \([0-9.]+ µs against [0-9.]+ µs on 4 lanes\)", f"keeps one lane per path ({own.group(1)} µs against {own.group(2)} µs on 4 lanes)", d)
d = sub(r"\(65 536 paths: [0-9.]+ → [0-9.]+ µs\)", f"(65 536 paths: {k32.group(1)} → {k32.group(2)} µs)", d)
# ---- secondary estimators
sec = json.load(open(os.path.join(P, f"{TAG}_bench_secondary_estimators.json")))["workloads"]
i0, i1 = d.index("The secondary estimators on the same kernels"), d.index("No row reaches 1.00 at either clock")
d = d[:i0] + (f"The secondary estimators on the same kernels [`profiles/{TAG}_bench_secondary_estimators.json`, `profiles/{TAG}_bench_all_workloads.log`]: antithetic vanilla f32 "
              f"{sec['vanilla_f32_anti']['value']:.4g} pairs/s (issue {sec['vanilla_f32_anti']['issue_frac']:.3f}), antithetic basket n=16 f64 {sec['basket16_f64_anti']['value']:.4g} "
              f"({sec['basket16_f64_anti']['issue_frac']:.3f}), control-variate basket n=16 f64 {sec['basket16_f64_cv']['value']:.4g} ({sec['basket16_f64_cv']['issue_frac']:.3f}; half-width "
              f"{sec['basket16_f64_cv']['confidence_95']:.2g} against the plain estimator's 0.000055).\n\n") + d[i1:]
open(D, "w", encoding="utf-8").write(d)

# ---- README
R = os.path.join(ROOT, "README.md")
s = open(R, encoding="utf-8").read()
rows = [("C2 vanilla, 1e8 paths fp32 (headline, median of five 1000-step regions)", "C2", "µs", 1, "vs Black-Scholes", True), ("C3 basket, 4 assets, 1e8 paths fp32", "C3", "µs", 1, "vs fp64", False),
        ("C4 basket, 16 assets, 1e9 paths fp64", "C4", "ms", 1e-3, "vs 1e10 paths", False), ("C5 CVA, 256 dates × 1e7 paths fp64", "C5", "ms", 1e-3, "vs closed form", False)]
new = []
for label, k, unit, scale, vs, bold in rows:
    e = c[k]
    v = f"{e['paths_per_s']:.4e}"
    new.append(f"| {label} | {'**' + v + '**' if bold else v} | {e['kernel_us'] * scale:.4g} {unit} | {e['frac']} | {e['issue_frac']} / {e['issue_frac_clk']} | {e['err']} {vs} | {e['cpu_baseline']['value']:.3e} |")
i0, i1 = s.index("| C2 vanilla, 1e8 paths fp32"), s.index("\n\nThe driver's 20-step run gives")
s = s[:i0] + "\n".join(new) + s[i1:]
s = sub(r"The driver's 20-step run gives " + NUM + r" and " + NUM, f"The driver's 20-step run gives {s1['value']:.4g} and {s2['value']:.4g}", s)
s = re.sub(r"(20-step run gives )(\d\.\d+)e\+12 and (\d\.\d+)e\+12", r"\1\2e12 and \3e12", s)
s = sub(r"workloads 0\.\d+ at the clock measured during the launches", f"workloads {mx:.3f} at the clock measured during the launches", s)
open(R, "w", encoding="utf-8").write(s)
# ---- profiles/README
PR = os.path.join(P, "README.md")
s = open(PR, encoding="utf-8").read()
s = sub(r"the ONE stdout line \(\d+ bytes\)", f"the ONE stdout line ({len(json.dumps(dflt))} bytes)", s)
s = sub(r"\([0-9.]+ s of wall time with four configs", f"({float(wall.split('m')[1][:-1]) + 60 * int(wall[0]):.1f} s of wall time with four configs", s)
s = sub(r"[0-9.]+ µs average over [0-9 ]+ launches single-stream", f"{float(avg) / 1e3:.2f} µs average over {int(calls):,}".replace(",", " ") + " launches single-stream", s)
s = sub(r"\(largest 0\.\d+: no row at or above 1\)", f"(largest {mx:.3f}: no row at or above 1)", s)
open(PR, "w", encoding="utf-8").write(s)
print("refreshed; now run: python tools/check_design_numbers.py && python tools/check_design_numbers.py README.md")
