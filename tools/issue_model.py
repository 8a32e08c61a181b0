#!/usr/bin/env python3
"""The VALU issue CEILING of every bench kernel, from its own instruction stream (VERDICT r03 "next" #2).

    make -C montecarlocuda_amd/csrc asm            # the ISA listing (mc_api.gfx950.s)
    python tools/issue_model.py [--tag r04]        # -> profiles/issue_model.json + profiles/<tag>_issue_model_<workload>.txt

Model.  A SIMD issues one VALU instruction of one wave at a time; an instruction occupies it for a number of cycles that
depends on the opcode.  `tools/ubench` measured them on MI355X at 8 waves per SIMD (profiles/r03_ubench_8waves.log): every
cost sits 1-4 % above a power of two -- 4.08-4.25 for the full-rate instructions (a wave64 instruction on a 16-lane SIMD IS four
cycles), 8.05-8.17 for the fp32 transcendentals, 16.2-16.4 for fp64 rcp / sqrt / rsq, 2.31-2.34 for the simple 32-bit ops in a
homogeneous run -- and the excess is the measurement's (loop overhead, ramp and tail of a 1.8 ms launch inside the event bracket),
not the instruction's.  Round 4-5 priced the ceiling at the cheapest MEASURED cost; one kernel then sat at 1.000 of it at
2.4 GHz while the chip sustains 2.29 GHz under that load: such a "ceiling" is not one.  Since round 6 every
opcode is priced at the ARCHITECTURAL cost of its class -- 4 / 8 / 16 cycles, 2 for the simple ops and for anything the ubench
never timed alone -- and the class is what the ubench decides.  So

    ceiling = sum over the hot path's VALU instructions of arch_cost(class(opcode))  x  wave-trips per SIMD  /  2.4 GHz

is a time no launch of that kernel can beat at any clock the chip can run at: `issue_frac` = ceiling / measured duration < 1, and
bench.py also prices it at the clock it measures during the launches (issue_frac_at_measured_clock), which must stay below 1 too
(tests/test_bench_cli.py asserts both over the committed all-workloads record).  The gap to 1 is what co-issue rules, operand-port
conflicts, LDS waits and launch ramp/tail cost; `typical_us` prices the same histogram at the costs seen in MIXED streams
(4.1 / 8.1 / 16.2) and is reported beside it, as an estimate, not a bound.

Hot path.  The innermost loop that holds the generator (Philox multiplies).  A single-block loop is its own hot path.  The
CVA kernels' date loops have several blocks (a Philox block on 3 of 4 trips in fp64, single-date tails that the BASELINE
grid never enters): their blocks are weighted by the rules in PATHS below, and EVERY model is cross-checked against the
hardware's own instruction counters (profiles/pmc_traffic.json: SQ_INSTS_VALU, _TRANS_F32, _TRANS_F64 per wave): the
histogram's VALU count per wave must agree with the counted one within 1 % -- the tool says so per kernel and bench.py
withholds the model for a kernel that fails.
"""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ASM = os.path.join(ROOT, "montecarlocuda_amd", "csrc", "mc_api.gfx950.s")
UBENCH = os.path.join(ROOT, "profiles", "r03_ubench_8waves.log")
SIMDS, CLOCK_HZ = 1024, 2.4e9      # 256 CUs x 4 SIMDs; the chip's peak shader clock (the bound must hold at any clock)

# ---- per-opcode issue cost (cycles per wave64 instruction on one SIMD) ---------------------------------------------------
# (regex on the opcode, ubench row(s) the cost comes from, class for the "typical" estimate)
COST_RULES = [
    (r"^v_(exp|log|sin|cos|sqrt|rcp|rsq)_f32", ["v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32"], "trans32"),
    (r"^v_(rcp|sqrt|rsq)_f64", ["v_rcp_f64", "v_sqrt_f64", "v_rsq_f64"], "trans64"),
    (r"^v_mad_u64_u32", ["v_mad_u64_u32"], "full"),
    (r"^v_bitop3_b32", ["v_bitop3_b32"], "full"),
    (r"^v_(fma|fmac|fmamk|fmaak)_f32", ["v_fma_f32", "v_fmac_f32", "v_fma_f32 dep", "v_fma 3 fresh src"], "full"),
    (r"^v_pk_(fma|mul|add)_f32", ["v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"], "full"),
    (r"^v_(fma|fmac|add|mul)_f64", ["v_fma_f64", "v_add_f64", "v_mul_f64", "v_fmac_f64"], "full"),
    (r"^v_(max|min)_f64", ["v_max_f64"], "full"),
    (r"^v_ldexp_f64", ["v_ldexp_f64"], "full"),
    (r"^v_frexp_", ["v_frexp_mant_f64", "v_frexp_exp_i32_f64"], "full"),
    (r"^v_cvt_f64_", ["v_cvt_f64_i32"], "full"),
    (r"^v_cvt_", ["v_cvt_f32_u32"], "full"),
    (r"^v_alignbit_b32", ["v_alignbit_b32"], "full"),
    (r"^v_mov_b64", ["v_mov_b64"], "full"),
    (r"^v_lshlrev_b64|^v_lshl_add_u64|^v_lshrrev_b64", ["v_lshlrev_b64"], "full"),
    (r"^v_(max|min)_(f32|i32|u32)", ["v_max_f32", "v_min_f32", "v_max_i32"], "full"),
    (r"^v_mul_(lo|hi)_u32|^v_mul_u32_u24|^v_mad_u32_u24", ["v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24"], "full"),
    (r"^v_mov_b32_dpp|_dpp$", ["v_mov_b32_dpp"], "full"),
    (r"^v_readlane_b32|^v_readfirstlane_b32|^v_writelane_b32", ["v_readlane_b32 (+s_xor)"], "full"),
    (r"^v_lshlrev_b32", ["v_lshlrev_b32"], "full"),
    (r"^v_(add|sub|subrev|mul)_f32", ["v_add_f32", "v_mul_f32", "v_sub_f32", "v_add_f32 dep", "v_add_f32(sgpr)", "v_sub_f32 clamp"], "simple"),
    (r"^v_(xor|and|or|not)_b32", ["v_xor_b32", "v_and_b32", "v_or_b32", "v_xor dep-chain", "v_xor_b32(sgpr)"], "simple"),
    (r"^v_(add|sub|subrev)_(u32|co_u32|i32)|^v_add3_u32|^v_lshl_add_u32|^v_add_lshl_u32", ["v_add_u32"], "simple"),
    (r"^v_lshrrev_b32|^v_ashrrev_i32|^v_bfe_", ["v_lshrrev_b32"], "simple"),
    (r"^v_mov_b32|^v_accvgpr", ["v_mov_b32"], "simple"),
]
# opcodes the ubench never timed alone (compares, selects, conversions of other widths ...): priced at the cheapest cost any
# VALU instruction was measured at -- the bound stays a bound
TYPICAL = {"simple": 4.1, "full": 4.1, "trans32": 8.1, "trans64": 16.2, "other": 4.1}
# the bound: cycles a wave64 instruction of the class occupies its SIMD for, by construction of the hardware (16 lanes per cycle:
# 4 passes; the transcendental unit at a quarter / an eighth of that rate; simple 32-bit ops two lanes' worth per cycle)
ARCH = {"simple": 2.0, "full": 4.0, "trans32": 8.0, "trans64": 16.0, "other": 2.0}


def ubench_costs():
    rows = {}
    for line in open(UBENCH):
        m = re.match(r"^(.*?)\s+(\d+\.\d+)\s+(\d+)\s+(\d+\.\d+)\s+(\d+)\s*$", line.rstrip())
        if m:
            rows[m.group(1).strip()] = float(m.group(4))
    return rows


UB = ubench_costs()
FLOOR = min(v for k, v in UB.items() if "/" not in k and "x3" not in k and "pair" not in k)


def cost_of(op):
    """(architectural cost of the opcode's class, class); the class comes from the ubench row(s) the opcode was timed in"""
    for pat, names, cls in COST_RULES:
        if re.search(pat, op):
            return ARCH[cls], cls
    return ARCH["other"], "other"


def measured_cost_of(op):
    for pat, names, cls in COST_RULES:
        if re.search(pat, op):
            return min(UB[n] for n in names if n in UB)
    return FLOOR


# ---- the hot path of a kernel ---------------------------------------------------------------------------------------------
def kernel_body(txt, pattern):
    names = [m for m in re.findall(r"^(_ZN2mc\w+):", txt, re.M) if pattern in m]
    if len(names) != 1:
        raise SystemExit(f"{pattern}: {len(names)} kernels match")
    body = re.search(r"^" + re.escape(names[0]) + r":[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M).group(1).split("\n")
    return names[0], body


def basic_blocks(body):
    """[(name, loop header or None, depth, [opcodes])] from hipcc's own listing: a block starts at a `.LBBx_y:` label or a
    `; %bb.N:` comment, ends at a branch; the loop a block belongs to is what LLVM's comments say ("in Loop: Header=BBx_y
    Depth=d", "=>This Inner Loop Header: Depth=d")."""
    out = []
    cur = {"name": "entry", "header": None, "depth": 0, "ops": []}

    def start(name):
        nonlocal cur
        if cur["ops"] or cur["name"] != "entry":
            out.append(cur)
        cur = {"name": name, "header": None, "depth": 0, "ops": []}
    for line in body:
        t = line.strip()
        m = re.match(r"^\.L(BB\d+_\d+):", t)
        if m:
            start(m.group(1))
        m2 = re.match(r"^; %bb\.(\d+):", t)
        if m2:
            start("%bb." + m2.group(1))
        if ";" in t:
            c = t[t.index(";"):]
            h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
            if h and cur["header"] is None and not cur["ops"]:
                cur["header"], cur["depth"] = h.group(1), int(h.group(2))
            h = re.search(r"=>\s*This (Inner )?Loop Header: Depth=(\d+)", c)
            if h and not cur["ops"]:
                cur["header"], cur["depth"] = cur["name"], int(h.group(2))
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        cur["ops"].append(op)
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["target"] = t.split()[-1].lstrip(".L")
            cur["uncond"] = op == "s_branch"
            hdr, dep = cur["header"], cur["depth"]
            start(cur["name"] + "+")          # the fall-through part of the same labelled region: same loop
            cur["header"], cur["depth"] = hdr, dep
    out.append(cur)
    return [b for b in out if b["ops"] or not b["name"].endswith("+")]


def n(ops, pat):
    return sum(1 for o in ops if re.search(pat, o))


T32 = r"^v_(exp|log|sin|cos|sqrt|rcp|rsq)_f32"
T64 = r"^v_(rcp|sqrt|rsq)_f64"
PHILOX = r"^v_mad_u64_u32"
STORE = r"^(global|flat|buffer)_store"

# ---- block weights: executions per trip of the hot loop -------------------------------------------------------------------
# A rule = (why, predicate on the block's opcode list, weight); first match wins, default 1.  The rules say in words what
# the branch conditions are; the hot-loop slope from the hardware counters (pmc_traffic.json "hot_loop_slope": counts at
# twice the paths minus counts at once the paths) checks them -- a wrong weight shows as a VALU / transcendental mismatch.
DUMP = ("per-path dump: `out` is NULL in pricing calls", lambda ops: n(ops, STORE) > 0, 0.0)


def flush(every):
    return (f"fp32 partial sums flushed to fp64 every {every}th trip", lambda ops: n(ops, r"^v_cvt_f64_f32") > 0, 1.0 / every)


def spec(pattern, units=None, dates_per_trip=None, rules=(), note=""):
    return {"pattern": pattern, "units": units, "dates_per_trip": dates_per_trip, "rules": list(rules), "note": note}


def cva_rules(philox_share, philox_blocks):
    return [DUMP,
            (f"generator: {philox_blocks} Philox block(s) share {philox_share:g} executions per trip (pair cursor / block of four normals)",
             lambda ops: n(ops, PHILOX) >= 8, philox_share / philox_blocks),
            ("single-date path (cva_single_date): never entered on a grid of 256 closed-form dates", lambda ops: "SINGLE" in ops, 0.0)]


WORKLOADS = {
    "vanilla_f32": spec("vanilla_f32_kernelILb0ENS_9GenPhiloxEEE", units=4, rules=[flush(8)]),
    "vanilla_f64": spec("vanilla_kernelINS_10VanillaF64EdLb0ENS_9GenPhiloxEEE", units=8),
    "vanilla_f64_n32": spec("vanilla_kernelINS_10VanillaF64EdLb0ENS_13GenPhiloxF32NEEE", units=4),
    "basket4_f32": spec("basket_f32_kernelILi4ELb0ENS_9GenPhiloxEEE", units=2, rules=[DUMP, flush(8)]),
    "basket16_f32": spec("basket_tiled_f32_kernelILi16ELb0ENS_9GenPhiloxEEE", units=2, rules=[DUMP]),
    "basket16_f64": spec("basket_tiled_kernelIdLi16ELb0ENS_9GenPhiloxEEE", units=1, rules=[DUMP]),
    "basket16_f64_n32": spec("basket_tiled_kernelIdLi16ELb0ENS_13GenPhiloxF32NEEE", units=1, rules=[DUMP]),
    "cva256_f64": spec("cva_kernelIdLb0ENS_9GenPhiloxEEE", dates_per_trip=8),
    "cva256_f64_n32": spec("cva_kernelIdLb0ENS_13GenPhiloxF32NEEE", dates_per_trip=8),
    "cva256_f32": spec("cva_kernelIfLb0ENS_9GenPhiloxEEE", dates_per_trip=4, note="lds_rows"),
    # secondary estimators (no performance record): antithetic = the ANTI instantiation of the same kernel (a unit is a
    # mirrored pair), control variate = the plain instantiation with its wave-uniform `cv` branches taken
    "vanilla_f32_anti": spec("vanilla_f32_kernelILb1ENS_9GenPhiloxEEE", units=4, rules=[flush(8)]),
    "basket16_f64_anti": spec("basket_tiled_kernelIdLi16ELb1ENS_9GenPhiloxEEE", units=1, rules=[DUMP], note="anti"),
    "basket16_f64_cv": spec("basket_tiled_kernelIdLi16ELb0ENS_9GenPhiloxEEE", units=1, rules=[DUMP], note="cv"),
}


# The fp64 CVA date loops since round 6 (mc_kernels.hpp: cva_path<double>): a GROUP loop of four Box-Muller pairs = eight dates per trip
# (one basic block, the cursor's phase a compile-time constant in each copy) while eight closed-form dates remain, then one pair per trip.
# The bench grid (T = 1 in 256 dates) has 255 closed-form dates and the maturity date: 31 group trips, then pairs 124, 125, 126 one at a
# time (closed-form; their cursor phases 0, 1, 2 draw one Philox block each -- the fp32-normals cursor draws one block per two pairs)
# and the pair (date 254, maturity) through cva_single_date, whose ~100 instructions per path the bound leaves out.
GROUPED = {
    "cva256_f64": {"group_trips": 31, "rest_trips": 4, "rest_closed": 3, "rest_philox_runs": 3},
    "cva256_f64_n32": {"group_trips": 31, "rest_trips": 4, "rest_closed": 3, "rest_philox_runs": 2},
}


def price(hist):
    cyc_min = cyc_typ = 0.0
    for op, cnt in hist.items():
        c, cls = cost_of(op)
        cyc_min += cnt * c
        cyc_typ += cnt * TYPICAL[cls]
    return cyc_min, cyc_typ


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--show", default="", help="print the block table of this workload and stop")
    args = ap.parse_args()
    import bench
    txt = open(ASM).read()
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    stamp = bench.launch_stamp()
    out = {}
    for wl, sp in WORKLOADS.items():
        if args.show and wl != args.show:
            continue
        name, body = kernel_body(txt, sp["pattern"])
        blocks = basic_blocks(body)
        # the hot region: every block of the outermost loop that (with its child loops) holds the generator's Philox multiplies
        parent = {}
        for bk in blocks:
            if bk["header"] == bk["name"] and bk.get("parent"):
                parent[bk["name"]] = bk["parent"]
        ph_headers = {bk["header"] for bk in blocks if bk["header"] and n(bk["ops"], PHILOX) >= 8}
        depth_of = {bk["name"]: bk["depth"] for bk in blocks if bk["header"] == bk["name"]}
        # layout range of the loops that hold Philox, widened to the shallowest loop enclosing them (LLVM lays a loop out contiguously)
        idx = [i for i, bk in enumerate(blocks) if bk["header"] in ph_headers]
        lo_i, hi_i = min(idx), max(idx)
        top = min(blocks[i]["depth"] for i in range(lo_i, hi_i + 1) if blocks[i]["depth"] > 0)
        while lo_i > 0 and blocks[lo_i - 1]["depth"] >= top and blocks[lo_i - 1]["header"]:
            lo_i -= 1
        while hi_i + 1 < len(blocks) and blocks[hi_i + 1]["depth"] >= top and blocks[hi_i + 1]["header"]:
            hi_i += 1
        loop = blocks[lo_i:hi_i + 1]
        header, depth = loop[0]["header"], top
        region = {bk["name"] for bk in loop}
        committed = pmc.get(wl, {})
        paths = committed.get("paths_per_launch")
        is_cva = sp["dates_per_trip"] is not None
        big = max(loop, key=lambda bk: n(bk["ops"], r"^v_"))

        def is_dump(ops):
            return n(ops, STORE) > 0

        def is_single(ops):
            # cva_single_date (a grid's odd last date, the intrinsic-value date) / the baskets' control variate: a block that
            # evaluates an exponential but is neither the generator (Philox, the Box-Muller block with its fp64 rsq / fp32 log)
            # nor the main block (most VALU)
            if ops is big["ops"]:
                return False
            if sp.get("note") == "anti" and n(ops, r"^v_") >= 100:
                return False   # the mirrored half of the pair (its exponentials and payoff): a block of its own in the ANTI instantiation, always run
            gen = n(ops, PHILOX) or n(ops, r"^v_(sqrt|rsq)_f64|^v_log_f32|^v_sin_f32")
            # (the fp32 CVA loop has TWO paired-exposure blocks per trip of four dates, each nearly the size of the biggest)
            return not gen and n(ops, r"^v_") < 0.7 * n(big["ops"], r"^v_") and n(ops, r"^v_(exp_f32|rcp_f32|rcp_f64|ldexp_f64)") > 0

        cv_on = sp.get("note") == "cv"   # the control variate's blocks run: only the dump stays cold

        def cold(ops):
            return is_dump(ops) or (is_single(ops) and not cv_on)

        def periodic(ops):   # the every-8th-trip flush of the packed fp32 sums: its own small block in the two kernels that have one
            return 1.0 / 8 if (wl in ("vanilla_f32", "vanilla_f32_anti", "basket4_f32") and n(ops, r"^v_cvt_f64_f32") > 0 and n(ops, PHILOX) == 0) else None
        # executions per trip of the hot loop.  No dynamic trace is to be had here (rocprofv3's thread trace needs a decoder the
        # image lacks), so the weights are rules, each a statement about a wave-uniform branch whose outcome is known for a
        # pricing call of the bench, and the hardware counters check the sum (below):
        #   1     every block of the loop nest, except
        #   0     per-path dump blocks (`out` is NULL), blocks of the control variate (off) and of the CVA's single-date path
        #         (a grid of 256 closed-form dates never enters it), and the blocks only they lead to;
        #   1/8   the flush of the fp32 partial sums (every 8th trip);
        #   CVA   the generator's Philox blocks by the stream layout: fp64 three blocks on 3 of 4 pair-trips, fp64 on fp32
        #         normals one block every second pair-trip, fp32 one block per trip of four dates; blocks of the path loop
        #         outside the date loop once per 256 dates.
        if not is_cva:
            unit_note = f"{sp['units']} paths per lane and trip"
            trips_per_path = 1.0 / sp["units"]
        else:
            unit_note = f"{sp['dates_per_trip']} dates of one path per lane and trip, 256 dates per path"
            trips_per_path = 256.0 / sp["dates_per_trip"]
        inner_depth = max(bk["depth"] for bk in loop)
        weight, why = {}, {}
        for bk in loop:
            nm, ops = bk["name"], bk["ops"]
            if is_dump(ops):
                weight[nm], why[nm] = 0.0, "per-path dump: `out` is NULL in pricing calls"
            elif is_single(ops) and not cv_on:
                weight[nm], why[nm] = 0.0, ("single-date path (cva_single_date): a grid of 256 closed-form dates never enters it" if is_cva
                                            else "control variate: off in the bench's plain estimator")
            elif sp.get("note") == "lds_rows" and n(ops, r"^v_rcp_f32") >= 4 and n(ops, r"^ds_read") == 0:
                # cva_path<float> holds both forms of the packed date pair behind a launch-uniform branch: rows from LDS (three ds_read_b128)
                # or from scalar registers (grids whose pair rows exceed 16 KB); the bench grid of 256 dates takes the first
                weight[nm], why[nm] = 0.0, "date pair on scalar-register rows: this launch carries the rows in LDS (mc_api.hip: pairs_fit)"
            elif periodic(ops):
                weight[nm], why[nm] = 1.0 / 8, "fp32 partial sums flushed to fp64 every 8th trip"
            elif is_cva and bk["depth"] < inner_depth and n(ops, PHILOX) < 8:
                weight[nm], why[nm] = 1.0 / trips_per_path, "path loop, outside the date loop: once per path"
            else:
                weight[nm] = 1.0
        phb = [bk for bk in loop if n(bk["ops"], PHILOX) >= 8]
        share = {"cva256_f32": 1.0}.get(wl)
        grp = GROUPED.get(wl)
        if grp:
            # weights = executions per path / 32: a "trip" is eight dates, the group loop's own
            assert n(big["ops"], PHILOX) >= 8 and big["depth"] == inner_depth, "the group loop's block was expected to be the biggest"
            rest_ph = [bk for bk in loop if bk is not big and n(bk["ops"], PHILOX) >= 8]   # (hipcc may place one in front of the one-pair loop)
            rest = [bk for bk in loop if (bk["depth"] == inner_depth and bk is not big) or bk in rest_ph]
            closed = max((bk for bk in rest if bk not in rest_ph), key=lambda bk: n(bk["ops"], r"^v_"))
            weight[big["name"]], why[big["name"]] = grp["group_trips"] / trips_per_path, f"group loop: four pairs = eight dates per trip, {grp['group_trips']} trips per path"
            for bk in rest:
                nm = bk["name"]
                if bk is closed:
                    weight[nm], why[nm] = grp["rest_closed"] / trips_per_path, f"one-pair loop, closed-form pair: {grp['rest_closed']} times per path"
                elif bk in rest_ph:
                    weight[nm] = grp["rest_philox_runs"] / len(rest_ph) / trips_per_path
                    why[nm] = f"one-pair loop, generator: {len(rest_ph)} Philox block(s), {grp['rest_philox_runs']} executions per path between them"
                elif weight[nm] == 0.0:
                    why[nm] = "one-pair loop, single-date path (cva_single_date): once per path (date 254 and maturity), left out of the bound"
                else:
                    weight[nm], why[nm] = grp["rest_trips"] / trips_per_path, f"one-pair loop: {grp['rest_trips']} trips per path"
        if share is not None:
            for bk in phb:
                weight[bk["name"]] = share / len(phb)
                why[bk["name"]] = f"generator: {len(phb)} Philox block(s), {share:g} executions per trip between them (stream layout, mc_rng.hpp)"
        # a block that only cold blocks lead to is cold too (one pass in layout order is enough for these kernels)
        preds = collections.defaultdict(list)
        for i, bk in enumerate(loop):
            if bk.get("target"):
                preds[bk["target"]].append(bk["name"])
            if not bk.get("uncond") and i + 1 < len(loop):
                preds[loop[i + 1]["name"]].append(bk["name"])
        for bk in loop[1:]:
            ps = preds.get(bk["name"], [])
            if ps and all(weight.get(q, 1.0) == 0.0 for q in ps) and weight[bk["name"]] > 0.0 and n(bk["ops"], r"^v_") < 10:
                weight[bk["name"]], why[bk["name"]] = 0.0, "reached only from blocks that never run"
        hist = collections.Counter()
        rows = []
        for bk in loop:
            w = weight.get(bk["name"], 0.0)
            ops = bk["ops"]
            rows.append((bk["name"], len(ops), n(ops, r"^v_"), n(ops, PHILOX), n(ops, T32), n(ops, T64), n(ops, r"^ds_"), n(ops, r"^s_"), w, why.get(bk["name"], "")))
            for o in ops:
                if o.startswith(("v_", "s_", "ds_")):
                    hist[o] += w
        vh = {o: c for o, c in hist.items() if o.startswith("v_")}
        cyc_min, cyc_typ = price(vh)
        valu = sum(vh.values())
        t32 = sum(c for o, c in vh.items() if re.search(T32, o))
        t64 = sum(c for o, c in vh.items() if re.search(T64, o))
        salu = sum(c for o, c in hist.items() if o.startswith("s_") and not o.startswith(("s_load", "s_buffer_load", "s_waitcnt", "s_nop")))
        lds = sum(c for o, c in hist.items() if o.startswith("ds_"))
        lines = [f"# {wl}: {name}", f"# hot loop: header {header} (depth {depth}), {len(loop)} basic blocks; {unit_note}",
                 "# per-opcode cost = the architectural cost of the opcode's class (2 / 4 / 8 / 16 cycles; untimed opcodes: 2); `measured` = its cheapest "
                 "context in profiles/r03_ubench_8waves.log",
                 "block             instrs  VALU  philox-mul  trans32  trans64   LDS  scalar  executions per trip"]
        for r in rows:
            lines.append(f"  {r[0]:15s} {r[1]:6d} {r[2]:5d} {r[3]:11d} {r[4]:8d} {r[5]:8d} {r[6]:5d} {r[7]:7d}  {r[8]:<8.4g} {r[9]}")
        lines.append("opcode                 per trip   arch cycles   measured   class")
        for op, cnt in sorted(vh.items(), key=lambda kv: -kv[1] * cost_of(kv[0])[0]):
            c, cls = cost_of(op)
            lines.append(f"  {op:22s} {cnt:8.2f} {c:10.2f} {measured_cost_of(op):10.2f}   {cls}")
        rec = {"kernel": name, "valu_per_trip": valu, "trans_f32_per_trip": t32, "trans_f64_per_trip": t64, "min_cycles_per_trip": cyc_min,
               "typical_cycles_per_trip": cyc_typ, "trips_per_path": trips_per_path, "unit": unit_note, "launch_stamp": stamp["stamp"],
               "source": f"profiles/{args.tag}_issue_model_{wl}.txt", "simds": SIMDS, "clock_hz": CLOCK_HZ,
               # per path and lane: what bench.py scales by the paths of a launch
               "min_cycles_per_path": cyc_min * trips_per_path, "typical_cycles_per_path": cyc_typ * trips_per_path,
               "valu_per_path": valu * trips_per_path}
        lines += ["", f"per trip: {valu:.2f} VALU instructions ({t32:g} fp32 transcendental, {t64:g} fp64 rcp/sqrt/rsq), {salu:.1f} scalar ALU, {lds:g} LDS = "
                      f"{cyc_min:.1f} cycles at the architectural cost of every opcode's class ({cyc_typ:.1f} at mixed-stream costs 4.1 / 8.1 / 16.2)",
                  f"per path: {valu * trips_per_path:.3f} VALU instructions per lane, {cyc_min * trips_per_path:.2f} cycles of a wave (64 paths at once)"]
        if paths:
            wave_trips = paths * trips_per_path / 64.0
            rec["paths_per_launch"] = paths
            rec["ceiling_us"] = cyc_min * wave_trips / SIMDS / CLOCK_HZ * 1e6
            rec["typical_us"] = cyc_typ * wave_trips / SIMDS / CLOCK_HZ * 1e6
            lines.append(f"launch of {paths:g} paths = {wave_trips:.6g} wave-trips on {SIMDS} SIMDs at {CLOCK_HZ / 1e9} GHz: "
                         f"ceiling_us = {rec['ceiling_us']:.2f}   (typical_us = {rec['typical_us']:.2f}: an estimate, not a bound)")
        # ---- cross-check: the hardware's own counters, hot loop alone (counts at 2 x paths minus counts at 1 x paths) --------
        slope = committed.get("hot_loop_slope") or {}
        rec["pmc_stamp_matches"] = committed.get("launch_stamp") == stamp["stamp"]
        if slope.get("valu_wave_insts_per_path"):
            got = {k: slope.get(k + "_wave_insts_per_path", 0.0) * 64.0 for k in ("valu", "trans_f32", "trans_f64", "salu", "lds")}
            want = {"valu": valu * trips_per_path, "trans_f32": t32 * trips_per_path, "trans_f64": t64 * trips_per_path,
                    "salu": salu * trips_per_path, "lds": lds * trips_per_path}
            rec["pmc_slope_per_path"], rec["model_per_path"] = got, want
            rec["pmc_vs_model"] = got["valu"] / want["valu"]
            ok = abs(got["valu"] / want["valu"] - 1.0) <= 0.01
            for k in ("trans_f32", "trans_f64"):
                ok = ok and abs(got[k] - want[k]) <= 0.01 * max(want[k], 1e-9) + 1e-6
            rec["cross_check_ok"] = bool(ok and rec["pmc_stamp_matches"])
            lines += ["cross-check against the hardware counters, hot loop alone (" + str(committed.get("source")) + ": counts at twice the paths minus counts "
                      "at once the paths; stamp " + ("matches" if rec["pmc_stamp_matches"] else "DIFFERS -- re-collect") + "), per path and lane:"]
            for k in ("valu", "trans_f32", "trans_f64", "salu", "lds"):
                lines.append(f"  {k:10s} counted {got[k]:10.4f}   histogram {want[k]:10.4f}" + (f"   ratio {got[k] / want[k]:.4f}" if want[k] else ""))
            if ok:
                lines.append("  -> ok: VALU within 1 %, transcendentals exact")
            else:
                # The rules' sum misses the counted instructions by more than 1 % (small glue blocks of a multi-block loop whose
                # execution counts the rules cannot know).  The COUNT is then taken from the hardware and only the MIX from the
                # histogram: transcendentals at their counted numbers, the remaining instructions at the histogram's average cost.
                rest_w = want["valu"] - want["trans_f32"] - want["trans_f64"]
                rest_g = got["valu"] - got["trans_f32"] - got["trans_f64"]
                c32, c64 = cost_of("v_exp_f32")[0], cost_of("v_rcp_f64")[0]
                rest_cyc = cyc_min * trips_per_path - want["trans_f32"] * c32 - want["trans_f64"] * c64
                rest_typ = cyc_typ * trips_per_path - want["trans_f32"] * TYPICAL["trans32"] - want["trans_f64"] * TYPICAL["trans64"]
                scaled_min = rest_cyc / rest_w * rest_g + got["trans_f32"] * c32 + got["trans_f64"] * c64
                scaled_typ = rest_typ / rest_w * rest_g + got["trans_f32"] * TYPICAL["trans32"] + got["trans_f64"] * TYPICAL["trans64"]
                rec["rescaled_to_counters"] = True
                rec["unscaled_min_cycles_per_path"] = rec["min_cycles_per_path"]
                rec["min_cycles_per_path"], rec["typical_cycles_per_path"], rec["valu_per_path"] = scaled_min, scaled_typ, got["valu"]
                if paths:
                    rec["ceiling_us"] = scaled_min * paths / 64.0 / SIMDS / CLOCK_HZ * 1e6
                    rec["typical_us"] = scaled_typ * paths / 64.0 / SIMDS / CLOCK_HZ * 1e6
                lines.append(f"  -> the rules miss the count by {100 * (want['valu'] / got['valu'] - 1):+.1f} %: instruction COUNTS taken from the counters, the cost MIX from the "
                             f"histogram: {scaled_min:.1f} cycles per path (ceiling_us = {rec.get('ceiling_us', float('nan')):.2f})")
        else:
            rec["cross_check_ok"] = False
            lines.append("cross-check: no hot-loop slope in profiles/pmc_traffic.json for this workload (tools/collect_pmc_all.sh collects it)")
        out[wl] = rec
        text = "\n".join(lines) + "\n"
        if args.show:
            print(text)
            return
        open(os.path.join(ROOT, "profiles", f"{args.tag}_issue_model_{wl}.txt"), "w").write(text)
        print(f"{wl:18s} VALU/path {valu * trips_per_path:9.3f}  min {cyc_min * trips_per_path:9.2f} cyc/path  ceiling {rec.get('ceiling_us', float('nan')):8.2f} us  "
              f"typical {rec.get('typical_us', float('nan')):8.2f} us  pmc/model {rec.get('pmc_vs_model', float('nan')):.4f} {'ok' if rec.get('cross_check_ok') else 'unchecked'}")
    json.dump(out, open(os.path.join(ROOT, "profiles", "issue_model.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
