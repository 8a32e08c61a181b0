#!/usr/bin/env python3
"""bench.py -- the hot path's throughput on N MI355X, one JSON line on rank 0.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A STEP is one pass of the hot path over one batch: one pricing call of the workload's path count
on every GPU (ONE kernel: simulation + in-kernel final reduction -> {sum, sum2, n} in HBM) and, for N > 1,
the RCCL all-reduce of that 24-byte triple.  Default workload = BASELINE.json configs[1]:
European vanilla call, 1 asset, 1e8 paths, fp32 simulation (fp64 accumulation), per GPU.

Scaling is WEAK: every GPU simulates `paths` paths per step, rank g taking the contiguous global
range [(step*N + g) * paths, +paths) of one Philox stream (no data-path collective besides the
triple).  Steps rotate over `--streams` (default 2) independent context/stream pairs so that one
pricing call's launch gap, ramp and tail overlap the next call's kernel -- fixed per-call costs
(~7 us of a ~57 us lone call) are hidden, each call is still one full launch.
The triples of `--bucket` consecutive steps are all-reduced as ONE message (fewer, larger
collectives: a 24-byte all-reduce is pure latency), asynchronously to the launch streams, which
keep simulating; every bucket is waited for inside the timed region.
Before the W warm-up steps the same kernel runs untimed for --preheat-ms (clock ramp; reported).
The timed region of exactly K steps is repeated --regions (5) times back to back, each bracketed by
barrier + torch.cuda.synchronize() on both sides with the MAX over ranks taken; ms_per_step, value
and timed_region_s are the MEDIAN region's (min / max / all regions reported beside them).  At N > 1
over RCCL the barrier that closes a region is the all-reduce of its last bucket of triples (an
all-reduce is a barrier; --closing barrier adds a separate dist.barrier()).

OUTPUT.  stdout carries ONE line: a JSON object of at most 4096 bytes -- the contract fields, `config`, `roofline`,
`cpu_baseline`, `fp64`, `configs` (BASELINE configs[1..4] with their roofline fractions and the reference's CPU path), `strong_summary` and,
at N > 1, the roster of ranks (`ranks`) -- built by
compact_line() from the full record.  The full record (every strong row, every line of the C library's child process,
host-side timings, the prose that explains each field) goes to --detail-file (default bench_detail.json next to this
script; the line names it as `detail`), never to stdout or stderr.  (Round 4's line had grown to 32 KB and the driver
could not parse it.)  --detail full adds the rows that --detail brief (default) leaves out: the x10 sizes on fp32 normals,
shard 0 of 2 and of 4, the -O0 build of the reference's CPU path.

In the full record, besides the contract fields:
  roofline      dominant kernel (the simulation kernel): achieved = algorithmic flop per launch (SURVEY 8d:
                15.5 flop/path vanilla, n^2+12.5n+6 basket, 60/path-step CVA) / its launch duration, measured live
                with HIP events bound to the dispatch over 50 launches one at a time right after the timed region
                (launches inside the region are co-resident: in_region); effective = per step period; issue_frac =
                issue-slot model from the committed PMC instruction counts; traffic / hbm_gbps from the committed
                PMC passes (tagged as such); the bound is VALU issue, not HBM or MFMA
  strong        every N: ONE call of C4 / C5 (and 10x) sharded over the N ranks, wall-clock to the all-reduced result
  c_multi       N = 1: the same rows through the C library alone (drivers/multiBench, libmc_multi.so + RCCL), child process
  cpu_baseline  the reference's own CPU path (oracle/_ref, compiled from MonteCarloHost.c) or, when
                that build is absent, the oracle port; one host core; rank 0 at N=1 only
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BS_EXACT = 10.386270784322328  # exact Black-Scholes for the vanilla workload

VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)                       # reference vanillaOpt.cu:22-26
CVA = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=256)  # cvaOpt.cu:22-34


def basket_inputs(mc, n, X):
    import numpy as np
    v = [0.3 if i % 2 == 0 else 0.2 for i in range(n)]
    L, bad = mc.chol(np.full((n, n), 0.5) + 0.5 * np.eye(n), X)
    assert bad == 0
    return dict(s=[100.0] * n, v=v, p=L.tolist(), d=[0.0] * n, w=[1.0 / n] * n, k=100.0, t=1.0, r=0.048790164)


def workloads(mc):
    f_basket = lambda n: n * n + 12.5 * n + 6  # noqa: E731
    return {
        # name: (product, precision, inputs, paths per GPU per step, flop per path, description)
        "vanilla_f32": ("vanilla", "f32", VAN, 10 ** 8, 15.5, "European vanilla call, 1 asset, 1e8 paths, fp32 (BASELINE configs[1])"),
        "vanilla_f64": ("vanilla", "f64", VAN, 10 ** 8, 15.5, "European vanilla call, 1 asset, 1e8 paths, fp64"),
        "basket4_f32": ("basket", "f32", lambda: basket_inputs(mc, 4, "f32"), 10 ** 8, f_basket(4), "Basket call, 4 correlated assets, 1e8 paths, fp32 (BASELINE configs[2])"),
        "basket16_f32": ("basket", "f32", lambda: basket_inputs(mc, 16, "f32"), 125 * 10 ** 6, f_basket(16), "Basket call, 16 correlated assets, 1e9/8 paths per GPU, fp32"),
        "basket16_f64": ("basket", "f64", lambda: basket_inputs(mc, 16, "f64"), 125 * 10 ** 6, f_basket(16), "Basket call, 16 correlated assets, 1e9/8 paths per GPU, fp64 (BASELINE configs[3])"),
        "cva256_f64": ("cva", "f64", CVA, 1250000, 60.0 * 256 + 5, "CVA on vanilla call, 256 dates x 1e7/8 paths per GPU, fp64 (BASELINE configs[4])"),
        "cva256_f32": ("cva", "f32", CVA, 1250000, 60.0 * 256 + 5, "CVA on vanilla call, 256 dates x 1e7/8 paths per GPU, fp32"),
        # the fp64 kernels on fp32 normals widened to double -- the reference's own dp arithmetic (dp/MonteCarloKernel.cu:68,78,250),
        # opt-in: mc_context_set_normals(ctx, MC_NORMALS_F32).  Normal generation counts 6.5 flop per normal either way.
        "vanilla_f64_n32": ("vanilla", "f64", VAN, 10 ** 8, 15.5, "European vanilla call, 1e8 paths, fp64 on fp32 normals (reference dp arithmetic)"),
        "basket16_f64_n32": ("basket", "f64", lambda: basket_inputs(mc, 16, "f64"), 125 * 10 ** 6, f_basket(16), "Basket call, 16 assets, 1e9/8 paths per GPU, fp64 on fp32 normals (reference dp arithmetic)"),
        "cva256_f64_n32": ("cva", "f64", CVA, 1250000, 60.0 * 256 + 5, "CVA, 256 dates x 1e7/8 paths per GPU, fp64 on fp32 normals (reference dp arithmetic)"),
        # the secondary estimators on the same kernels (the reference has the plain estimator only).  A "path" of the antithetic
        # estimator is a mirrored PAIR on one normal (6.5 + 2 x 9 flop vanilla; n^2 + n + 6.5 n + 2 (5 n + 6) basket); the control variate
        # adds the geometric basket's weighted log-sum and payoff (2 n + 4 flop)
        "vanilla_f32_anti": ("vanilla", "f32", VAN, 10 ** 8, 24.5, "European vanilla call, 1e8 antithetic pairs, fp32"),
        "basket16_f64_anti": ("basket", "f64", lambda: basket_inputs(mc, 16, "f64"), 125 * 10 ** 6, 16 * 16 + 17.5 * 16 + 12,
                              "Basket call, 16 assets, 1.25e8 antithetic pairs, fp64"),
        "basket16_f64_cv": ("basket", "f64", lambda: basket_inputs(mc, 16, "f64"), 125 * 10 ** 6, f_basket(16) + 2 * 16 + 4,
                            "Basket call, 16 assets, 1.25e8 paths, fp64, geometric-basket control variate"),
    }


def workload_settings(name):
    """Engine settings a workload runs under (besides the defaults)."""
    if name.endswith("_n32"):
        return {"normals": "f32"}
    if name.endswith("_anti"):
        return {"antithetic": True}
    if name.endswith("_cv"):
        return {"control_variate": True}
    return {}


def elf_section(path, name):
    """Bytes of one section of an ELF64 little-endian file (no binutils needed on the box)."""
    import struct
    b = open(path, "rb").read()
    if b[:4] != b"\x7fELF" or b[4] != 2:
        raise ValueError(f"{path}: not an ELF64 file")
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)

    def header(i):
        return struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize)
    strtab = header(shstrndx)[4]
    for i in range(shnum):
        n, _, _, _, off, size, *_ = header(i)
        if b[strtab + n:b.index(b"\0", strtab + n)].decode() == name:
            return b[off:off + size]
    raise KeyError(f"{path}: no section {name}")


def launch_stamp(root=ROOT):
    """Stamp of what a launch IS, for the committed PMC profiles (profiles/pmc_traffic.json, profiles/issue_model.json):
      device_code   sha256 of the .hip_fatbin section of the built libmc_mi355x.so -- the code object the device runs
                    (covers every kernel header, the math tables, the compiler and its flags; a rebuild of unchanged
                    sources reproduces it bit for bit, whatever the directory);
      launch_shape  sha256 of csrc/mc_launch_shape.hpp -- grid rules, kernel-family limits: what decides workgroups and
                    waves per launch without touching the device code;
      hipflags      the Makefile's HIPFLAGS line (a flag edit marks the counts stale even before the rebuild).
    `stamp` = sha256 of the three.  tools/summarize_pmc.py writes it next to the counts (with the grid each PMC pass ran),
    bench.py compares: stale counts are flagged and the instruction-count model is withheld."""
    import hashlib
    import re
    csrc = os.path.join(root, "montecarlocuda_amd", "csrc")
    so = os.path.join(csrc, "libmc_mi355x.so")
    try:
        device = hashlib.sha256(elf_section(so, ".hip_fatbin")).hexdigest()
    except (OSError, KeyError, ValueError):
        device = "library not built"
    shape = hashlib.sha256(open(os.path.join(csrc, "mc_launch_shape.hpp"), "rb").read()).hexdigest()
    flags = " ".join(m.strip() for m in re.findall(r"^HIPFLAGS\s*\??=.*$", open(os.path.join(csrc, "Makefile")).read(), re.M))
    return {"stamp": hashlib.sha256("\n".join((device, shape, flags)).encode()).hexdigest(), "device_code_sha256": device,
            "launch_shape_sha256": shape, "hipflags": flags}


def kernel_name(prod, X, inputs):
    """The simulation kernel a workload runs (montecarlocuda_amd/csrc/mc_api.hip picks it)."""
    if prod == "vanilla":
        return "mc::vanilla_f32_kernel" if X == "f32" else "mc::vanilla_kernel<f64>"
    if prod == "basket":
        n = len(inputs["s"])
        pad = (n + 3) // 4 * 4
        if X == "f32":
            if n <= 12:
                return f"mc::basket_f32_kernel<{n}>"
            return f"mc::basket_tiled_f32_kernel<{n if n <= 14 else pad}>" if n <= 32 else "mc::basket_dyn_f32_kernel"
        if n <= 8:
            return f"mc::basket_kernel<f64, {n}>"
        return f"mc::basket_tiled_kernel<f64, {n if n <= 16 else pad}>" if n <= 32 else "mc::basket_dyn_kernel<f64>"
    # (a call that ends in a partial wave-trip of at most 60 % runs as cva_split_kernel: the same one-lane-per-path loop plus
    # date-parallel workgroups for the remainder -- 1.25e6 paths = 19 trips + 4816 paths; csrc/mc_launch_shape.hpp: cva_plan)
    return f"mc::cva_kernel<{X}> (as cva_split_kernel when the call ends in a partial wave-trip)"


PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # MI355X vector peaks (MI355X_MICROARCH.md; fp64 vector = half)


def cpu_baseline(prod, X, inputs, seconds, full=False):
    """Time the reference's CPU path on ONE host core for about `seconds` seconds."""
    from oracle import pyoracle as po   # checker / baseline only
    po.build()
    n_assets = len(inputs["s"]) if prod == "basket" else 3
    use_ref = po.ref_available(X, n_assets)
    if prod == "vanilla":
        run = (lambda n: po.Ref(X, 3).vanilla(inputs, n, 12345)) if use_ref else (lambda n: po.host_vanilla(X, inputs, n, 12345))
        rate_guess, per_unit, unit = 1.5e7, 1, "paths/s"
    elif prod == "basket":
        run = (lambda n: po.Ref(X, n_assets).basket(inputs, n, 12345)) if use_ref else (lambda n: po.host_basket(X, inputs, n, 12345))
        rate_guess, per_unit, unit = 1.5e7 / n_assets, 1, "paths/s"
    else:
        run = (lambda n: po.Ref(X, 3).cva(inputs, n, 12345)) if use_ref else (lambda n: po.host_cva(X, inputs, n, 12345))
        rate_guess, per_unit, unit = 8e6 / inputs["n_grid"], 1, "paths/s"
    n = max(1000, int(rate_guess * 0.5))
    t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0   # calibration pass
    n = int(min(2 ** 31 - 1, max(n, n * seconds / max(dt, 1e-3))))
    t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
    out = {"value": n * per_unit / dt, "unit": unit, "cores": 1, "kind": "reference" if use_ref else "port",
           "sample_short": f"{n} paths of the same workload in {dt:.1f} s, 1 thread, gcc -O2",
           "sample": f"{n} paths of the same workload in {dt:.1f} s, single thread (the reference is single-threaded: "
                     f"MonteCarloHost.c:185-229), gcc -O2 -ffp-contract=off",
           "host_cores_available": len(os.sched_getaffinity(0)), "host_cpus_granted": cpus_granted()}
    if full and prod == "vanilla" and po.ref_available(X, 3, "_O0"):
        # footnote: the reference's own Makefile compiles the host file without -O (Makefile:157,252-253)
        n0 = max(1000, int(n * min(1.0, 3.0 / max(dt, 1e-3))))
        t0 = time.perf_counter(); po.Ref(X, 3, "_O0").vanilla(inputs, n0, 12345); dt0 = time.perf_counter() - t0
        out["value_at_O0"] = n0 / dt0
        out["O0_note"] = f"the same object built with gcc -O0, {n0} paths in {dt0:.1f} s (the reference Makefile's optimisation level)"
    return out


def cpus_granted():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), or None when unlimited / unknown: the GPU boxes show
    256 hardware threads and grant 16."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else -(-int(q) // int(p))
    except (OSError, ValueError):
        return None


class SclkSampler:
    """Shader clock of the card at PCI address `pci` while the block runs, from amdgpu's hwmon file (montecarlocuda_amd/sclk.py),
    sampled every millisecond by a side thread: .mhz = median of the second half of the samples (the first half may still
    be the ramp), None when the file does not exist."""

    def __init__(self, pci):
        from montecarlocuda_amd import sclk
        self._read = (lambda: sclk.read_mhz(0, pci)) if sclk.source(0, pci) else None
        self.mhz, self.samples = None, 0

    def __enter__(self):
        import threading
        self._stop, self._v = threading.Event(), []
        if self._read:
            def run():
                while not self._stop.is_set():
                    self._v.append(self._read())
                    time.sleep(0.001)
            self._th = threading.Thread(target=run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._read:
            self._th.join()
            v = sorted(x for x in self._v[len(self._v) // 2:] if x)
            self.samples = len(v)
            self.mhz = v[len(v) // 2] if v else None


def sustained_clock(torch, engines, streams, launch, pci, ms=150.0, preheat_ms=300.0):
    """Shader clock while `launch(i, engine, stream)` runs back to back on the given context/stream pairs: `preheat_ms` of it
    unsampled first (the hwmon figure is a moving average: read straight after a change of load it still shows the previous
    state -- 2017 MHz for a kernel that holds 2393), then `ms` milliseconds sampled."""
    i = 0

    def burn(for_ms):
        nonlocal i
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < for_ms:
            for _ in range(16):
                launch(i, engines[i % len(engines)], streams[i % len(streams)])
                i += 1
            torch.cuda.synchronize()
    burn(preheat_ms)
    with SclkSampler(pci) as sm:
        burn(ms)
    return sm.mhz


def issue_ceiling(workload, paths, kernel_s, step_s, stamp, traffic_stale, committed, sclk_mhz=None):
    """SURVEY 8d's second fraction: the VALU issue CEILING of one launch of `workload` over `paths` paths, from the kernel's own
    instruction stream -- the opcode histogram of its hot loop (ISA listing) x the ARCHITECTURAL issue cost of each opcode
    (tools/issue_model.py -> profiles/issue_model.json: 4 cycles for a full-rate wave64 instruction on a 16-lane SIMD, 8 / 16 for the
    fp32 / fp64 transcendentals, 2 for the simple 32-bit ops tools/ubench measured at 2.3), cross-checked against the hardware's
    instruction counters.  Priced at the chip's peak 2.4 GHz (issue_frac: a time no launch can beat at any clock) and at the
    clock the card actually held while this kernel ran back to back (issue_frac_at_measured_clock: what the kernel leaves on the
    table at the clock DVFS gives it).  Returns (fields for the roofline dict, or None when the committed model does not
    describe this build)."""
    try:
        model = json.load(open(os.path.join(ROOT, "profiles", "issue_model.json"))).get(workload, {})
    except (OSError, ValueError):
        model = {}
    if not model:
        return None
    if not (kernel_s and not traffic_stale and model.get("launch_stamp") == stamp["stamp"] and
            (model.get("cross_check_ok") or model.get("rescaled_to_counters"))):
        return {"issue_model_withheld": "profiles/issue_model.json does not describe this build (stamp) or failed its cross-check against the "
                                        "hardware counters: re-run tools/collect_pmc_all.sh and tools/issue_model.py"}
    waves_trips = paths / 64.0
    ceil_s = model["min_cycles_per_path"] * waves_trips / model["simds"] / model["clock_hz"]
    typ_s = model["typical_cycles_per_path"] * waves_trips / model["simds"] / model["clock_hz"]
    out = {"issue_frac": ceil_s / kernel_s,
           "issue_model": {
               "ceiling_us": ceil_s * 1e6, "kernel_us": kernel_s * 1e6, "frac_effective": (ceil_s / step_s) if step_s else None,
               "typical_us": typ_s * 1e6, "typical_frac": typ_s / kernel_s,
               "valu_insts_per_path": model["valu_per_path"], "min_cycles_per_path": model["min_cycles_per_path"],
               "simds": model["simds"], "clock_hz": model["clock_hz"], "source": model.get("source"),
               "cross_check": {"pmc_vs_histogram_valu": model.get("pmc_vs_model"), "ok_within_1pct": bool(model.get("cross_check_ok")),
                               "rescaled_to_counters": bool(model.get("rescaled_to_counters"))},
               "valu_busy_long_launch": (committed or {}).get("valu_busy_long_launch"), "valu_busy_source": (committed or {}).get("valu_busy_source"),
               "note": "ceiling = sum over the hot loop's VALU instructions of the ARCHITECTURAL issue cost of the opcode x wave-trips / "
                       "(1024 SIMDs x 2.4 GHz): a lower bound on the launch time at any clock; typical_us prices the same histogram at "
                       "the costs tools/ubench measured in mixed streams (4.1 / 8.1 / 16.2 cycles) -- an estimate, not a bound"}}
    if sclk_mhz:
        out["sclk_mhz"] = sclk_mhz
        out["issue_frac_at_measured_clock"] = out["issue_frac"] * model["clock_hz"] / (sclk_mhz * 1e6)
        out["issue_model"]["ceiling_us_at_measured_clock"] = ceil_s * 1e6 * model["clock_hz"] / (sclk_mhz * 1e6)
    return out


def cva_analytic(c):
    """E[CVA] of the reference's estimator in closed form: under the risk-neutral measure E[C(S_t, T - t)] = C_0 e^{r t}, so
    CVA = LGD * C_0 * sum_j dPD_j e^{r t_j} over the dates with a non-negative residual maturity."""
    from statistics import NormalDist
    s0, k, r, v, t = c["s"], c["k"], c["r"], c["v"], c["t"]
    d1 = (math.log(s0 / k) + (r + 0.5 * v * v) * t) / (v * math.sqrt(t))
    c0 = s0 * NormalDist().cdf(d1) - k * math.exp(-r * t) * NormalDist().cdf(d1 - v * math.sqrt(t))
    n, lam = c["n_grid"], c["defint"]
    dt, ttm, tot = t / n, t, 0.0
    for j in range(1, n + 1):
        ttm -= dt
        if ttm < 0:
            break
        tot += (math.exp(-lam * dt * (j - 1)) - math.exp(-lam * dt * j)) * math.exp(r * dt * j)
    return c["lgd"] * c0 * tot


def configs_block(mc, torch, engines, launch_streams, pci, headline, strong, stamp, cpu_seconds=2.0):
    """BASELINE.json configs[1..4], each on this ONE GPU with its roofline fractions and the reference's CPU path beside it:
      C2  vanilla, 1e8 paths fp32 (the headline: copied from it) + what ONE synchronous call of it costs (single_call: the reference
          driver times exactly that, dp/vanillaOpt.cu:77-83 -- a lone launch plus the call's fixed cost, without the second stream
          that hides ramp and tail in the stepped region)
      C3  basket, 4 assets, 1e8 paths fp32: stepped over the two context/stream pairs like the headline; price against the SAME
          basket in fp64 (1e9 paths)
      C4  basket, 16 assets, 1e9 paths fp64 on one GPU: T(1) of the strong block; price against the 1e10-path run (C4x10)
      C5  CVA, 256 dates x 1e7 paths fp64: T(1) of the strong block; price against the closed form (cva_analytic)
    kernel_us = the config's kernel alone (HIP events bound to the dispatch, launches one at a time); frac = algorithmic flop
    / kernel_us / the vector peak of the dtype; issue_frac(_at_measured_clock): issue_ceiling().  cpu = the compiled reference's
    own host_basketOpt / host_cvaEquityOption (oracle/_ref, gcc -O2, ONE thread: the reference is single-threaded) on a bounded
    sample; for the baskets the sp object -- the dp object's multiStockValue omits the volatility, its time is
    reported beside it as cpu_dp."""
    import numpy as np
    W = workloads(mc)
    seed = mc.MC_DEFAULT_SEED
    rows = {(r["config"]): r for r in (strong or {}).get("rows", [])}
    out = {}

    def lone_kernel(eng, prod, X, struct, count, n, stream_handle):
        eng.profile(1)
        scratch = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        for i in range(n):
            eng.launch(prod, X, struct, seed, (1 << 52) + i * count, count, scratch[i].data_ptr(), stream_handle)
        torch.cuda.synchronize()
        k, ms = eng.profile_read()
        eng.profile(0)
        return (ms / k * 1e-3) if k else None

    def entry(name, wl, paths_per_s, kernel_s, count, price, ci, target, versus, sclk):
        prod, X, _, _, flop, desc = W[wl]
        e = {"workload": desc, "paths": count, "paths_per_s": paths_per_s, "kernel_us": kernel_s * 1e6 if kernel_s else None,
             "frac": (flop * count / kernel_s / 1e12 / PEAK_TFLOPS[X]) if kernel_s else None, "dtype": X,
             "price": price, "confidence_95": ci, "err": abs(price - target) if target is not None else None, "vs": versus}
        im = issue_ceiling(wl, count, kernel_s, None, stamp, False, None, sclk) or {}
        for k in ("issue_frac", "issue_frac_at_measured_clock"):
            if k in im:
                e[k] = im[k]
        if im.get("issue_model"):
            e["ceiling_us"] = im["issue_model"]["ceiling_us"]
        if sclk:
            e["sclk_mhz"] = sclk
        return e

    # ---- C2: the headline's own numbers + one synchronous call ----
    r = headline["roofline"]
    c2 = {"workload": headline["config"]["workload"], "paths": headline["config"]["paths_per_gpu_per_step"], "paths_per_s": headline["value"],
          "kernel_us": r.get("avg_kernel_us"), "frac": r.get("frac"), "dtype": "f32", "price": headline["price"],
          "confidence_95": headline["confidence_95"], "err": headline.get("price_error_vs_black_scholes"), "vs": "Black-Scholes"}
    for k in ("issue_frac", "issue_frac_at_measured_clock", "sclk_mhz"):
        if r.get(k) is not None:
            c2[k] = r[k]
    eng = engines[0]
    eng.set_timing(False)                      # what the legacy symbols run with: the last workgroup writes the pinned slot, the host polls
    walls = []
    for i in range(-5, 30):
        t0 = time.perf_counter()
        est = eng.vanilla(VAN, 10 ** 8, seed, 0, "f32")     # the legacy symbol's range: paths [0, n) on every call
        if i >= 0:
            walls.append(time.perf_counter() - t0)
    eng.set_timing(True)
    walls.sort()
    c2["single_call"] = {"wall_us": walls[len(walls) // 2] * 1e6, "paths_per_s": 10 ** 8 / walls[len(walls) // 2], "calls": len(walls),
                         "what": "ONE synchronous mc_vanilla_run_f32 of 1e8 paths, back to back, median (dev_vanillaOpt's path, timing off)"}
    out["C2"] = c2

    # ---- C3: stepped like the headline ----
    prod, X, inputs, count, flop, desc = W["basket4_f32"]
    inputs = inputs()
    structs = [e_.prepared(prod, X, inputs) for e_ in engines]
    launch3 = lambda i, e_, st: e_.launch(prod, X, structs[engines.index(e_)][0], seed, (1 << 51) + i * count, count, warm[i % 2].data_ptr(), st)   # noqa: E731
    warm = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
    sclk3 = sustained_clock(torch, engines, launch_streams, launch3, pci, 150.0)
    K3, R3 = 40, 3
    tri = torch.zeros((R3 * K3, 3), dtype=torch.float64, device="cuda")
    regs = []
    for r_ in range(R3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(r_ * K3, (r_ + 1) * K3):
            e_ = i % len(engines)
            engines[e_].launch(prod, X, structs[e_][0], seed, i * count, count, tri[i].data_ptr(), launch_streams[e_])
        torch.cuda.synchronize()
        regs.append(time.perf_counter() - t0)
    regs.sort()
    tot = tri.sum(dim=0).cpu().tolist()
    disc32 = math.exp(-float(np.float32(inputs["r"])) * float(np.float32(inputs["t"])))
    p3, ci3 = mc.closing(tot[0], tot[1], int(tot[2]), disc32)
    k3 = lone_kernel(engines[0], prod, X, structs[0][0], count, 20, launch_streams[0])
    ref64 = engines[0].basket(basket_inputs(mc, 4, "f64"), 10 ** 9, seed, 0, "f64")
    out["C3"] = entry("C3", "basket4_f32", count * K3 / regs[len(regs) // 2], k3, count, p3, ci3, ref64.expected, "the same basket in fp64, 1e9 paths", sclk3)
    out["C3"]["vs_confidence_95"] = ref64.confidence
    out["C3"]["steps"] = [K3, R3]

    # ---- C4, C5: T(1) from the strong block, the kernel alone here ----
    for name, wl, total, n_lone in (("C4", "basket16_f64", 10 ** 9, 3), ("C5", "cva256_f64", 10 ** 7, 8)):
        prod, X, inputs, _, flop, desc = W[wl]
        if callable(inputs):
            inputs = inputs()
        struct, keep = engines[0].prepared(prod, X, inputs)
        launchN = lambda i, e_, st: e_.launch(prod, X, struct, seed, (1 << 51) + i * total, total, warm[0].data_ptr(), st)   # noqa: E731
        sclk = sustained_clock(torch, engines[:1], launch_streams[:1], launchN, pci, 200.0)
        kN = lone_kernel(engines[0], prod, X, struct, total, n_lone, launch_streams[0])
        row = rows.get(name)
        if row:
            t1, price, ci = row["wall_ms_median"] * 1e-3, row["value"], row["confidence_95"]
        else:       # the strong block was skipped: one timed call here
            t0 = time.perf_counter()
            est = getattr(engines[0], prod)(inputs, total, seed, 0, X)
            t1, price, ci = time.perf_counter() - t0, est.expected, est.confidence
        if name == "C4":
            big = rows.get("C4x10")
            target, versus = (big["value"], "the same basket, 1e10 paths (C4x10)") if big else (None, None)
        else:
            target, versus = cva_analytic(inputs), "closed form LGD C0 sum dPD_j e^(r t_j)"
        out[name] = entry(name, wl, total / t1, kN, total, price, ci, target, versus, sclk)
        out[name]["workload"] = desc.replace("1e9/8 paths per GPU", "1e9 paths, one GPU").replace("1e7/8 paths per GPU", "1e7 paths, one GPU")
        if name == "C4" and rows.get("C4x10"):
            out[name]["vs_confidence_95"] = rows["C4x10"]["confidence_95"]

    # ---- the reference's CPU path beside each (bounded samples; rank 0, N = 1 only) ----
    if cpu_seconds > 0:
        if headline.get("cpu_baseline"):
            out["C2"]["cpu"] = _pick(headline["cpu_baseline"], "value", "unit", "cores", "kind", "sample_short")
        b4, b16 = basket_inputs(mc, 4, "f32"), basket_inputs(mc, 16, "f32")
        for name, prodc, Xc, inp, also_dp in (("C3", "basket", "f32", b4, False), ("C4", "basket", "f32", b16, True), ("C5", "cva", "f64", CVA, False)):
            c = cpu_baseline(prodc, Xc, inp, cpu_seconds)
            e = _pick(c, "value", "unit", "cores", "kind", "sample_short")
            e["object"] = f"oracle/_ref/libref_{Xc}_n{len(inp['s']) if prodc == 'basket' else 3}.so" if c["kind"] == "reference" else "oracle port"
            if prodc == "basket":
                e["note"] = "sp object: the dp host's basket omits v[i] (SURVEY 2.3 #1)"
            if also_dp:
                c2_ = cpu_baseline(prodc, "f64", basket_inputs(mc, 16, "f64"), max(1.0, cpu_seconds / 2))
                e["cpu_dp"] = c2_["value"]
            out[name]["cpu"] = e
    return out


def cpu_all_cores(seconds=2.0):
    """The product's own CPU twin (libmchost: OpenMP, same Philox stream and estimator as the GPU) on all
    host cores: the fair many-core figure next to the single-threaded reference."""
    import ctypes as C
    path = os.path.join(ROOT, "montecarlocuda_amd", "csrc", "libmchost_f32.so")
    if not os.path.exists(path):
        return None

    class OptionData(C.Structure):
        _fields_ = [(k, C.c_float) for k in "skrvt"]

    class OptionValue(C.Structure):
        _fields_ = [("Expected", C.c_float), ("Confidence", C.c_float)]
    L = C.CDLL(path)
    L.host_vanillaOpt.argtypes = [OptionData, C.c_int]
    L.host_vanillaOpt.restype = OptionValue
    o = OptionData(*[VAN[k] for k in "skrvt"])
    n = 50_000_000
    t0 = time.perf_counter(); L.host_vanillaOpt(o, n); dt = time.perf_counter() - t0
    n = int(min(2 ** 31 - 1, max(n, n * seconds / max(dt, 1e-3))))     # the reference's API counts paths in an int
    calls, t0 = 0, time.perf_counter()
    while calls == 0 or time.perf_counter() - t0 < seconds:
        v = L.host_vanillaOpt(o, n)
        calls += 1
    dt = time.perf_counter() - t0
    L.mc_host_threads.restype = C.c_int
    return {"value": calls * n / dt, "unit": "paths/s", "cores": int(L.mc_host_threads()), "hardware_threads_visible": len(os.sched_getaffinity(0)),
            "kind": "libmchost_f32 (OpenMP CPU twin of the engine)", "sample": f"{calls} calls of {n} paths in {dt:.2f} s", "price": float(v.Expected),
            "note": "threads = OpenMP's default capped by the container's cgroup CPU quota (MC_HOST_THREADS overrides)"}


def strong_scaling_block(mc, torch, dist, eng, stream, rank, world, grouped, backend, barrier, reps, preheat_ms=300.0, full=False):
    """BASELINE.json's strong-scaling target, measured on this N: ONE pricing call of configs[3] (C4: basket, 16 assets,
    1e9 paths, fp64) and of configs[4] (C5: CVA, 256 dates x 1e7 paths, fp64) -- and of 10x those sizes, and of
    all four once more on fp32 normals (the reference's own dp arithmetic: shorter kernels, fixed costs weigh more) --
    sharded over the N ranks (mc_shard_range), timed wall-clock from the first launch to the all-reduced triple on the
    host, max over ranks, median of `reps` (>= 10) calls.  The driver's lines for N = 1, 2, 4, 8 give the
    efficiency T1 / (N TN) of each row.

    Clock conditioning (profiles/r04_shard_clock_quantisation_vs_dvfs.log): an idle MI355X runs its first ~30 ms of work at
    1.9-2.2 GHz; C5's 1 ms shard measured after two warm-up calls took 1.15-1.19 ms, after 300 ms of load 0.97-0.98 ms.  So
    every row -- T(1) and T(shard) alike -- is measured HOT: this rank's own shard launched back to back for `preheat_ms`
    (no collective), two warm-up calls, then the timed calls.  The base-size rows are also measured COLD (0.5 s of idle, two
    warm-up calls, 5 calls: what a one-off call sees) and reported beside it ("cold")."""
    import numpy as np
    rows = []
    c4, c4d = basket_inputs(mc, 16, "f64"), "Basket call, 16 correlated assets, 1e9 paths, fp64 (BASELINE configs[3])"
    c5d = "CVA on vanilla call, 256 dates x 1e7 paths, fp64 (BASELINE configs[4])"
    n32 = " on fp32 normals (the reference's dp arithmetic)"
    # name, product, inputs, total paths, description, normals mode, base size (measured cold as well)
    specs = [("C4", "basket", c4, 10 ** 9, c4d, "native", True), ("C4x10", "basket", c4, 10 ** 10, "the same, 1e10 paths", "native", False),
             ("C5", "cva", CVA, 10 ** 7, c5d, "native", True), ("C5x10", "cva", CVA, 10 ** 8, "the same, 1e8 paths", "native", False),
             ("C4_n32", "basket", c4, 10 ** 9, c4d + n32, "f32", True), ("C4x10_n32", "basket", c4, 10 ** 10, "the same, 1e10 paths", "f32", False),
             ("C5_n32", "cva", CVA, 10 ** 7, c5d + n32, "f32", True), ("C5x10_n32", "cva", CVA, 10 ** 8, "the same, 1e8 paths", "f32", False)]
    if not full:     # --detail brief: the x10 sizes on fp32 normals only under --detail full
        specs = [sp for sp in specs if sp[0] not in ("C4x10_n32", "C5x10_n32")]
    out = torch.zeros(3, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(3, dtype=torch.float64, device="cuda")
    pinned = torch.zeros(3, dtype=torch.float64).pin_memory()
    # N = 1 also times what ONE rank does at N = 2, 4, 8: shard 0 of S of every row through the same code path
    # ("shard_of" rows) -- the device side of the scaling curve, which a one-GPU box can measure; the all-reduce
    # between ranks is the only part missing from them
    shard_rows = []
    runs = [(spec, 1) for spec in specs]
    if world == 1:
        runs += [(spec, S) for S in ((2, 4, 8) if full else (8,)) for spec in specs]
    t_full, t_full_cold = {}, {}

    def one_call(prod, struct, first, count):
        """first launch -> the all-reduced triple on this host: returns the triple as a 3-element float64 tensor"""
        # launch, RCCL and the read-back all on torch's current stream (a handle torch owns): no hop between streams.
        # The result comes back the way libmc_multi reads its devices: the last workgroup (N = 1) or a one-lane kernel
        # behind the all-reduce (RCCL) stores the triple into pinned host memory and this thread polls it from user
        # space -- no copy command, no sleeping synchronize (mc_context_arm_direct / mc_context_publish).
        direct = count > 0 and (not grouped or backend == "nccl")
        slot = eng.arm_direct() if (direct and not grouped) else None
        if count:
            eng.launch(prod, "f64", struct, mc.MC_DEFAULT_SEED, first, count, out.data_ptr(), stream.cuda_stream)
        else:
            out.zero_()
        if grouped and backend == "nccl":
            dist.all_reduce(out, op=dist.ReduceOp.SUM)      # RCCL, ordered behind the launch (current stream)
            if direct:
                slot = eng.publish(out.data_ptr(), stream.cuda_stream)
        if slot is not None:
            host = torch.tensor(eng.wait_slot(slot), dtype=torch.float64)
        else:
            pinned.copy_(out, non_blocking=True)            # 24 bytes into pinned host memory
            stream.synchronize()
            host = pinned.clone()
        if grouped and backend != "nccl":
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
        return host

    def timed(prod, struct, first, count, n_reps, hot):
        host = None
        if hot and preheat_ms > 0 and count:
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < preheat_ms:     # this rank's own shard, back to back, no collective
                for _ in range(2):
                    eng.launch(prod, "f64", struct, mc.MC_DEFAULT_SEED, (1 << 50) + first, count, scratch.data_ptr(), stream.cuda_stream)
                stream.synchronize()
        elif not hot:
            barrier()
            time.sleep(0.5)
        times = []
        for r_ in range(-2, n_reps):
            barrier()
            t0 = time.perf_counter()
            host = one_call(prod, struct, first, count)
            if r_ >= 0:
                times.append(time.perf_counter() - t0)
        t = torch.tensor(times, dtype=torch.float64, device="cuda" if (grouped and backend == "nccl") else "cpu")
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)            # a repeat ends when its slowest rank has the result
        return sorted(t.tolist()), host

    def timed_local(prod, struct, first, count, n_reps):
        """this rank's launch alone, first launch -> its own triple on this host (pinned slot), no collective: median seconds"""
        ts = []
        for r_ in range(-2, n_reps):
            t0 = time.perf_counter()
            slot = eng.arm_direct()
            eng.launch(prod, "f64", struct, mc.MC_DEFAULT_SEED, first, count, scratch.data_ptr(), stream.cuda_stream)
            eng.wait_slot(slot)
            if r_ >= 0:
                ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def gather_floats(x):
        where = "cuda" if backend == "nccl" else "cpu"
        mine = torch.tensor([x], dtype=torch.float64, device=where)
        every = [torch.zeros(1, dtype=torch.float64, device=where) for _ in range(world)]
        dist.all_gather(every, mine)
        return [float(t.item()) for t in every]

    for (name, prod, inputs, total, desc, normals, base), shard_of in runs:
        eng.set_normals(normals)
        struct, keep = eng.prepared(prod, "f64", inputs)
        first, count = mc.shard_range(total, rank, world) if shard_of == 1 else mc.shard_range(total, 0, shard_of)
        times, host = timed(prod, struct, first, count, reps, True)
        decomposition = None
        if world > 1 and shard_of == 1 and base and count:
            # what the N-rank wall time is made of (the driver's efficiency T1 / (N TN) needs its own N = 1 line; this line explains
            # itself): every rank's own shard WITHOUT the collective (device side), and T(1) of the whole config on rank 0 alone
            shard_s = gather_floats(timed_local(prod, struct, first, count, reps))
            barrier()
            t1_s = timed_local(prod, struct, 0, total, min(reps, 5)) if rank == 0 else 0.0
            barrier()
            decomposition = (shard_s, t1_s)
        cold = None
        if base:
            cold_times, _ = timed(prod, struct, first, count, min(reps, 5), False)
            cold = float(np.median(cold_times))
        s_, s2_, n_ = (float(x) for x in host.tolist())
        disc = 1.0 if prod == "cva" else math.exp(-float(inputs["r"]) * float(inputs["t"]))
        price, ci = mc.closing(s_, s2_, int(n_), disc)
        med = float(np.median(times))
        if shard_of == 1:
            t_full[name], t_full_cold[name] = med, cold
            row = {"config": name, "workload": desc, "normals": normals, "paths_total": total, "paths_per_gpu": count, "reps": reps,
                   "preheat_ms": preheat_ms, "wall_ms_median": med * 1e3, "wall_ms_min": times[0] * 1e3, "paths_per_s": total / med,
                   "value": price, "confidence_95": ci, "paths_priced": int(n_)}
            if cold is not None:
                row["cold"] = {"wall_ms_median": cold * 1e3, "what": "0.5 s idle, 2 warm-up calls, median of 5"}
            if decomposition:
                shard_s, t1_s = decomposition
                row["t_shard_ms"] = [min(shard_s) * 1e3, max(shard_s) * 1e3]
                row["t_shard_ms_by_rank"] = [x * 1e3 for x in shard_s]
                row["collective_ms"] = (med - max(shard_s)) * 1e3
                if rank == 0 and t1_s > 0:
                    row["t1_ms_rank0"] = t1_s * 1e3
                    row["eff"] = t1_s / (world * med)
                    row["eff_device_side"] = t1_s / (world * max(shard_s))
                row["decomposition"] = ("t_shard_ms = [min, max] over ranks of each rank's own shard, launch -> its triple on its host, no collective; "
                                        "collective_ms = wall_ms_median - the slowest shard: what the all-reduce and its hand-overs add; t1_ms_rank0 = the "
                                        "whole config on rank 0 alone, same read-back; eff = t1 / (N wall), eff_device_side = t1 / (N slowest shard): the "
                                        "difference is the collective's share")
            rows.append(row)
        else:
            row = {"config": name, "normals": normals, "shard_of": shard_of, "paths": count, "reps": reps, "preheat_ms": preheat_ms,
                   "wall_ms_median": med * 1e3, "wall_ms_min": times[0] * 1e3, "device_side_efficiency": t_full[name] / (shard_of * med),
                   "what": f"shard 0 of {shard_of} on this one GPU: T(1) / ({shard_of} T(shard)), hot / hot; the all-reduce between "
                           "ranks is not in it"}
            if cold is not None and t_full_cold.get(name):
                row["cold"] = {"wall_ms_median": cold * 1e3, "device_side_efficiency": t_full_cold[name] / (shard_of * cold)}
            shard_rows.append(row)
    eng.set_normals("native")
    allreduce_us = None
    if grouped:
        # the collective alone: 24 bytes, what closes every strong call (pure latency).  Outside every timed region, after a
        # barrier; each call timed on the host from issue to completion (RCCL: to the end of the stream's work)
        t24 = torch.zeros(3, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        lat = []
        barrier()
        for i in range(-20, 200):
            t0 = time.perf_counter()
            dist.all_reduce(t24, op=dist.ReduceOp.SUM)
            if backend == "nccl":
                torch.cuda.current_stream().synchronize()
            if i >= 0:
                lat.append(time.perf_counter() - t0)
        lat.sort()
        allreduce_us = {"median": lat[len(lat) // 2] * 1e6, "p10": lat[len(lat) // 10] * 1e6, "p90": lat[len(lat) * 9 // 10] * 1e6, "calls": len(lat),
                        "backend": backend, "what": "24-byte dist.all_reduce, issue -> complete on this rank's host (rank 0), back to back after a barrier"}
    return {"scaling": "strong", "n_gpus": world, "rows": rows, "shard_rows": shard_rows, "allreduce_us": allreduce_us,
            "timing": "wall-clock, first launch -> all-reduced {sum, sum2, n} on the host, max over ranks; every row HOT: "
                      f"{preheat_ms:.0f} ms of this rank's own shard back to back, 2 warm-up calls, median of `reps`; base sizes also "
                      "COLD (0.5 s idle first)",
            "note": "strong-scaling efficiency of a row = wall_ms_median(N=1) / (N * wall_ms_median(N)), from the driver's "
                    "own N = 1, 2, 4, 8 lines"}


def c_multi_block(max_seconds, full=False):
    """The same strong-scaling rows through the C library alone: drivers/multiBench (plain C, libmc_multi.so: ONE
    process drives G = 1, 2, 4, 8 ... of the visible GPUs, shards + one direct RCCL all-reduce).  Run as a child
    process with a time limit so that nothing it does can disturb the headline measurement above."""
    import subprocess
    exe = os.path.join(ROOT, "drivers", "multiBench")
    if not os.path.exists(exe):
        return {"error": "drivers/multiBench not built (make -C drivers)"}
    try:
        cmd = [exe, "--reps", "10"] + ([] if full else ["--brief"])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=max_seconds)
    except subprocess.TimeoutExpired:
        return {"error": f"drivers/multiBench exceeded {max_seconds} s"}
    rows = []
    for line in out.stdout.splitlines():
        try:
            rows.append(json.loads(line))
        except ValueError:
            pass
    res = {"command": "drivers/" + " ".join(["multiBench"] + cmd[1:]), "rc": out.returncode, "rows": rows,
           "what": "one C process, libmc_multi.so: mc_shard_range + mc_*_launch_* per device + ONE ncclAllReduce(3, ncclDouble); "
                   "wall-clock first launch -> closed estimate"}
    if out.returncode != 0:
        res["error"] = (out.stderr or out.stdout)[-400:]
    return res


LINE_LIMIT = 4096   # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout: a longer line arrives headless)


def _sig(x, n=6):
    """Floats to n significant digits, recursively (the compact line only; the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if d and k in d and d[k] is not None}


SHARD8_COLS = ["t1_ms", "t_shard8_ms", "shard8_device_side_eff_1gpu", "shard8_device_side_eff_1gpu_cold"]


def strong_summary(detail):
    """A few numbers per strong-scaling config out of the full record's rows.
    N = 1: {"cols": SHARD8_COLS, "src": [bench.py's torch path, the C library (drivers/multiBench, libmc_multi.so)], config: [[a, b] per
    column]}: t1_ms = one call of the whole config on this GPU (wall, hot, median); t_shard8_ms = shard 0 of 8 of it ON THIS ONE GPU;
    shard8_device_side_eff_1gpu(_cold) = T(1) / (8 T(shard)) hot (cold) -- a ratio of two single-GPU timings: the device side of the
    8-GPU point, everything but the collective between ranks; NOT a measured 8-GPU efficiency (round 5 called it eff8).
    `c_devices` (when the C library drove G > 1 of the visible devices, real RCCL): {config: {G: [wall ms, efficiency vs its own
    G = 1, fan-out us, collective us]}}.
    N > 1: per config wall_ms (one call sharded over the N ranks, first launch -> all-reduced triple, max over ranks), the ranks'
    own shard times t_shard_ms [min, max], collective_ms, t1_ms_rank0, eff and eff_device_side (strong_scaling_block), and
    allreduce_us (the 24-byte collective alone)."""
    strong = detail.get("strong") or {}
    rows = {r["config"]: r for r in strong.get("rows", [])}
    if not rows:
        return None
    if detail.get("n_gpus", 1) > 1:
        out = {c: _pick(r, "wall_ms_median", "paths_per_gpu", "value", "t_shard_ms", "collective_ms", "t1_ms_rank0", "eff", "eff_device_side")
               for c, r in rows.items()}
        if strong.get("allreduce_us"):
            out["allreduce_us"] = _pick(strong["allreduce_us"], "median", "p10", "p90", "calls")
        return out
    shard = {(r["config"], r["shard_of"]): r for r in strong.get("shard_rows", [])}
    crows_all = (detail.get("c_multi") or {}).get("rows", [])
    crows = {r.get("config"): r for r in crows_all if r.get("devices") == 1 and "shard_of" not in r and "config" in r}
    cshard = {(r.get("config"), r["shard_of"]): r for r in crows_all if "shard_of" in r}
    out = {"cols": SHARD8_COLS, "src": ["bench.py", "libmc_multi"]}
    multi_all = {}
    for c, r in rows.items():
        s8, c1, c8 = shard.get((c, 8)), crows.get(c), cshard.get((c, 8))
        cols = [[r["wall_ms_median"], c1["wall_ms_median"] if c1 else None]]
        if s8 or c8:
            cols.append([s8["wall_ms_median"] if s8 else None, c8["wall_ms_median"] if c8 else None])
            cols.append([s8["device_side_efficiency"] if s8 else None, c8["device_side_efficiency"] if c8 else None])
            cold = [(x or {}).get("cold", {}).get("device_side_efficiency") for x in (s8, c8)]
            if any(v is not None for v in cold):
                cols.append(cold)
        out[c] = cols
        multi = {str(x["devices"]): [x["wall_ms_median"], x.get("strong_efficiency_vs_1"), x.get("fanout_us"), x.get("collective_us")]
                 for x in crows_all if x.get("config") == c and x.get("devices", 1) > 1 and "shard_of" not in x}
        if multi:
            multi_all[c] = multi
    if multi_all:
        out["c_devices"] = multi_all
    return out


def configs_summary(detail):
    """BASELINE configs[1..4] in the compact line: per config paths_per_s, kernel_us (the kernel alone), frac (flop), issue_frac (ceiling
    at 2.4 GHz / kernel), issue_frac_clk (= issue_frac_at_measured_clock, the ceiling at sclk_mhz), price, err against `vs`, and
    cpu_baseline = the reference's CPU path timed on this host (paths/s, cores, kind); C2 also single_call = [wall us, paths/s] of ONE
    synchronous call.  4 significant digits except the rates."""
    cf = detail.get("configs")
    if not cf:
        return None
    short = {"Black-Scholes": "BS", "the same basket in fp64, 1e9 paths": "fp64 1e9 paths", "the same basket, 1e10 paths (C4x10)": "C4x10",
             "closed form LGD C0 sum dPD_j e^(r t_j)": "closed form"}
    out = {}
    for name, e in cf.items():
        o = _pick(e, "paths_per_s", "kernel_us", "frac", "issue_frac")
        if e.get("issue_frac_at_measured_clock") is not None:
            o["issue_frac_clk"] = e["issue_frac_at_measured_clock"]
        o.update(_pick(e, "sclk_mhz", "err"))
        if e.get("vs"):
            o["vs"] = short.get(e["vs"], e["vs"])
        o = _sig(o, 4)
        if e.get("price") is not None:
            o["price"] = float(f"{e['price']:.7g}")
        if e.get("paths_per_s"):
            o["paths_per_s"] = float(f"{e['paths_per_s']:.5g}")
        if e.get("cpu"):
            o["cpu_baseline"] = {"value": float(f"{e['cpu'].get('value', 0):.4g}"), "cores": e["cpu"].get("cores"), "kind": e["cpu"].get("kind")}
        if e.get("single_call"):
            o["single_call"] = [float(f"{e['single_call']['wall_us']:.4g}"), float(f"{e['single_call']['paths_per_s']:.4g}")]
        out[name] = o
    return out


def compact_line(detail, detail_name="bench_detail.json"):
    """The ONE stdout line: contract fields + config + roofline + cpu_baseline + fp64 + strong_summary (+ roster), floats at
    6 significant digits (value / ms_per_step / timed_region_s in full), at most LINE_LIMIT bytes.  Pure function of the
    full record, so that a CPU test can bound its size from a recorded fixture (tests/test_bench_cli.py)."""
    d = detail
    line = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                              "vs_baseline", "dtype", "data") if k in d}
    cfg = d.get("config", {})
    line["config"] = _pick(cfg, "workload", "paths_per_gpu_per_step", "global_paths_per_step", "parallelism", "rng", "seed", "grid",
                           "streams", "finish", "preheat_ms", "engine_settings", "detail")
    if "parallelism" in line["config"]:
        line["config"]["parallelism"] = str(line["config"]["parallelism"]).split(",")[0]      # "path-sharded xN" (the rest: detail file)
    if "rng" in line["config"]:
        line["config"]["rng"] = "Philox4x32-10, counter = path index"
    line.update(_pick(d, "timed_region_s", "regions"))
    body = _pick(d, "ms_per_step_min", "ms_per_step_max", "price", "confidence_95", "paths_priced", "price_error_vs_black_scholes")
    r = d.get("roofline") or {}
    roof = _pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "kernel", "avg_kernel_us",
                 "issue_frac", "issue_frac_at_measured_clock", "sclk_mhz", "hbm_gbps", "grid_workgroups")
    roof.setdefault("traffic", None)
    if r.get("in_region"):
        roof["in_region"] = _pick(r["in_region"], "avg_kernel_us", "concurrent_launches", "step_period_us")
    if r.get("effective"):
        roof["effective"] = _pick(r["effective"], "achieved", "frac")
    if r.get("issue_model"):
        roof["issue_model"] = _pick(r["issue_model"], "ceiling_us", "frac_effective", "valu_insts_per_path", "valu_busy_long_launch")
    elif r.get("issue_model_withheld"):
        roof["issue_model"] = "withheld: committed counters do not describe this build"
    body["roofline"] = roof
    if d.get("cpu_baseline"):
        c = d["cpu_baseline"]
        body["cpu_baseline"] = _pick(c, "value", "unit", "cores", "kind", "host_cpus_granted", "value_at_O0")
        body["cpu_baseline"]["sample"] = c.get("sample_short") or str(c.get("sample", ""))[:80]
    if d.get("fp64"):
        body["fp64"] = _pick(d["fp64"], "value", "unit", "steps", "ms_per_step", "price_error_vs_black_scholes")
    cs = configs_summary(d)
    if cs:
        body["configs"] = cs
    ss = strong_summary(d)
    if ss:
        body["strong_summary"] = ss
    body.update(_pick(d, "world_size", "backend", "rccl_version", "ranks", "devices_visible", "device"))
    line.update(_sig(body, 5))
    line["detail"] = detail_name
    # never above the limit: shed the least important parts first, and say so
    dropped = []

    def size():
        return len(json.dumps(dict(line, dropped_for_size=dropped) if dropped else line))
    for k in ("strong_summary.c_devices", "fp64", "device", "strong_summary", "ranks", "configs"):
        if size() <= LINE_LIMIT:
            break
        if k == "strong_summary.c_devices":
            if (line.get("strong_summary") or {}).pop("c_devices", None) is not None:
                dropped.append(k)
        elif line.pop(k, None) is not None:
            dropped.append(k)
    if dropped:
        line["dropped_for_size"] = dropped
    if len(json.dumps(line)) > LINE_LIMIT:
        raise RuntimeError(f"bench line is {len(json.dumps(line))} bytes (> {LINE_LIMIT}) even without {dropped}")
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--regions", type=int, default=5,
                    help="the K-step timed region is repeated this many times back to back (fresh path ranges each time, one "
                         "pre-heat before the first); ms_per_step and value are the MEDIAN region's, min and max are reported: a "
                         "20-step region is 1 ms, one sample of it moves by +-4 %% between runs")
    ap.add_argument("--workload", default="vanilla_f32")
    ap.add_argument("--paths", type=int, default=0, help="paths per GPU per step (default: the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0,
                    help="CPU baseline sample length of the headline workload (0 = skip every CPU baseline); the other configs' CPU paths "
                         "run for 2 s each")
    ap.add_argument("--configs", type=int, default=1,
                    help="1 (default): N = 1 also measures BASELINE configs[2..4] (C3 basket n=4 fp32, C4 basket n=16 fp64 1e9 paths, C5 CVA "
                         "256 x 1e7 fp64) with roofline fractions, price error and the reference's CPU path beside each; 0 = skip")
    ap.add_argument("--detail", default="brief", choices=["brief", "full"],
                    help="brief (default): strong rows C4, C5, their 10x sizes, C4 / C5 on fp32 normals, shard 0 of 8 -- the driver's run "
                         "stays under a minute.  full: also the 10x sizes on fp32 normals, shard 0 of 2 and of 4, the -O0 CPU build")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where rank 0 writes the full record (every row; the stdout line carries the summary and names this file)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU simulates `paths` per step. strong: `paths` per step in total, "
                         "rank g taking mc_shard_range(paths, g, N) of every step")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent (context, HIP stream) pairs the steps rotate over; >1 lets consecutive pricing "
                         "calls overlap each other's launch gaps, ramps and tails")
    ap.add_argument("--stream-source", default="context", choices=["torch", "context"],
                    help="where the launch streams come from: each context's own stream (mc_context_stream; default) or "
                         "torch's stream pool.  Two streams of torch's pool were seen to share ONE hardware queue on some "
                         "boxes (57.6 instead of 49.3 us/step: no overlap at all; GPU_MAX_HW_QUEUES=8 or a third stream "
                         "restores it), the contexts' own streams never did: profiles/r02_stream_queue_sweep.log")
    ap.add_argument("--profile-every", type=int, default=8)
    ap.add_argument("--blocks", type=int, default=0, help="workgroups per launch (0 = the engine's default, 8 per CU)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default). gloo = rehearsal of the multi-rank logic on a box with fewer "
                         "GPUs than ranks: ranks share GPUs (LOCAL_RANK mod device count), triples are reduced on the host")
    ap.add_argument("--fp64-steps", type=int, default=300, help="steps of the fp64 side measurement (0 = skip)")
    ap.add_argument("--exclusive-launches", type=int, default=50,
                    help="after the timed region, this many launches one at a time on one stream, each timed on the device: "
                         "the kernel's duration without a neighbour (roofline.exclusive); 0 = skip")
    ap.add_argument("--finish", default="fused", choices=["fused", "kernel"],
                    help="fused (default): the last workgroup of a call adds the per-workgroup pairs inside the simulation "
                         "kernel (one launch per call). kernel: a second one-workgroup launch does (A/B baseline)")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed device work before the warm-up steps, so that the W warm-up steps and the K timed steps "
                         "run at the GPU's sustained clock (a cold MI355X needs tens of ms of load to ramp; with --warmup 5 "
                         "the timed region would otherwise measure the ramp).  Reported as config.preheat_ms; 0 = off")
    ap.add_argument("--strong-reps", type=int, default=10,
                    help="timed calls of each strong-scaling row (C4, C5, 10x sizes, and the same on fp32 normals, sharded over the N "
                         "ranks; SURVEY 8d: median of >= 10; 0 = skip the block)")
    ap.add_argument("--strong-preheat-ms", type=float, default=300.0,
                    help="device load before every strong-scaling row (this rank's own shard, back to back), so that T(1) and "
                         "T(shard) are both measured at the sustained clock; 0 = off")
    ap.add_argument("--c-multi-seconds", type=int, default=150,
                    help="N=1 only: time limit of the child process drivers/multiBench (the C library's own multi-GPU path over "
                         "1, 2, 4, 8 ... of the visible GPUs); 0 = skip")
    ap.add_argument("--poll", type=int, default=0,
                    help="1: the host polls the launch streams (mc_context_idle) before the closing synchronize of the timed "
                         "region instead of sleeping in it; 0 (default): blocking synchronize only -- measured equal within noise "
                         "(profiles/r02_short_run_variance.log)")
    ap.add_argument("--collective", default="sync", choices=["sync", "async"],
                    help="N > 1, RCCL: the bucket all-reduce as a synchronous-mode collective on torch's current stream (default) or "
                         "async_op=True on the process group's internal stream")
    ap.add_argument("--closing", default="collective", choices=["collective", "barrier"],
                    help="N > 1 over RCCL: what closes a timed region.  collective (default): the all-reduce of the region's last "
                         "bucket of triples, which is a barrier (no rank's completes before every rank has entered it), then "
                         "torch.cuda.synchronize().  barrier: that all-reduce, then a separate dist.barrier(), then the synchronize")
    ap.add_argument("--bucket", type=int, default=25,
                    help="steps whose triples share one all-reduce (bucketed collective: 24 B x bucket); 1 = one per step")
    args = ap.parse_args()

    # the host driver of these boxes supports dmabuf IPC only: without this RCCL's communicator set-up between processes fails with
    # "hipIpcGetMemHandle: invalid argument" (it is exported on the boxes already; kept for any environment built by hand)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import montecarlocuda_amd as mc
    from montecarlocuda_amd import distributed as D

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    rank, world, local = D.init_from_env(args.backend)
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    grouped = dist.is_initialized()   # one process per GPU under torch.distributed.run (RCCL), also for N=1
    engines = [mc.Engine(local, args.blocks) for _ in range(max(1, args.streams))]
    settings = workload_settings(args.workload)
    for e_ in engines:
        e_.set_finish(args.finish == "fused")
        if "normals" in settings:
            e_.set_normals(settings["normals"])
        if settings.get("antithetic"):
            e_.set_antithetic(True)
        if settings.get("control_variate"):
            e_.set_control_variate(True)
    eng = engines[0]
    eng_info = eng.info()
    prod, X, inputs, paths, flop_per_path, desc = workloads(mc)[args.workload]
    if callable(inputs):
        inputs = inputs()
    if args.paths:
        paths = args.paths
    K, W, R = args.steps, args.warmup, max(1, args.regions)
    struct, keep = eng.prepared(prod, X, inputs)
    seed = mc.MC_DEFAULT_SEED
    # Launch streams: each context's own non-blocking stream (raw hipStream_t handles; torch never sees them) or
    # streams of torch's pool.  `stream` = torch's CURRENT stream, always one of torch's own: torch's copies and RCCL's
    # waits are ordered on it, and it is ordered behind the launch streams in flush_bucket (mc_context_order).
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)
    if args.stream_source == "context":
        launch_streams = [e.stream for e in engines]
    else:
        pool = [stream] + [torch.cuda.Stream(device=local) for _ in engines[1:]]
        launch_streams = [s_.cuda_stream for s_ in pool]
    assert all(h != 0 for h in launch_streams) and stream.cuda_stream != 0
    structs = [e.prepared(prod, X, inputs) for e in engines]
    triples = torch.zeros((W + R * K, 3), dtype=torch.float64, device="cuda")
    works = []

    pending = [0, 0]   # [first step not yet all-reduced, one past the last launched step]

    def flush_bucket():
        # one RCCL all-reduce for the triples of steps [pending[0], pending[1]): the rows are
        # contiguous, so a bucket is a single (bucket x 3) fp64 message, asynchronous to compute
        if grouped:
            for e_ in engines:
                e_.order(stream.cuda_stream)    # the bucket's triples come from every launch stream
        if grouped and pending[1] > pending[0]:
            rows = triples[pending[0]:pending[1]]
            if args.backend == "nccl":
                if args.collective == "sync":
                    # a synchronous-mode collective runs on torch's CURRENT stream (PyTorch >= 2.8), which executes no
                    # pricing kernel here (those are on the contexts' streams): still asynchronous to compute and to the
                    # host, and without the hand-overs to and from the process group's internal stream
                    dist.all_reduce(rows, op=dist.ReduceOp.SUM)
                else:
                    works.append(dist.all_reduce(rows, op=dist.ReduceOp.SUM, async_op=True))
            else:   # rehearsal path: reduce on the host
                host = rows.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                rows.copy_(host)
        pending[0] = pending[1]

    if args.scaling == "strong":
        shard_first, shard_count = mc.shard_range(paths, rank, world)   # this rank's slice of every step
        step_total = paths
    else:
        shard_first, shard_count = rank * paths, paths
        step_total = world * paths

    def step(i, last_of_region=False):
        first = i * step_total + shard_first
        e = i % len(engines)
        engines[e].launch(prod, X, structs[e][0], seed, first, shard_count, triples[i].data_ptr(), launch_streams[e])
        pending[1] = i + 1
        # a region's last step never flushes here: its bucket is flushed by the region's closing collective (below)
        if pending[1] - pending[0] >= max(1, args.bucket) and not last_of_region:
            flush_bucket()

    def drain(host_sync=True):
        flush_bucket()
        if not host_sync:
            # The caller's barrier() follows at once: its collective runs on RCCL's stream BEHIND the bucket's all-reduce
            # and the host waits for it, so the all-reduced triples are complete when it returns.  Making torch's stream
            # wait for the all-reduce first (Work.wait) would put two more stream-to-stream hand-overs (RCCL -> torch ->
            # RCCL, ~12 us each on this platform) into the dependency chain; the works are collected after the barrier.
            return
        for w in works:
            w.wait()          # stream-level: torch's stream waits for RCCL's
        works.clear()
        if args.poll:
            # poll the launch streams (and torch's) from user space until the work is done, THEN synchronise
            while not (all(e_.idle() for e_ in engines) and stream.query()):
                pass
        torch.cuda.synchronize()

    def barrier():
        if grouped:
            dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # N > 1 over RCCL: the barrier that closes a timed region IS the all-reduce of the region's last bucket of triples.
    # An all-reduce is a barrier -- no rank's copy completes before every rank has entered it, and a rank enters it only
    # behind its own last step (mc_context_order) -- so `all-reduce; torch.cuda.synchronize()` is the contract's
    # "barrier + synchronize" with one collective instead of two (dist.barrier() is itself a one-element all-reduce:
    # ~30 us more per region, 3 % of the driver's 20-step region).  --closing barrier restores the separate one.
    closing_is_collective = grouped and args.backend == "nccl" and args.closing == "collective" and args.collective == "sync"

    def close_region():
        if closing_is_collective:
            flush_bucket()               # >= 1 step pending: step() never flushes a region's last step
            torch.cuda.synchronize()     # launch streams, torch's stream (the all-reduce), everything
        else:
            drain(host_sync=not grouped)   # N > 1: RCCL's barrier queues right behind the last bucket's all-reduce
            barrier()

    preheat_ms = 0.0
    if args.preheat_ms > 0:
        # same kernel, same inputs, its own output slot; path ranges far above the timed steps'
        scratch = torch.zeros((len(engines), 3), dtype=torch.float64, device="cuda")
        t_pre = time.perf_counter()
        j = 0
        while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
            for _ in range(64):
                e = j % len(engines)
                engines[e].launch(prod, X, structs[e][0], seed, (1 << 50) + j * shard_count, shard_count, scratch[e].data_ptr(),
                                  launch_streams[e])
                j += 1
            torch.cuda.synchronize()
        preheat_ms = (time.perf_counter() - t_pre) * 1e3
    for i in range(W):
        step(i)
    drain()
    eng.profile(1 if R * K < 100 else args.profile_every)   # short runs: every launch of the first context is sampled
    # R timed regions of exactly K steps each, back to back: barrier + synchronize, K steps, drain, barrier + synchronize,
    # MAX over ranks -- the contract's bracket, repeated.  ms_per_step / value come from the MEDIAN region.
    region_s, host_sides = [], []
    for r_ in range(R):
        barrier()
        t0 = time.perf_counter()
        last = W + (r_ + 1) * K - 1
        for i in range(W + r_ * K, last + 1):
            step(i, last_of_region=(i == last))
        t_enqueued = time.perf_counter()
        close_region()
        t_drained = time.perf_counter()
        el = time.perf_counter() - t0
        for w in works:          # complete since the barrier (same RCCL stream, earlier in order): orders torch's stream, costs nothing
            w.wait()
        works.clear()
        host_sides.append({"enqueue_K_steps_ms": (t_enqueued - t0) * 1e3, "close_region_ms": (t_drained - t_enqueued) * 1e3})
        region_s.append(el)
    samples, kernel_ms_total = eng.profile_read()
    eng.profile(0)
    if grouped:
        t = torch.tensor(region_s, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(x) for x in t.tolist()]
    order = sorted(range(R), key=lambda i_: region_s[i_])
    median_region = order[(R - 1) // 2]
    elapsed = region_s[median_region]
    host_side = dict(host_sides[median_region],
                     what="host wall-clock inside the median timed region of this rank: launching the K steps (asynchronous), then "
                          "closing the region (last bucket's all-reduce = the closing barrier at N > 1, and the synchronize)")

    # The dominant kernel alone on the device (outside the timed region): with 2 streams the timed launches
    # overlap their neighbours, which stretches every per-kernel duration.
    exclusive = None
    my_pci = mc.pci_bus_id(local)
    sclk_headline = None
    if args.exclusive_launches > 0:
        # the clock the card holds while this kernel runs back to back (amdgpu hwmon, sampled by a side thread): the exclusive
        # launches below follow at once, in that state
        clk_out = torch.zeros((len(engines), 3), dtype=torch.float64, device="cuda")
        sclk_headline = sustained_clock(torch, engines, launch_streams,
                                        lambda i, e_, st_: e_.launch(prod, X, structs[engines.index(e_)][0], seed, (1 << 52) + i * shard_count, shard_count,
                                                                     clk_out[engines.index(e_)].data_ptr(), st_), my_pci, 150.0)
        n_ex = min(args.exclusive_launches, 512)
        solo = torch.zeros((n_ex, 3), dtype=torch.float64, device="cuda")
        eng.profile(1)
        for i in range(n_ex):
            eng.launch(prod, X, structs[0][0], seed, (W + R * K + i) * step_total + shard_first, shard_count, solo[i].data_ptr(),
                       launch_streams[0])
        torch.cuda.synchronize()
        ex_samples, ex_ms = eng.profile_read()
        eng.profile(0)
        if ex_samples:
            exclusive = (ex_samples, ex_ms / ex_samples * 1e-3)
    # the shape the workload's OWN kernel was just launched with -- read here, before the fp64 side measurement and the strong
    # rows put other kernels (other grids) on this context: read at the end, the default run compared the committed fp32 grid
    # (2048 workgroups) with the fp64 side kernel's (3072) and called fresh counters stale
    live_grid = eng.last_launch()[0] if exclusive else None

    # BASELINE.json's metric names both precisions: the same step in fp64, measured after (and
    # outside) the headline region, reported as a side figure.
    fp64_side = None
    if args.workload == "vanilla_f32" and args.fp64_steps > 0:
        s64 = [e.prepared("vanilla", "f64", VAN)[0] for e in engines]
        side = torch.zeros((args.fp64_steps + 5, 3), dtype=torch.float64, device="cuda")

        def step64(i):
            e = i % len(engines)
            engines[e].launch("vanilla", "f64", s64[e], seed, i * step_total + shard_first, shard_count, side[i].data_ptr(),
                              launch_streams[e])
        for i in range(5):
            step64(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(5, 5 + args.fp64_steps):
            step64(i)
        for e_ in engines:
            e_.order(stream.cuda_stream)
        if grouped:
            if args.backend == "nccl":
                dist.all_reduce(side[5:], op=dist.ReduceOp.SUM)
            else:
                host = side[5:].cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                side[5:].copy_(host)
        barrier()
        dt64 = time.perf_counter() - t1
        tot64 = side[5:].sum(dim=0).cpu().tolist()
        p64, ci64 = mc.closing(tot64[0], tot64[1], int(tot64[2]), math.exp(-VAN["r"] * VAN["t"]))
        fp64_side = {"value": step_total * args.fp64_steps / dt64, "unit": "paths/s", "steps": args.fp64_steps,
                     "ms_per_step": dt64 / args.fp64_steps * 1e3, "price": p64, "confidence_95": ci64,
                     "price_error_vs_black_scholes": abs(p64 - BS_EXACT),
                     "workload": "same option and path count, fp64 simulation (vanilla_kernel<f64>)"}

    strong = None
    if args.strong_reps > 0 and args.workload == "vanilla_f32":
        strong = strong_scaling_block(mc, torch, dist, eng, stream, rank, world, grouped, args.backend, barrier,
                                      args.strong_reps, args.strong_preheat_ms, full=args.detail == "full")

    # who took part: one entry per rank -- local device index, its PCI bus id (hipDeviceGetPCIBusId through the C ABI), host.
    # Exchanged as fixed-size byte tensors through dist.all_gather (the same call on RCCL and on gloo, also with ONE rank, so the
    # one-GPU boxes exercise the exact code an 8-GPU node runs; all_gather_object would pickle and stage through the device)
    me = {"rank": rank, "device": local, "pci": my_pci, "host": os.uname().nodename[:64], "pid": os.getpid()}
    roster = [me]
    if grouped:
        where = "cuda" if args.backend == "nccl" else "cpu"
        raw = json.dumps(me).encode()[:255]
        mine = torch.zeros(256, dtype=torch.uint8, device=where)
        mine[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8, device=where)
        every = [torch.zeros(256, dtype=torch.uint8, device=where) for _ in range(world)]
        dist.all_gather(every, mine)
        roster = [json.loads(bytes(t.cpu().tolist()).rstrip(b"\0").decode()) for t in every]
        assert [m_["rank"] for m_ in roster] == list(range(world)), roster

    if rank == 0:
        tot = triples[W:].sum(dim=0).cpu().tolist()           # every step's triple is already all-reduced
        r, t_ = float(inputs["r"]), float(inputs["t"])
        if X == "f32":
            import numpy as np
            r, t_ = float(np.float32(r)), float(np.float32(t_))
        disc = 1.0 if prod == "cva" else math.exp(-r * t_)
        price, ci = mc.closing(tot[0], tot[1], int(tot[2]), disc)
        if settings.get("control_variate"):     # the simulated quantity is payoff - control: its closed-form mean comes back on the host
            price += disc * mc.basket_control_mean(inputs, X)
        assert int(tot[2]) == R * K * step_total, (tot[2], R * K * step_total)
        units_per_step = step_total
        value = units_per_step * K / elapsed
        # ---- roofline of the dominant kernel (the simulation kernel) -------------------------------------------
        # Durations, all measured live with HIP events bound to the kernel's own dispatch on its launch stream:
        #   exclusive   n launches one at a time after the timed region: the kernel's duration -> achieved / frac
        #   in_region   sampled launches inside the timed region; with S streams S launches are co-resident and
        #               share the machine, so this is NOT a per-kernel duration (kept for the record)
        #   effective   one GPU's flop per step / the step period: what the whole job sustains per GPU
        peak = PEAK_TFLOPS[X]
        flop_launch = flop_per_path * shard_count
        in_region_s = (kernel_ms_total / samples) * 1e-3 if samples else None
        ex_n, ex_s = exclusive if exclusive else (0, None)
        kernel_s = ex_s or in_region_s
        ach = flop_launch / kernel_s / 1e12 if kernel_s else None
        step_s = elapsed / K
        committed = {}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                committed = json.load(open(pmc)).get(args.workload, {})
            except Exception:
                committed = {}
        # the committed counts describe launches of one device code object and one launch shape: stale once either changes
        # (launch_stamp), or when the grid this run launched differs from the grid the PMC passes ran
        stamp = launch_stamp()
        grid_mismatch = bool(committed) and live_grid is not None and committed.get("grid_workgroups") not in (None, live_grid)
        traffic_stale = bool(committed) and (committed.get("launch_stamp") != stamp["stamp"] or grid_mismatch)
        out = {
            "metric": "Monte Carlo paths/sec", "value": value, "unit": "paths/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": X, "data": "synthetic", "timed_region_s": elapsed, "timed_region_host": host_side,
            "regions": R, "ms_per_step_min": min(region_s) / K * 1e3, "ms_per_step_max": max(region_s) / K * 1e3,
            "region_ms_per_step": [x / K * 1e3 for x in region_s],
            "region_note": "R back-to-back timed regions of exactly K steps each (barrier + synchronize on both sides, max over ranks); "
                           "ms_per_step, value and timed_region_s are the median region's",
            "config": {"workload": desc, "paths_per_gpu_per_step": shard_count, "global_paths_per_step": units_per_step,
                       "parallelism": f"path-sharded x{world}, all-reduce of the fp64 (sum,sum2,n) triples, {args.bucket} steps per message",
                       "rng": "Philox4x32-10 + Box-Muller, counter = global path index", "seed": hex(seed),
                       "grid": f"{eng.blocks}x256", "streams": len(engines), "stream_source": args.stream_source,
                       "finish": args.finish, "preheat_ms": round(preheat_ms, 1),
                       "region_bracket": ("opening: dist.barrier + torch.cuda.synchronize; closing: all-reduce of the last bucket of triples "
                                          "(a barrier by construction) + torch.cuda.synchronize" if closing_is_collective else
                                          "barrier + torch.cuda.synchronize on both sides"), **({"engine_settings": settings} if settings else {})},
            "price": price, "confidence_95": ci, "paths_priced": int(tot[2]),
            "roofline": {"bound": "valu", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": (ach / peak) if ach else None,
                         "traffic": committed.get("hbm_bytes_per_launch"), "traffic_stale": traffic_stale,
                         "launch_stamp": stamp["stamp"][:16], "grid_workgroups": live_grid, "pmc_grid_workgroups": committed.get("grid_workgroups"),
                         "hbm_gbps": (committed["hbm_bytes_per_launch"] * shard_count / committed.get("paths_per_launch", shard_count)
                                      / kernel_s / 1e9) if committed.get("hbm_bytes_per_launch") and kernel_s else None,
                         "hbm_frac_of_8TBps": (committed["hbm_bytes_per_launch"] * shard_count / committed.get("paths_per_launch", shard_count)
                                               / kernel_s / 8e12) if committed.get("hbm_bytes_per_launch") and kernel_s else None,
                         "traffic_source": (committed.get("source", "") + " (committed PMC profile, not measured in this run)")
                         if committed else None,
                         "kernel": kernel_name(prod, X, inputs) + (" [antithetic]" if settings.get("antithetic") else " [control variate]" if settings.get("control_variate") else ""),
                         "flop_per_path": flop_per_path,
                         "avg_kernel_us": kernel_s * 1e6 if kernel_s else None, "kernel_samples": ex_n if ex_s else samples,
                         "kernel_paths_per_s": shard_count / kernel_s if kernel_s else None,
                         "duration_basis": "exclusive: %d launches of the same kernel on the same inputs, one at a time on one "
                                           "stream right after the timed region, each timed on the device by HIP events bound to "
                                           "its dispatch" % ex_n if ex_s else "in-region samples (no exclusive phase was run)",
                         "in_region": {"avg_kernel_us": in_region_s * 1e6 if in_region_s else None, "kernel_samples": samples,
                                       "concurrent_launches": len(engines), "step_period_us": step_s * 1e6,
                                       "note": "launches of different streams are co-resident and share the CUs: a launch lasts "
                                               "about `concurrent_launches` step periods; not a per-kernel duration"},
                         "effective": {"achieved": flop_launch / step_s / 1e12, "frac": flop_launch / step_s / 1e12 / peak,
                                       "paths_per_s_per_gpu": shard_count / step_s,
                                       "note": "flop of one GPU's step / step period (whole job, launch gaps and tails included)"}},
        }
        if traffic_stale:
            out["roofline"]["traffic_stale_note"] = ("the device code object, the launch-shape rules (csrc/mc_launch_shape.hpp), HIPFLAGS or the "
                                                     "launched grid changed since profiles/pmc_traffic.json was collected (tools/collect_pmc_all.sh): "
                                                     "traffic is the old launches', issue_frac is withheld")
        # SURVEY 8d's second fraction (issue_ceiling above): at the chip's peak clock and at the clock measured during this run
        out["roofline"].update(issue_ceiling(args.workload, shard_count, kernel_s, step_s, stamp, traffic_stale, committed, sclk_headline) or {})
        if sclk_headline and "sclk_mhz" not in out["roofline"]:
            out["roofline"]["sclk_mhz"] = sclk_headline
        if prod == "vanilla":
            out["price_error_vs_black_scholes"] = abs(price - BS_EXACT)
        if fp64_side:
            out["fp64"] = fp64_side
        if strong:
            out["strong"] = strong
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(prod, X, inputs, args.cpu_seconds, full=args.detail == "full")
            if prod == "vanilla" and X == "f32":
                extra = cpu_all_cores(1.5)
                if extra:
                    out["cpu_all_cores"] = extra
        if world == 1 and args.workload == "vanilla_f32" and args.configs:
            # BASELINE.json names five configs: configs[0] is the CPU run (tests), the other four are measured here, one GPU
            out["configs"] = configs_block(mc, torch, engines, launch_streams, my_pci, out, strong, stamp,
                                           cpu_seconds=min(2.0, args.cpu_seconds) if args.cpu_seconds > 0 else 0.0)
        if world == 1 and args.c_multi_seconds > 0 and args.workload == "vanilla_f32":
            for e in engines:       # the child process gets the GPU to itself
                e.close()
            engines = []
            out["c_multi"] = c_multi_block(args.c_multi_seconds, full=args.detail == "full")
        # the participants ("did RCCL see N ranks?" must be answerable from the record)
        out["world_size"] = dist.get_world_size() if grouped else 1
        out["backend"] = (dist.get_backend() if grouped else "none (single process)")
        if grouped and args.backend == "nccl":
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                out["rccl_version"] = None
        if world > 1:
            out["ranks"] = [{"rank": m_["rank"], "device": m_["device"], "pci": m_["pci"], "host": m_["host"]} for m_ in roster]
            out["distinct_devices"] = len({(m_["host"], m_["pci"]) for m_ in roster})
        else:
            out["devices_visible"] = torch.cuda.device_count()
            out["device"] = {"index": local, "pci": me["pci"], "name": eng_info["name"], "compute_units": eng_info["compute_units"]}
        out["config"]["detail"] = args.detail
        detail_name = os.path.relpath(args.detail_file, ROOT) if os.path.abspath(args.detail_file).startswith(ROOT + os.sep) else args.detail_file
        line = compact_line(out, detail_name)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail_file)), exist_ok=True)
            with open(args.detail_file, "w") as f:
                json.dump(dict(out, line=line), f, indent=1)
        except OSError as e_:
            line["detail"] = f"not written: {e_}"[:120]
        print(json.dumps(line), flush=True)
    if grouped:
        barrier()
        dist.destroy_process_group()
    for e in engines:
        e.close()


if __name__ == "__main__":
    main()
