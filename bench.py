#!/usr/bin/env python3
"""bench.py -- the hot path's throughput on N MI355X, one JSON line on rank 0.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A STEP is one pass of the hot path over one batch: one pricing call of the workload's path count
on every GPU (simulation kernel + on-device reduction -> {sum, sum2, n} in HBM) and, for N > 1,
the RCCL all-reduce of that 24-byte triple.  Default workload = BASELINE.json configs[1]:
European vanilla call, 1 asset, 1e8 paths, fp32 simulation (fp64 accumulation), per GPU.

Scaling is WEAK: every GPU simulates `paths` paths per step, rank g taking the contiguous global
range [(step*N + g) * paths, +paths) of one Philox stream (no data-path collective besides the
triple).  Steps rotate over `--streams` (default 2) independent context/stream pairs so that one
pricing call's launch gap and finishing kernel overlap the next call's simulation kernel -- fixed
per-call costs (~11 us of a ~61 us step) are hidden, each call is still one full launch.
The triples of `--bucket` consecutive steps are all-reduced as ONE message (fewer, larger
collectives: a 24-byte all-reduce is pure latency), asynchronously on RCCL's stream while the
compute stream keeps simulating; every bucket is waited for inside the timed region.

Reported besides the contract fields:
  roofline      dominant kernel (the simulation kernel) timed with HIP events on its launch stream
                for every 8th step inside the timed region; achieved = algorithmic flop per launch
                (15.5 flop/path vanilla, n^2+12.5n+6 basket, 60/path-step CVA) / that
                duration; the bound is VALU issue, not HBM or MFMA (DESIGN.md "Roofline")
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
    }


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
    return f"mc::cva_kernel<{X}>"


PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # MI355X vector peaks (MI355X_MICROARCH.md; fp64 vector = half)


def cpu_baseline(prod, X, inputs, seconds):
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
    return {"value": n * per_unit / dt, "unit": unit, "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": f"{n} paths of the same workload in {dt:.1f} s, single thread (the reference is single-threaded: "
                      f"MonteCarloHost.c:185-229), gcc -O2 -ffp-contract=off",
            "host_cores_available": len(os.sched_getaffinity(0))}


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
    n = int(min(2 ** 31 - 1, max(n, n * seconds / max(dt, 1e-3))))
    t0 = time.perf_counter(); v = L.host_vanillaOpt(o, n); dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "paths/s", "cores": len(os.sched_getaffinity(0)), "kind": "libmchost_f32 (OpenMP CPU twin of the engine)",
            "sample": f"{n} paths in {dt:.2f} s", "price": float(v.Expected)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="vanilla_f32")
    ap.add_argument("--paths", type=int, default=0, help="paths per GPU per step (default: the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample length (0 = skip)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU simulates `paths` per step. strong: `paths` per step in total, "
                         "rank g taking mc_shard_range(paths, g, N) of every step")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent (context, HIP stream) pairs the steps rotate over; >1 lets consecutive pricing "
                         "calls overlap each other's launch gaps and finishing kernels")
    ap.add_argument("--stream-source", default="torch", choices=["torch", "context"],
                    help="where the launch streams come from: torch's stream pool, or each context's own stream")
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
    ap.add_argument("--bucket", type=int, default=25,
                    help="steps whose triples share one all-reduce (bucketed collective: 24 B x bucket); 1 = one per step")
    args = ap.parse_args()

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
    for e_ in engines:
        e_.set_finish(args.finish == "fused")
    eng = engines[0]
    prod, X, inputs, paths, flop_per_path, desc = workloads(mc)[args.workload]
    if callable(inputs):
        inputs = inputs()
    if args.paths:
        paths = args.paths
    K, W = args.steps, args.warmup
    struct, keep = eng.prepared(prod, X, inputs)
    seed = mc.MC_DEFAULT_SEED
    # Explicit non-default streams, one per context; the first is made torch's CURRENT stream so that
    # torch's copies and RCCL's waits are ordered behind the launches.
    if args.stream_source == "context":   # every context's own non-blocking stream (mc_context_stream)
        streams = [torch.cuda.ExternalStream(e.stream, device=local) for e in engines]
    else:
        streams = [torch.cuda.Stream(device=local) for _ in engines]
    stream = streams[0]
    torch.cuda.set_stream(stream)
    assert all(s_.cuda_stream != 0 for s_ in streams)
    structs = [e.prepared(prod, X, inputs) for e in engines]
    triples = torch.zeros((K + W, 3), dtype=torch.float64, device="cuda")
    works = []

    pending = [0, 0]   # [first step not yet all-reduced, one past the last launched step]

    def flush_bucket():
        # one RCCL all-reduce for the triples of steps [pending[0], pending[1]): the rows are
        # contiguous, so a bucket is a single (bucket x 3) fp64 message, asynchronous to compute
        for s_ in streams[1:]:
            stream.wait_stream(s_)          # the bucket's triples come from every stream
        if grouped and pending[1] > pending[0]:
            rows = triples[pending[0]:pending[1]]
            if args.backend == "nccl":
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

    def step(i):
        first = i * step_total + shard_first
        e = i % len(engines)
        engines[e].launch(prod, X, structs[e][0], seed, first, shard_count, triples[i].data_ptr(), streams[e].cuda_stream)
        pending[1] = i + 1
        if pending[1] - pending[0] >= max(1, args.bucket):
            flush_bucket()

    def drain():
        flush_bucket()
        for w in works:
            w.wait()
        works.clear()
        torch.cuda.synchronize()

    def barrier():
        if grouped:
            dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

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
                                  streams[e].cuda_stream)
                j += 1
            torch.cuda.synchronize()
        preheat_ms = (time.perf_counter() - t_pre) * 1e3
    for i in range(W):
        step(i)
    drain()
    eng.profile(args.profile_every)
    barrier()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    samples, kernel_ms_total = eng.profile_read()
    eng.profile(0)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The dominant kernel alone on the device (outside the timed region): with 2 streams the timed launches
    # overlap their neighbours, which stretches every per-kernel duration.
    exclusive = None
    if args.exclusive_launches > 0:
        n_ex = min(args.exclusive_launches, 512)
        solo = torch.zeros((n_ex, 3), dtype=torch.float64, device="cuda")
        eng.profile(1)
        for i in range(n_ex):
            eng.launch(prod, X, structs[0][0], seed, (W + K + i) * step_total + shard_first, shard_count, solo[i].data_ptr(),
                       streams[0].cuda_stream)
        torch.cuda.synchronize()
        ex_samples, ex_ms = eng.profile_read()
        eng.profile(0)
        if ex_samples:
            exclusive = (ex_samples, ex_ms / ex_samples * 1e-3)

    # BASELINE.json's metric names both precisions: the same step in fp64, measured after (and
    # outside) the headline region, reported as a side figure.
    fp64_side = None
    if args.workload == "vanilla_f32" and args.fp64_steps > 0:
        s64 = [e.prepared("vanilla", "f64", VAN)[0] for e in engines]
        side = torch.zeros((args.fp64_steps + 5, 3), dtype=torch.float64, device="cuda")

        def step64(i):
            e = i % len(engines)
            engines[e].launch("vanilla", "f64", s64[e], seed, i * step_total + shard_first, shard_count, side[i].data_ptr(),
                              streams[e].cuda_stream)
        for i in range(5):
            step64(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(5, 5 + args.fp64_steps):
            step64(i)
        for s_ in streams[1:]:
            stream.wait_stream(s_)
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

    if rank == 0:
        tot = triples[W:].sum(dim=0).cpu().tolist()           # every step's triple is already all-reduced
        r, t_ = float(inputs["r"]), float(inputs["t"])
        if X == "f32":
            import numpy as np
            r, t_ = float(np.float32(r)), float(np.float32(t_))
        disc = 1.0 if prod == "cva" else math.exp(-r * t_)
        price, ci = mc.closing(tot[0], tot[1], int(tot[2]), disc)
        assert int(tot[2]) == K * step_total, (tot[2], K * step_total)
        units_per_step = step_total
        value = units_per_step * K / elapsed
        kernel_s = (kernel_ms_total / samples) * 1e-3 if samples else None
        ach = flop_per_path * shard_count / kernel_s / 1e12 if kernel_s else None
        traffic = valu_busy = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(args.workload, {})
                traffic, valu_busy = rec.get("hbm_bytes_per_launch"), rec.get("valu_busy_long_launch")
            except Exception:
                traffic = valu_busy = None
        out = {
            "metric": "Monte Carlo paths/sec", "value": value, "unit": "paths/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": X, "data": "synthetic",
            "config": {"workload": desc, "paths_per_gpu_per_step": shard_count, "global_paths_per_step": units_per_step,
                       "parallelism": f"path-sharded x{world}, all-reduce of the fp64 (sum,sum2,n) triples, {args.bucket} steps per message",
                       "rng": "Philox4x32-10 + Box-Muller, counter = global path index", "seed": hex(seed),
                       "grid": f"{eng.blocks}x256", "streams": len(engines), "finish": args.finish,
                       "preheat_ms": round(preheat_ms, 1)},
            "price": price, "confidence_95": ci, "paths_priced": int(tot[2]),
            "roofline": {"bound": "valu", "achieved": ach, "peak": PEAK_TFLOPS[X], "unit": "TFLOP/s",
                         "frac": (ach / PEAK_TFLOPS[X]) if ach else None, "traffic": traffic,
                         "kernel": kernel_name(prod, X, inputs), "avg_kernel_us": kernel_s * 1e6 if kernel_s else None,
                         "kernel_samples": samples, "flop_per_path": flop_per_path,
                         "valu_busy": valu_busy,   # PMC, committed profile of a ~10 ms launch of this kernel (not measured live)
                         "kernel_paths_per_s": shard_count / kernel_s if kernel_s else None,
                         "concurrent_launches": len(engines), "step_period_us": elapsed / K * 1e6,
                         "note": "with 2 streams consecutive launches overlap (a launch's tail and finishing kernel run "
                                 "beside the next launch's head), so per-kernel durations exceed the step period"
                                 if len(engines) > 1 else "one launch at a time"},
        }
        if exclusive:
            ex_n, ex_s = exclusive
            ex_ach = flop_per_path * shard_count / ex_s / 1e12
            out["roofline"]["exclusive"] = {
                "avg_kernel_us": ex_s * 1e6, "launches": ex_n, "achieved": ex_ach, "frac": ex_ach / PEAK_TFLOPS[X],
                "kernel_paths_per_s": shard_count / ex_s,
                "note": "the same kernel, one launch at a time on one stream after the timed region"}
        if (prod, X) == ("vanilla", "f32"):
            # SURVEY 8d's second fraction: the issue-slot ceiling of this kernel's instruction mix.  Per wave-trip
            # (4 paths per lane) the ISA has 51 full-rate VALU instructions (4.1-4.2 cycles each next to multiplies,
            # tools/ubench) and 12 transcendentals (8.1-8.3 cycles); with the lower ends 306.3 cycles; 1024 SIMDs x
            # 64 lanes at the clock tools/clock_probe.py measures inside this kernel (2.39 GHz).
            cycles, clock = 51 * 4.1 + 12 * 8.1, 2.39e9
            ceiling = 1024 * 64 * 4 / cycles * clock
            best = max(out["roofline"]["kernel_paths_per_s"] or 0, (exclusive and shard_count / exclusive[1]) or 0, value / world)
            out["roofline"]["issue_model"] = {
                "cycles_per_wave_trip": cycles, "paths_per_wave_trip": 256, "clock_hz": clock,
                "ceiling_paths_per_s": ceiling, "achieved_paths_per_s": best, "frac": best / ceiling,
                "note": "ceiling of the instruction mix actually issued (Philox is integer work and transcendentals are "
                        "half-rate, so the flop fraction above cannot approach 1); achieved = best of the per-GPU whole-job "
                        "rate and the two per-kernel rates"}
        if prod == "vanilla":
            out["price_error_vs_black_scholes"] = abs(price - BS_EXACT)
        if fp64_side:
            out["fp64"] = fp64_side
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(prod, X, inputs, args.cpu_seconds)
            if prod == "vanilla" and X == "f32":
                extra = cpu_all_cores()
                if extra:
                    out["cpu_all_cores"] = extra
        print(json.dumps(out), flush=True)
    if grouped:
        barrier()
        dist.destroy_process_group()
    for e in engines:
        e.close()


if __name__ == "__main__":
    main()
