"""A book of small pricing calls: what the per-call fixed cost is, and what a hipGraph of the asynchronous launches buys.

The reference prices one option per API call and pays allocation + RNG seeding + a blocking copy each time
(dp/MonteCarloKernel.cu:296-363).  Here a call is one kernel launch (mc_*_launch_*: no allocation, no synchronisation), so a
book of B options can be (a) B synchronous calls, (b) B asynchronous launches on one stream and one synchronize,
(c) the same B launches captured once into a hipGraph and replayed (torch.cuda.CUDAGraph is only the capture plumbing),
(d) as (b) but round-robin over 4 CONTEXTS, each on its own stream (small grids leave CUs free: launches of different
contexts can overlap; one context is one in-order pipeline -- its scratch is reused call after call, so moving it between
streams costs a cross-stream dependency per call and is the wrong tool), (e) the 4-context form captured into one hipGraph.
Prints microseconds per call and paths/s for each form.
    python tools/graph_book.py > gpurun_out/graph_book.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlocuda_amd as mc  # noqa: E402

VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
SEED = mc.MC_DEFAULT_SEED if hasattr(mc, "MC_DEFAULT_SEED") else 0x4D435F4D49333535
B = 1024


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    eng = mc.Engine(0)
    eng.set_timing(False)
    engs = [mc.Engine(0) for _ in range(4)]
    print(f"book of {B} vanilla calls; us per call (paths/s)")
    print(f"{'precision':9s} {'paths/call':>10s} {'sync calls':>22s} {'async, 1 stream':>22s} {'hipGraph replay':>22s} {'async, 4 contexts':>22s} {'graph, 4 contexts':>22s}")
    for X in ("f32", "f64"):
        struct, _ = eng.prepared("vanilla", X, VAN)
        for n in (10 ** 4, 10 ** 5, 10 ** 6, 10 ** 7):
            st = torch.cuda.Stream()
            side = [torch.cuda.Stream() for _ in range(4)]
            structs = [e.prepared("vanilla", X, VAN)[0] for e in engs]
            with torch.cuda.stream(st):
                out = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
            ptrs = [out[i].data_ptr() for i in range(B)]

            def sync_calls():
                for i in range(B):
                    eng.vanilla(VAN, n, SEED, i * n, X)

            def enqueue(stream_of):
                for i in range(B):
                    eng.launch("vanilla", X, struct, SEED, i * n, n, ptrs[i], stream_of(i))

            def async_one():
                enqueue(lambda i: st.cuda_stream)

            def async_four():
                for i in range(B):
                    engs[i & 3].launch("vanilla", X, structs[i & 3], SEED, i * n, n, ptrs[i], side[i & 3].cuda_stream)

            async_one()
            torch.cuda.synchronize()
            want = out.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                enqueue(lambda i: torch.cuda.current_stream().cuda_stream)
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert bool((out == want).all()), "graph replay differs from the eager launches"
            ref = eng.vanilla(VAN, n, SEED, 5 * n, X)
            assert want[5].tolist() == [ref.sum, ref.sum2, float(n)]
            out.zero_()
            async_four()
            torch.cuda.synchronize()
            assert bool((out == want).all()), "4-context launches differ"
            # fork / join inside the capture: the side streams wait for the capturing stream and are joined back into it
            g4 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g4, stream=st):
                cur = torch.cuda.current_stream()
                for s_ in side:
                    s_.wait_stream(cur)
                async_four()
                for s_ in side:
                    cur.wait_stream(s_)
            out.zero_()
            g4.replay()
            torch.cuda.synchronize()
            assert bool((out == want).all()), "4-context graph differs"
            cells = []
            for f in (sync_calls, async_one, g.replay, async_four, g4.replay):
                t = best(f)
                cells.append(f"{t / B * 1e6:8.2f} ({B * n / t:9.3e})")
            print(f"{X:9s} {n:10d} " + " ".join(f"{c:>22s}" for c in cells))
    eng.close()
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
