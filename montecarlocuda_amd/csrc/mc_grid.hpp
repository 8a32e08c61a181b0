// mc_grid.hpp -- compatibility mode: the reference's launch geometry and per-thread XORWOW normal streams
// (mc_*_run_grid_*, include/mc_mi355x.h "launch geometry"; host side: mc_api.hip grid_run).
//
// Two forms of the same sample:
//   fused (round 4, the default)  grid_*_kernel below: launched with the REFERENCE's geometry, every thread owns its XORWOW
//           normal stream in registers and walks the reference's own thread-to-path map -- thread t of block b prices paths
//           t, t + T, ... of its block (dp/MonteCarloKernel.cu:146,191,240) -- through the per-path device functions of the
//           hot kernels (vanilla_unit_pk, basket_path, cva_path with a stream policy): no normal touches HBM;
//   staged (round 3, the checker and the fallback for shapes the fused kernels do not cover)  grid_normals_kernel WRITES the
//           normals of the whole call into HBM in path order, the simulation kernels of mc_kernels.hpp then price them
//           through the external-normals policy.
// Per-path values of the two forms are the same bits (tests/test_gpu_grid.py): same normals, same per-path functions.
#pragma once
#include "mc_kernels.hpp"
#include "mc_rng.hpp"

namespace mc {

// ---- the reference's launch geometry (compatibility mode: mc_*_run_grid_*, include/mc_mi355x.h) -------------------
// dp/MonteCarloKernel.cu:285-290 gives every THREAD of a (num_blocks x num_threads) launch its own XORWOW state:
// curand_init(seed = blockIdx.x + gridDim.x, subsequence = threadIdx.x, offset 0) (rocRAND's seeding: what the same call
// gives through hipRAND on this hardware).
// Sub-streams: the reference's 512 x 128 launch is ONE wave per SIMD of an MI355X, and a lone wave issues a dependent
// instruction only every ~8 cycles.  So the fused kernels may cut every thread's stream into `sub` consecutive pieces of
// `step` words each and run the pieces side by side (sub x the waves): row (b * sub + s) * T + t of `states` is thread
// (b, t)'s state advanced by s * step words -- the xorshift part through the offset matrices A^(2^i), the Weyl word by
// s * step * 362437.  The SAME stream, read from `sub` places at once; sub = 1 is the reference's layout (the staged form
// and mc_grid_normals use it).
__global__ __launch_bounds__(256) void xorwow_grid_init_kernel(const uint32_t *__restrict__ jump, uint32_t num_blocks, uint32_t num_threads,
                                                               uint32_t sub, uint32_t step, uint32_t *__restrict__ states)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= num_blocks * sub * num_threads)
        return;
    const uint32_t t = lane % num_threads, bs = lane / num_threads, s = bs % sub, b = bs / sub;
    uint32_t v[5], weyl;
    xorwow_seed((uint64_t)b + num_blocks, v, weyl);
    xorwow_jump(v, t, jump);
    xorwow_skip(v, weyl, s * step, jump);
#pragma unroll
    for (int k = 0; k < 5; ++k)
        states[6 * (size_t)lane + k] = v[k];
    states[6 * (size_t)lane + 5] = weyl;
}

// curand_normal / rocrand_normal on an XORWOW state (rocrand_normal.h: rocrand_normal(rocrand_state_xorwow *),
// box_muller(x, y)): two words make one Box-Muller pair, the sine member is returned first and the cosine member is kept
// for the next call.  Same expressions and the same device math calls as rocRAND's header, so the normals are the ones a
// hipRAND build of the reference draws on this GPU, bit for bit (tests/test_gpu_grid.py).
struct XorwowNormalStream {
    GenXorwow rng;
    float kept;
    bool have = false;
    __device__ __forceinline__ explicit XorwowNormalStream(const uint32_t *states) : rng(states) {}
    __device__ __forceinline__ XorwowNormalStream(const uint32_t *row, GenXorwow::Row tag) : rng(row, tag) {}
    // the next Box-Muller pair: first = the member curand_normal returns first (sine), second = the one it keeps (cosine)
    __device__ __forceinline__ void pair(float &first, float &second)
    {
        const uint32_t x = rng.next(), y = rng.next();
        const float u = 2.3283064e-10f + (x * 2.3283064e-10f);
        const float v = 1.46291807e-09f + (y * 1.46291807e-09f);
        const float s = sqrtf(-2.0f * logf(u));
        float sn, cs;
        __sincosf(v, &sn, &cs);
        first = sn * s, second = cs * s;
    }
    __device__ __forceinline__ float next()
    {
        if (have) {
            have = false;
            return kept;
        }
        float first;
        pair(first, kept);
        have = true;
        return first;
    }
};

// The normals of a whole grid-geometry call, laid out for the external-normals policy (GenExternal): thread t of block b
// prices paths i = t, t + T, ... < paths_per_block of its block (dp/MonteCarloKernel.cu:146,191,240) and draws `draws`
// normals for each, one after the other, from its stream; path p = b * paths_per_block + i gets row p of `ext`
// (`per_unit` Reals: the draws, widened as `double z = curand_normal(...)` does, then zeros).
template <class Real>
__global__ __launch_bounds__(256) void grid_normals_kernel(const uint32_t *__restrict__ states, uint32_t num_blocks, uint32_t num_threads,
                                                           uint64_t paths_per_block, uint32_t draws, uint32_t per_unit,
                                                           Real *__restrict__ ext)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= num_blocks * num_threads)
        return;
    const uint32_t b = lane / num_threads, t = lane % num_threads;
    XorwowNormalStream z(states);
    for (uint64_t i = t; i < paths_per_block; i += num_threads) {
        Real *row = ext + ((uint64_t)b * paths_per_block + i) * per_unit;
        for (uint32_t j = 0; j < draws; ++j)
            row[j] = (Real)z.next();
        for (uint32_t j = draws; j < per_unit; ++j)
            row[j] = (Real)0;
    }
}

// ---- fused form: the reference's launch, thread for thread --------------------------------------------------------------
// Generator policy of the fused kernels: the normals of a "block" are the thread's next draws, widened to Real as
// `double z = curand_normal(...)` does.  WHOLE = every draw of every block is used and blocks are pair-aligned (vanilla: 4 or
// 8 paths per trip, one draw each): whole Box-Muller pairs, no bookkeeping.  Otherwise draw `idx < w.ext_per_unit` of a unit
// is drawn and the rest of the block is zero WITHOUT drawing (a basket path draws n, a CVA path one per date that draws) --
// exactly the rows the staged form writes.  The counter arguments are ignored: the stream is sequential.
template <bool WHOLE>
struct GenGridStream {
    static constexpr bool external = true;   // values, not words: the kernels take their fma(z, b, a) paths
    static constexpr int cursor_phases = 1;
    template <class Real> static constexpr int npb() { return sizeof(Real) == 4 ? 4 : 8; }
    XorwowNormalStream z;
    __device__ __forceinline__ explicit GenGridStream(const uint32_t *row) : z(row, GenXorwow::Row{}) {}
    template <class Real, int N>
    __device__ __forceinline__ void normals(const Work &w, uint32_t, uint32_t block, uint32_t, Real (&out)[N])
    {
        if constexpr (WHOLE) {
            static_assert(N % 2 == 0, "whole pairs");
#pragma unroll
            for (int j = 0; j < N; j += 2) {
                float a, b;
                z.pair(a, b);
                out[j] = (Real)a, out[j + 1] = (Real)b;
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j)
                out[j] = block * (uint32_t)N + (uint32_t)j < w.ext_per_unit ? (Real)z.next() : (Real)0;
        }
    }
    struct Carry { const F64K *K = nullptr; };
    __device__ __forceinline__ void pair(const Work &w, uint32_t, uint32_t, uint32_t P, Carry &, double &z0, double &z1)
    {
        z0 = 2u * P < w.ext_per_unit ? (double)z.next() : 0.0;
        z1 = 2u * P + 1u < w.ext_per_unit ? (double)z.next() : 0.0;
    }
    __device__ __forceinline__ void pairs_done(uint32_t) {}
};

// The reference's launch: num_blocks blocks of num_threads threads, paths_per_block paths per block.  The fused kernels are
// launched with num_blocks * sub workgroups of num_threads rounded up to whole waves (the extra lanes idle): workgroup
// b * sub + s runs piece s of every thread of the reference's block b -- the thread's paths [s * seg, (s + 1) * seg) of its
// own t, t + T, ... sequence (`seg` a multiple of 8 paths, so a piece starts on a Box-Muller pair boundary whatever a path
// draws) -- from the start states xorwow_grid_init_kernel prepared for (sub, step = seg * draws per path).
struct GridGeom {
    const uint32_t *states;
    uint32_t num_threads;
    uint32_t paths_per_block;
    uint32_t sub, seg;   // pieces per thread; paths per piece
};
constexpr int GRID_MAX_THREADS = 1024;   // the reference's blockDim limit
constexpr uint32_t GRID_SEG_ALIGN = 8;   // paths: a whole fp64 vanilla trip, and an even number of draws for any draws per path

// this lane's piece of its reference thread's paths t, t + T, ... < paths_per_block: local path numbers [lo, hi)
__device__ __forceinline__ void grid_thread_piece(const GridGeom &geo, uint32_t &lo, uint32_t &hi)
{
    const uint32_t n_t = threadIdx.x < geo.paths_per_block ? (geo.paths_per_block - threadIdx.x + geo.num_threads - 1) / geo.num_threads : 0u;
    const uint32_t s = blockIdx.x % geo.sub;
    lo = s * geo.seg;
    hi = lo + geo.seg < n_t ? lo + geo.seg : n_t;
    if (lo > hi)
        lo = hi;
}
__device__ __forceinline__ const uint32_t *grid_state_row(const GridGeom &geo)
{
    return geo.states + 6u * (blockIdx.x * geo.num_threads + threadIdx.x);   // row (b * sub + s) * T + t
}
// index (in the call's path order, block-major) of this thread's k-th path: what `out` is indexed by
__device__ __forceinline__ uint64_t grid_path_index(const GridGeom &geo, uint32_t k)
{
    return (uint64_t)(blockIdx.x / geo.sub) * geo.paths_per_block + threadIdx.x + (uint64_t)k * geo.num_threads;
}

// Vanilla: one draw per path, so a trip of NPB draws = NPB consecutive paths of the thread = NPB / 2 whole Box-Muller pairs.
// vanilla_unit_pk / vanilla_unit are the hot kernels' per-unit functions (mc_kernels.hpp) on their external-normals path.
// DUMP = the per-path values are also stored (tests): its own instantiation, so that the pricing form carries no store and
// no path-index arithmetic.  Whole trips run unmasked; the thread's last, partial trip is peeled.
template <bool DUMP>
__global__ __launch_bounds__(GRID_MAX_THREADS) void grid_vanilla_f32_kernel(const Tail /* first argument, read late */, const VanillaF32 o, const Work w,
                                                                            const GridGeom geo, float *__restrict__ out, float out_scale)
{
    double acc_s = 0.0, acc_q = 0.0;
    if (threadIdx.x < geo.num_threads) {
        uint32_t lo, n_t;
        grid_thread_piece(geo, lo, n_t);
        const uint32_t whole = lo + ((n_t - lo) & ~3u);
        GenGridStream<true> gen(grid_state_row(geo));
        f2 s2 = {0.0f, 0.0f}, q2 = {0.0f, 0.0f};   // fp32 partial sums, flushed to fp64 every 8 trips like the hot kernel's
        for (uint32_t k = lo; k < whole; k += 4) {
            f2 pc, ps;   // {path k, k + 2}, {path k + 1, k + 3}
            vanilla_unit_pk<false>(gen, o, w, 0u, pc, ps);
            if (DUMP) {
                out[grid_path_index(geo, k)] = pc.x * out_scale, out[grid_path_index(geo, k + 1)] = ps.x * out_scale;
                out[grid_path_index(geo, k + 2)] = pc.y * out_scale, out[grid_path_index(geo, k + 3)] = ps.y * out_scale;
            }
            s2 += pc;
            s2 += ps;
            q2 = pk_fma(pc, pc, q2);
            q2 = pk_fma(ps, ps, q2);
            if ((k & 28u) == 28u) {
                acc_s += (double)(s2.x + s2.y);
                acc_q += (double)(q2.x + q2.y);
                s2 = (f2){0.0f, 0.0f};
                q2 = (f2){0.0f, 0.0f};
            }
        }
        if (whole < n_t) {   // 1..3 paths left: the draws beyond them are made and dropped (nothing follows in this stream)
            float p[4];
            vanilla_unit<false>(gen, o, w, 0u, p);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (whole + j < n_t) {
                    if (DUMP)
                        out[grid_path_index(geo, whole + j)] = p[j] * out_scale;
                    s2.x += p[j];
                    q2.x = __builtin_fmaf(p[j], p[j], q2.x);
                }
        }
        acc_s += (double)(s2.x + s2.y);
        acc_q += (double)(q2.x + q2.y);
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}
template <bool DUMP>
__global__ __launch_bounds__(GRID_MAX_THREADS) void grid_vanilla_f64_kernel(const Tail /* first argument, read late */, const VanillaF64 o, const Work w,
                                                                            const GridGeom geo, double *__restrict__ out, double out_scale)
{
    stage_tables<double>();
    double acc_s = 0.0, acc_q = 0.0;
    if (threadIdx.x < geo.num_threads) {
        uint32_t lo, n_t;
        grid_thread_piece(geo, lo, n_t);
        const uint32_t whole = lo + ((n_t - lo) & ~7u);
        GenGridStream<true> gen(grid_state_row(geo));
        for (uint32_t k = lo; k < whole; k += 8) {
            double p[8];
            vanilla_unit<false>(gen, o, w, 0u, p);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (DUMP)
                    out[grid_path_index(geo, k + j)] = p[j] * out_scale;
                acc_s += p[j];
                acc_q = __builtin_fma(p[j], p[j], acc_q);
            }
        }
        if (whole < n_t) {
            double p[8];
            vanilla_unit<false>(gen, o, w, 0u, p);
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (whole + j < n_t) {
                    if (DUMP)
                        out[grid_path_index(geo, whole + j)] = p[j] * out_scale;
                    acc_s += p[j];
                    acc_q = __builtin_fma(p[j], p[j], acc_q);
                }
        }
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// Basket: a path draws its n normals one after the other (an odd n makes a Box-Muller pair straddle two paths, as in the
// reference).  basket_path is the kernel-argument family's per-path function; NA is the compiled size, a smaller basket
// runs it zero-padded (rows beyond n: zero factor, base and weight -- they add exactly 0; w.ext_per_unit = n draws).
template <class Real, int NA, bool DUMP>
__global__ __launch_bounds__(GRID_MAX_THREADS) void grid_basket_kernel(const Tail /* first argument, read late */, const BasketArgs<Real, NA> o, const Work w,
                                                                       const GridGeom geo, Real *__restrict__ out, Real out_scale)
{
    stage_tables<Real>();
    // constants from LDS where the hot kernels stage them -- except fp64 at 8 assets, where hipcc keeps the staged values in
    // more registers than a 1024-thread workgroup may have (128: it spilled 332 bytes of scratch per lane)
    constexpr bool IN_LDS = basket_consts_in_lds<Real, NA>() && !(sizeof(Real) == 8 && NA == 8);
    __shared__ Real lds_consts[IN_LDS ? ConstsLds<Real, NA>::COUNT : 1];
    if (IN_LDS)
        ConstsLds<Real, NA>::stage(lds_consts, o);
    double acc_s = 0.0, acc_q = 0.0;
    if (threadIdx.x < geo.num_threads) {
        uint32_t lo, n_t;
        grid_thread_piece(geo, lo, n_t);
        GenGridStream<false> gen(grid_state_row(geo));
        for (uint32_t k = lo; k < n_t; ++k) {
            Real p;
            if constexpr (IN_LDS)
                p = basket_path<Real, NA, false>(gen, o, ConstsLds<Real, NA>{lds_consts}, w, 0u);
            else
                p = basket_path<Real, NA, false>(gen, o, ConstsArg<Real, NA>{o}, w, 0u);
            acc_s += (double)p;
            acc_q = __builtin_fma((double)p, (double)p, acc_q);
            if (DUMP)
                out[grid_path_index(geo, k)] = p * out_scale;
        }
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// CVA: cva_path is the hot kernel's per-path function; a path draws for the dates whose `t -= dt` is still >= 0
// (w.ext_per_unit of them, dp/MonteCarloKernel.cu:249), the remaining dates of its blocks read 0 without drawing.
template <class Real, bool DUMP>
__global__ __launch_bounds__(GRID_MAX_THREADS) void grid_cva_kernel(const Tail /* first argument, read late */, const CvaArgs<Real> o, const Work w,
                                                                    const GridGeom geo, Real *__restrict__ out)
{
    stage_tables<Real>();
    double acc_s = 0.0, acc_q = 0.0;
    if (threadIdx.x < geo.num_threads) {
        uint32_t lo, n_t;
        grid_thread_piece(geo, lo, n_t);
        GenGridStream<false> gen(grid_state_row(geo));
        for (uint32_t k = lo; k < n_t; ++k) {
            const Real p = cva_path<false>(gen, o, w, 0u);
            acc_s += (double)p;
            acc_q = __builtin_fma((double)p, (double)p, acc_q);
            if (DUMP)
                out[grid_path_index(geo, k)] = p;
        }
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

}  // namespace mc
