// mc_kernels.hpp -- the simulation kernels (gfx950, wave64).
//
// Shape common to all three products (replaces the kernel skeleton of
// dp/MonteCarloKernel.cu:133-177,179-220,222-283):
//   * a persistent-style grid (blocks x 256 lanes) strides over "units" of work; a unit is one
//     block of vanilla paths (4 in f32 = one Philox block, 8 in f64 = three) or one whole basket / CVA path;
//   * everything a lane needs is in registers: counter-based normals (mc_rng.hpp), per-lane fp64
//     (sum, sum2) accumulators; wave-uniform constants never cost a VALU slot -- kernel arguments in
//     SGPRs while they fit, otherwise LDS-staged (broadcast ds_read) or fetched tile by tile with
//     scalar loads from constant memory (the basket kernels), per-date rows through scalar loads (CVA);
//   * one DPP+LDS reduction and one 16-byte store per workgroup at the end; the last workgroup of the call to
//     arrive adds the pairs and writes the call's {sum, sum2, n} (mc_reduce.hpp: finish_group).
// HBM traffic per launch: kernel arguments (and a table of <= 20 KB) in, 16 B per workgroup out --
// the kernels are bound by VALU/transcendental issue, not by memory (DESIGN.md "Roofline").
//
// Work (one segment of units) and the generator policies (`Gen`: where the normals come from): mc_rng.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mc_reduce.hpp"
#include "mc_rng.hpp"

namespace mc {

constexpr int GROUP = 256;  // lanes per workgroup = 4 waves = one wave per SIMD

// precision-generic fused multiply-add (__builtin_fma alone is the double form)
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bcast(float x) { return (f2){x, x}; }
__device__ __forceinline__ f2 pk_exp2(f2 x) { return (f2){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

// =========================================================================================
// Vanilla call.  Reference device formula, dp/MonteCarloKernel.cu:67-71:
//     payoff = max(S exp((r - v^2/2) T + v sqrt(T) z) - K, 0)
// =========================================================================================

// f32 constants, all prepared in fp64 on the host (mc_api.hip VanillaTraits<float>::prepare):
//   payoff = S 2^k * clamp(2^(a2k + b2 z) - kappa_k, 0, 1)
//   a2k = (r - v^2/2) T log2(e) - k,  b2 = v sqrt(T) log2(e),  kappa_k = (K/S) 2^-k,
//   radius2 = -2 ln2 * b2^2  (folds b2 into Box-Muller's radius: one fma per path after its normal)
//   k = integer with 2^(a2k + b2 |z|) <= 1 for every z the generator can produce (|z| < 6.77):
//       the payoff's max(.,0) then rides on the subtract as the free [0,1] output clamp
//       (v_sub_f32 ... clamp) instead of a separate v_max_f32; 2^k is exact, so the scaling
//       costs no precision.  Sums are scaled back by S 2^k and (S 2^k)^2 in the finishing kernel.
struct VanillaF32 {
    float a2k, radius2, kappa_k;
    float b2;   // read by the external-normals form only (the hot form has it folded into radius2)
};
struct VanillaF64 {
    double drift, vol, strike, spot;
};

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// clamp(a + b, 0, 1) for BOTH halves of a packed pair in ONE instruction: v_pk_add_f32 with the output clamp.  hipcc
// never selects it (a packed min/max pair becomes two v_pk ops plus two scalar v_max ... clamp), hence the asm.  The
// fp32 vanilla payoffs max(2^x - kappa, 0) of two paths cost one issue slot this way instead of two v_sub_f32 ... clamp
// (63 -> 61 VALU instructions per Philox block; -2.7 % time, profiles/r02_ab_packed_clamp.log).  Not used in the basket
// kernels: there the "v" operands push wave-uniform constants out of the scalar registers and cost more v_readlane than
// the packing saves (145 -> 152 at n = 4).
__device__ __forceinline__ f2 pk_add_clamp01(f2 a, f2 b)
{
    // s_nop 0: gfx950 needs one wait state between a transcendental (the v_exp_f32 that produced `a`) and a VALU
    // instruction reading its result; hipcc inserts it for its own instructions but does not look inside an asm
    // (without it the antithetic kernel read a stale exponential: caught by test_vanilla_many_trips_vs_oracle)
    f2 r;
    asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Box-Muller for the packed-fp32 kernels: the two word pairs (xa, ya) and (xb, yb) -- one Philox block's two pairs in
// the vanilla kernel, the same pair of TWO paths' blocks in the basket kernels -- as the halves of packed registers, so
// the uniform scaling, the radius scaling and z = r trig issue as v_pk_*_f32 (2 results per 4-cycle issue slot).
// Returns the radius {a, b} scaled by sqrt(scale2 / (-2 ln 2)) and the trig values; callers form z = r c, r s (or fold
// them into an fma).  v_log_f32 is log2 and v_sin/v_cos_f32 take revolutions: 2 pi never appears.
__device__ __forceinline__ void box_muller_pk(uint32_t xa, uint32_t ya, uint32_t xb, uint32_t yb, float scale2, f2 &rad, f2 &c, f2 &s)
{
    const f2 ua = pk_fma((f2){(float)xa, (float)xb}, bcast(0x1p-32f), bcast(0x1p-33f));  // radius uniforms
    const f2 ub = {angle_f32(ya), angle_f32(yb)};                                        // angle uniforms (revolutions, in [1, 2))
    const f2 t = (f2){__builtin_amdgcn_logf(ua.x), __builtin_amdgcn_logf(ua.y)} * bcast(scale2);
    rad = (f2){__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
    c = (f2){__builtin_amdgcn_cosf(ub.x), __builtin_amdgcn_cosf(ub.y)};
    s = (f2){__builtin_amdgcn_sinf(ub.x), __builtin_amdgcn_sinf(ub.y)};
}

// The 4 scaled payoffs of one unit (one Philox block).  Box-Muller pair A = words (x, y), pair B = words
// (z, w); values are kept as {A, B} register pairs:
//   pc = cos-branch payoffs {A, B} = paths 4q+0, 4q+2;  ps = sin-branch {A, B} = paths 4q+1, 4q+3
// b2 is folded into the radius (o.radius2), so a path's exponent is ONE fma after its trig value.
//
// ANTI = antithetic variates (not in the reference): every normal z also prices the
// mirrored path -z; the sample is the mean of the two payoffs (here their sum: the 1/2 rides on the
// finishing step's scale).  Costs one more fma + exponential + clamp-subtract per path.
//
// GenExternal (from-normals hooks, launch-geometry mode): the four normals come from memory and the exponent is fma(z, b2, a2k); everything after the
// exponent -- exponential, clamp-subtract, sums, flushes, final reduction -- is the code of the hot path.
template <bool ANTI, class Gen>
__device__ __forceinline__ void vanilla_unit_pk(Gen &gen, const VanillaF32 &o, const Work &w, uint32_t c0, f2 &pc, f2 &ps)
{
    const f2 a = bcast(o.a2k);
    f2 yc, ys, mc_, ms_;   // exponents (log2 units) of the cos / sin branch paths and of their mirrors
    if constexpr (Gen::external) {
        float z[4];
        gen.normals(w, c0, 0u, 1u /*MC_DOMAIN_VANILLA*/, z);
        const f2 zc = {z[0], z[2]}, zs = {z[1], z[3]}, b = bcast(o.b2);
        yc = pk_fma(zc, b, a), ys = pk_fma(zs, b, a);
        mc_ = pk_fma(-zc, b, a), ms_ = pk_fma(-zs, b, a);
    } else {
        const u32x4 r = gen.words(w, c0, 0u, 1u /*MC_DOMAIN_VANILLA*/);
        f2 rad, c, s;
        box_muller_pk(r.x, r.y, r.z, r.w, o.radius2, rad, c, s);
        yc = pk_fma(c, rad, a), ys = pk_fma(s, rad, a);
        mc_ = pk_fma(-c, rad, a), ms_ = pk_fma(-s, rad, a);
    }
    const f2 neg_kappa = bcast(-o.kappa_k);
    pc = pk_add_clamp01(pk_exp2(yc), neg_kappa);
    ps = pk_add_clamp01(pk_exp2(ys), neg_kappa);
    if (ANTI) {
        pc += pk_add_clamp01(pk_exp2(mc_), neg_kappa);
        ps += pk_add_clamp01(pk_exp2(ms_), neg_kappa);
    }
}

// path order inside the unit: 4q+0 = cos A, 4q+1 = sin A, 4q+2 = cos B, 4q+3 = sin B
template <bool ANTI, class Gen>
__device__ __forceinline__ void vanilla_unit(Gen &gen, const VanillaF32 &o, const Work &w, uint32_t c0, float (&p)[4])
{
    f2 pc, ps;
    vanilla_unit_pk<ANTI>(gen, o, w, c0, pc, ps);
    p[0] = pc.x;
    p[1] = ps.x;
    p[2] = pc.y;
    p[3] = ps.y;
}
template <bool ANTI, class Gen, int NPB>
__device__ __forceinline__ void vanilla_unit(Gen &gen, const VanillaF64 &o, const Work &w, uint32_t c0, double (&p)[NPB])
{
    double z[NPB];
    gen.normals(w, c0, 0u, 1u /*MC_DOMAIN_VANILLA*/, z);
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        p[j] = fmax(o.spot * exp_f64(o.drift + o.vol * z[j]) - o.strike, 0.0);
        if (ANTI)
            p[j] = 0.5 * (p[j] + fmax(o.spot * exp_f64(o.drift - o.vol * z[j]) - o.strike, 0.0));
    }
}

// Hot kernels: every unit is complete.
//
// f32: per-lane sums live in two packed-f32 pairs for FLUSH iterations (16 values per fp32
// accumulator) and are then flushed to the lane's fp64 accumulators -- never a long fp32 running
// sum.  The trip count is wave-uniform (scalar loop control); the one partial
// trip at the end is peeled.
constexpr uint32_t VANILLA_F32_FLUSH = 8;

template <bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void vanilla_f32_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const VanillaF32 o, const Work w)
{
    const uint32_t stride = gridDim.x * GROUP;
    const uint32_t gtid = blockIdx.x * GROUP + threadIdx.x;
    const uint32_t full_trips = w.n_units / stride;
    double acc_s = 0.0, acc_q = 0.0;
    f2 s2 = {0.0f, 0.0f}, q2 = {0.0f, 0.0f};
    uint32_t c0 = w.unit_lo + gtid;
    Gen gen(w);   // Philox: stateless; XORWOW: the lane's sequence (units ascending, as the masked kernel draws them)
    for (uint32_t trip = 0; trip < full_trips; ++trip, c0 += stride) {
        f2 pc, ps;
        vanilla_unit_pk<ANTI>(gen, o, w, c0, pc, ps);
        s2 += pc;
        s2 += ps;
        q2 = pk_fma(pc, pc, q2);
        q2 = pk_fma(ps, ps, q2);
        if ((trip & (VANILLA_F32_FLUSH - 1)) == VANILLA_F32_FLUSH - 1) {
            acc_s += (double)(s2.x + s2.y);
            acc_q += (double)(q2.x + q2.y);
            s2 = (f2){0.0f, 0.0f};
            q2 = (f2){0.0f, 0.0f};
        }
    }
    if (full_trips * stride + gtid < w.n_units) {  // the partial last trip
        f2 pc, ps;
        vanilla_unit_pk<ANTI>(gen, o, w, c0, pc, ps);
        s2 += pc + ps;
        q2 += pc * pc + ps * ps;
    }
    acc_s += (double)(s2.x + s2.y);
    acc_q += (double)(q2.x + q2.y);
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// f64 (and the generic form): each unit's payoffs go straight into the fp64 accumulators.
template <class Opt, class Real, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void vanilla_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const Opt o, const Work w)
{
    stage_tables<Real>();
    constexpr int NPB = Gen::template npb<Real>();
    const uint32_t stride = gridDim.x * GROUP;
    const uint32_t gtid = blockIdx.x * GROUP + threadIdx.x;
    double acc_s = 0.0, acc_q = 0.0;
    Gen gen(w);
    for (uint32_t i = gtid; i < w.n_units; i += stride) {
        Real p[NPB];
        vanilla_unit<ANTI>(gen, o, w, w.unit_lo + i, p);
        Real s = p[0], q = p[0] * p[0];
#pragma unroll
        for (int j = 1; j < NPB; ++j) {
            s += p[j];
            q = fma_r(p[j], p[j], q);
        }
        acc_s += (double)s;
        acc_q += (double)q;
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// Masked kernel: honours the path window (partial first/last units) and optionally stores
// every per-path payoff (currency units) to `out[p - first_path]`.  Used for range edges and
// by the parity tests; never on the hot path.
template <class Opt, class Real, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void vanilla_masked_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const Opt o, const Work w,
                                                               Real *__restrict__ out, Real out_scale)
{
    stage_tables<Real>();
    constexpr int NPB = Gen::template npb<Real>();
    const uint32_t stride = gridDim.x * GROUP;
    const uint32_t gtid = blockIdx.x * GROUP + threadIdx.x;
    double acc_s = 0.0, acc_q = 0.0;
    Gen gen(w);
    for (uint32_t i = gtid; i < w.n_units; i += stride) {
        Real p[NPB];
        vanilla_unit<ANTI>(gen, o, w, w.unit_lo + i, p);
        const uint64_t unit = ((uint64_t)w.unit_hi << 32) | (uint32_t)(w.unit_lo + i);
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const uint64_t path = unit * NPB + j;
            if (path >= w.first_path && path < w.end_path) {
                acc_s += (double)p[j];
                acc_q += (double)p[j] * (double)p[j];
                if (out)
                    out[path - w.first_path] = p[j] * out_scale;
            }
        }
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// =========================================================================================
// Vanilla call with pathwise Greeks (the reference prices only).  Per path, on the same
// normal as the pricing kernels:  S_T = S exp(drift + vol z),  I = [S_T > K],
//     payoff = I (S_T - K),   d payoff / dS = I S_T / S,   d payoff / d sigma = I S_T (sqrt(T) z - sigma T)
// Three (sum, sum2) pairs per workgroup, one per plane of the call's pair buffer: q = price, delta, vega.
// Always honours the path window (no separate hot variant: a secondary kernel).
// =========================================================================================
// lr_delta = 1 / (S sigma sqrt T), inv_sigma = 1 / sigma: the scores of the likelihood-ratio estimators (LR = true):
//     delta = payoff z / (S sigma sqrt T),   vega = payoff ((z^2 - 1) / sigma - z sqrt T)
// (differentiate the lognormal density instead of the payoff: no indicator, so they also serve payoffs with jumps).
struct GreeksF32 { float drift2, vol2, spot, strike, sqrt_t, sigma_t, lr_delta, inv_sigma; };   // exponent in log2 units
struct GreeksF64 { double drift, vol, spot, strike, sqrt_t, sigma_t, lr_delta, inv_sigma; };

__device__ __forceinline__ float greeks_spot(const GreeksF32 &o, float z) { return o.spot * __builtin_amdgcn_exp2f(__builtin_fmaf(o.vol2, z, o.drift2)); }
__device__ __forceinline__ double greeks_spot(const GreeksF64 &o, double z) { return o.spot * exp_f64(__builtin_fma(o.vol, z, o.drift)); }

template <class Opt, class Real, bool LR>
__global__ __launch_bounds__(GROUP) void vanilla_greeks_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const Opt o, const Work w)
{
    stage_tables<Real>();
    constexpr int NPB = GenPhilox::npb<Real>();
    const uint32_t stride = gridDim.x * GROUP;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    GenPhilox gen(w);
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += stride) {
        Real z[NPB];
        gen.normals(w, w.unit_lo + i, 0u, 1u /*MC_DOMAIN_VANILLA*/, z);
        const uint64_t unit = ((uint64_t)w.unit_hi << 32) | (uint32_t)(w.unit_lo + i);
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const uint64_t path = unit * NPB + j;
            if (path >= w.first_path && path < w.end_path) {
                const Real st = greeks_spot(o, z[j]);
                const bool itm = st > o.strike;
                const Real payoff = itm ? st - o.strike : (Real)0;
                const double pay = (double)payoff;
                double dl, vg;
                if (LR) {
                    dl = (double)(payoff * z[j] * o.lr_delta);
                    vg = (double)(payoff * ((z[j] * z[j] - (Real)1) * o.inv_sigma - z[j] * o.sqrt_t));
                } else {
                    dl = itm ? (double)(st / o.spot) : 0.0;
                    vg = itm ? (double)(st * (o.sqrt_t * z[j] - o.sigma_t)) : 0.0;
                }
                acc[0] += pay, acc[1] = __builtin_fma(pay, pay, acc[1]);
                acc[2] += dl, acc[3] = __builtin_fma(dl, dl, acc[3]);
                acc[4] += vg, acc[5] = __builtin_fma(vg, vg, acc[5]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        group_sum2(acc[2 * q], acc[2 * q + 1]);
        publish_pair(late_tail(acc[2 * q]), q, acc[2 * q], acc[2 * q + 1]);
    }
    arrive_and_finish(late_tail(acc[0]));
}

// =========================================================================================
// Basket call.  Reference device formulas, dp/MonteCarloKernel.cu:74-101:
//     bt = L g + d;  s_a = S_a exp((r - v_a^2/2) T + v_a sqrt(T) bt_a);
//     payoff = max(sum_a w_a s_a - K, 0)
// folded on the host (fp64) into  payoff = max(sum_a coef_a E(base_a + sum_{b<=a} m_ab g_b) - K, 0)
// with m = diag(v sqrt T) L (lower triangle only: the reference multiplies the structural
// zeros too, :79-84), base_a = (r - v_a^2/2) T + v_a sqrt(T) d_a, coef_a = w_a S_a;
// in f32 m and base are pre-multiplied by log2(e) and E = 2^x (v_exp_f32).
// =========================================================================================
template <class Real, int NA>
struct BasketArgs {
    Real m[NA * (NA + 1) / 2];  // packed rows: m[a(a+1)/2 + b], b <= a
    Real base[NA];
    Real coef[NA];
    Real strike;
    // geometric-basket control variate (off in the reference's plain estimator):
    // ln G = cg + sum_a wg_a x_a in the kernel's exponent units, x_a = asset a's exponent.  When
    // cv != 0 the per-path value is payoff(arithmetic) - payoff(geometric); the closed-form mean of
    // the latter (mc_basket_control_mean_*) is added back on the host.
    Real wg[NA];
    Real cg;
    int cv;
};

// Where a basket kernel reads its folded constants from.  Small baskets keep them in SGPRs straight from
// the kernel arguments; larger ones do not fit the scalar register file (hipcc then "spills" SGPRs into
// VGPR lanes and pays a v_readlane_b32 -- a full VALU issue slot -- per use: 272 of 870 instructions per
// trip at n=16 f32), so they are staged once per workgroup into LDS and read back as broadcast
// ds_reads, which do not occupy the VALU.
// Which sizes stage (in-process A/B on MI355X, tools/ab_basket.py, profiles/r01_ab_basket_lds.log; kernel time LDS vs
// SGPR constants):
//   f32  n=6 -1 %, n=8 -11 %, n=10 -15 % (no fence);  n=12 -15 %, n=16 -20 % (fence every 4 rows; -6 % / -5 % without)
//   f64  n=4 -7 %, n=7 -10 %, n=8 -12 %, n=9 -11 %, n=10 +1 %, n=12 +2 %, n=16 +6 % (no fence; a fence costs f64
//        3-9 %).  From n=10 hipcc keeps the staged constants in > 256 registers: one wave per SIMD, every LDS
//        latency exposed; asking for more waves (amdgpu_waves_per_eu) turns that into scratch spills, so those
//        sizes stay on the SGPR path (and run the tiled kernels by default: mc_api.hip).
template <class Real, int NA> constexpr bool basket_consts_in_lds()
{
    return sizeof(Real) == 4 ? NA > 5 : (NA > 3 && NA <= 9);
}
// Whether the LDS reads are additionally pinned every few rows (ConstsLds::fence): the fence period in rows (0 = never).
template <class Real, int NA> constexpr int basket_fence_rows() { return sizeof(Real) == 4 && NA >= 11 ? 4 : 0; }

template <class Real, int NA>
struct ConstsArg {  // kernel arguments (SGPRs)
    const BasketArgs<Real, NA> &o;
    __device__ __forceinline__ Real m(int i) const { return o.m[i]; }
    __device__ __forceinline__ Real base(int a) const { return o.base[a]; }
    __device__ __forceinline__ Real coef(int a) const { return o.coef[a]; }
    __device__ __forceinline__ Real wg(int a) const { return o.wg[a]; }
    template <class T> __device__ __forceinline__ void fence(int, T) {}
};
template <class Real, int NA>
struct ConstsLds {  // staged copy: m | base | coef | wg
    static constexpr int NM = NA * (NA + 1) / 2, COUNT = NM + 3 * NA;
    const Real *p;
    int off = 0;  // always 0, but opaque to the compiler (see fence)
    __device__ __forceinline__ Real m(int i) const { return p[off + i]; }
    __device__ __forceinline__ Real base(int a) const { return p[off + NM + a]; }
    __device__ __forceinline__ Real coef(int a) const { return p[off + NM + NA + a]; }
    __device__ __forceinline__ Real wg(int a) const { return p[off + NM + 2 * NA + a]; }
    // The staged constants never change, so left alone hipcc hoists every ds_read out of the path loop
    // (or to the top of a trip) and pins ~140 values in VGPRs -- 168 VGPRs and scratch at n=16.  Tying
    // the read offset to the value computed just before (a row's exponent) keeps each row's reads next to
    // the row that uses them.
    template <class T> __device__ __forceinline__ void fence(int row, T dep)
    {
        constexpr int PERIOD = basket_fence_rows<Real, NA>();
        if (PERIOD > 0 && row % (PERIOD > 0 ? PERIOD : 1) == 0)  // row is a compile-time constant after unrolling
            asm volatile("" : "+v"(off) : "v"(dep));
    }
    // Thread 0 writes every constant with a compile-time index (SGPR -> v_mov -> ds_write): a per-thread
    // index into the kernel-argument struct would make hipcc copy the whole struct to scratch first.
    __device__ __forceinline__ static void stage(Real *lds, const BasketArgs<Real, NA> &o)
    {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < NM; ++i)
                lds[i] = o.m[i];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                lds[NM + a] = o.base[a];
                lds[NM + NA + a] = o.coef[a];
                lds[NM + 2 * NA + a] = o.wg[a];
            }
        }
        __syncthreads();
    }
};

__device__ __forceinline__ float exp_model(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double exp_model(double x) { return exp_f64(x); }

// The end of a basket path, shared by every one-path-per-lane basket kernel: from the weighted sum of the terminal
// prices (and of the mirrored path's, ANTI) to the sample.  Reference: dp/MonteCarloKernel.cu:96-100.
//   cv (wave-uniform): geometric-basket control variate -- the sample is payoff(arithmetic) - payoff(geometric),
//   lg / lgm = ln G of the path and of its mirror in the kernel's exponent units.
template <bool ANTI, class Real>
__device__ __forceinline__ Real basket_sample(Real basket, Real mirror, Real lg, Real lgm, Real strike, int cv)
{
    const Real v = basket - strike;
    Real pay = v > 0 ? v : 0;
    if (cv) {
        const Real gv = exp_model(lg) - strike;
        pay -= gv > 0 ? gv : 0;
    }
    if (ANTI) {
        const Real vm = mirror - strike;
        Real pm = vm > 0 ? vm : 0;
        if (cv) {
            const Real gm = exp_model(lgm) - strike;
            pm -= gm > 0 ? gm : 0;
        }
        pay = (Real)0.5 * (pay + pm);
    }
    return pay;
}
// the same for two paths in packed halves (the fp32 tiled / generic kernels: plain max, no power-of-two rescale)
template <bool ANTI>
__device__ __forceinline__ f2 basket_sample_pk(f2 basket, f2 mirror, f2 lg, f2 lgm, float strike_, int cv)
{
    const f2 zero = {0.0f, 0.0f}, strike = bcast(strike_);
    f2 p = __builtin_elementwise_max(basket - strike, zero);
    if (cv)
        p -= __builtin_elementwise_max(pk_exp2(lg) - strike, zero);
    if (ANTI) {
        f2 pm = __builtin_elementwise_max(mirror - strike, zero);
        if (cv)
            pm -= __builtin_elementwise_max(pk_exp2(lgm) - strike, zero);
        p = bcast(0.5f) * (p + pm);
    }
    return p;
}

// One row of the folded model: the asset's exponent x is done; add its terminal price (and the mirrored path's, whose
// exponent is base - m g = 2 base - x) to the weighted sums, and its log to the control variate's.
// wg() yields the row's control-variate weight; it is only evaluated where it is used (for the LDS-staged constants it
// is a read), cv is wave-uniform.
template <bool ANTI, class Real, class Wg>
__device__ __forceinline__ void basket_row(Real x, Real base, Real coef, Wg wg, int cv, Real &basket, Real &mirror, Real &lg, Real &lgm)
{
    basket = fma_r(coef, exp_model(x), basket);
    if (cv)
        lg = fma_r(wg(), x, lg);
    if (ANTI) {
        const Real xm = fma_r((Real)-1, x, 2 * base);
        mirror = fma_r(coef, exp_model(xm), mirror);
        if (cv)
            lgm = fma_r(wg(), xm, lgm);
    }
}
template <bool ANTI, class Wg>
__device__ __forceinline__ void basket_row_pk(f2 x, float base, float coef, Wg wg, int cv, f2 &basket, f2 &mirror, f2 &lg, f2 &lgm)
{
    basket = pk_fma(bcast(coef), pk_exp2(x), basket);
    if (cv)
        lg = pk_fma(bcast(wg()), x, lg);
    if (ANTI) {
        const f2 xm = pk_fma(bcast(-1.0f), x, bcast(2.0f * base));
        mirror = pk_fma(bcast(coef), pk_exp2(xm), mirror);
        if (cv)
            lgm = pk_fma(bcast(wg()), xm, lgm);
    }
}

// All normals of one path: blocks 0 .. NBLK-1 of unit c0 in the basket domain
template <class Gen, class Real, int NG>
__device__ __forceinline__ void basket_normals(Gen &gen, const Work &w, uint32_t c0, Real (&g)[NG])
{
    constexpr int NPB = Gen::template npb<Real>();
    static_assert(NG % NPB == 0, "whole blocks");
#pragma unroll
    for (int b = 0; b < NG / NPB; ++b) {
        Real z[NPB];
        gen.normals(w, c0, (uint32_t)b, 2u /*MC_DOMAIN_BASKET*/, z);
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            g[b * NPB + j] = z[j];
    }
}
// ... of TWO paths (units cA, cB) as packed halves {A, B}: the fp32 two-paths-per-lane kernels.  Blocks first_block ..
// first_block + NBLK - 1; dst(k) = where normal k of them goes (a register array or the lane's LDS column).
template <int NBLK, class Gen, class Dst>
__device__ __forceinline__ void basket_normals_pk(Gen &gen, const Work &w, uint32_t cA, uint32_t cB, Dst dst, uint32_t first_block = 0)
{
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        if constexpr (Gen::external) {
            float za[4], zb[4];
            gen.normals(w, cA, first_block + (uint32_t)b, 2u /*MC_DOMAIN_BASKET*/, za);
            gen.normals(w, cB, first_block + (uint32_t)b, 2u, zb);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst(4 * b + j) = (f2){za[j], zb[j]};
        } else {
            const u32x4 ra = gen.words(w, cA, first_block + (uint32_t)b, 2u /*MC_DOMAIN_BASKET*/);
            const u32x4 rb = gen.words(w, cB, first_block + (uint32_t)b, 2u);
            // Box-Muller pair X = words (x, y), pair Z = words (z, w); halves = {path A, path B}
            f2 rx, cx, sx, rz, cz, sz;
            box_muller_pk(ra.x, ra.y, rb.x, rb.y, NEG_2LN2_F32, rx, cx, sx);
            box_muller_pk(ra.z, ra.w, rb.z, rb.w, NEG_2LN2_F32, rz, cz, sz);
            dst(4 * b + 0) = rx * cx;
            dst(4 * b + 1) = rx * sx;
            dst(4 * b + 2) = rz * cz;
            dst(4 * b + 3) = rz * sz;
        }
    }
}

template <class Real, int NA, bool ANTI, class Gen, class Consts>
__device__ __forceinline__ Real basket_path(Gen &gen, const BasketArgs<Real, NA> &o, Consts k, const Work &w, uint32_t c0)
{
    constexpr int NPB = Gen::template npb<Real>();
    constexpr int NBLK = (NA + NPB - 1) / NPB;
    Real g[NBLK * NPB];
    basket_normals(gen, w, c0, g);
    Real basket = 0, mirror = 0, lg = o.cg, lgm = o.cg;
    k.fence(0, g[0]);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const Real base = k.base(a), coef = k.coef(a);
        Real x = base;
#pragma unroll
        for (int b = 0; b <= a; ++b)
            x = fma_r(k.m(a * (a + 1) / 2 + b), g[b], x);
        basket_row<ANTI>(x, base, coef, [&] { return k.wg(a); }, o.cv, basket, mirror, lg, lgm);
        k.fence(a + 1, x);
    }
    return basket_sample<ANTI>(basket, mirror, lg, lgm, o.strike, o.cv);
}

template <class Real, int NA, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void basket_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketArgs<Real, NA> o, const Work w,
                                                       Real *__restrict__ out)
{
    stage_tables<Real>();
    const uint32_t stride = gridDim.x * GROUP;
    const uint32_t gtid = blockIdx.x * GROUP + threadIdx.x;
    double acc_s = 0.0, acc_q = 0.0;
    constexpr bool IN_LDS = basket_consts_in_lds<Real, NA>();
    __shared__ Real lds_consts[IN_LDS ? ConstsLds<Real, NA>::COUNT : 1];
    if (IN_LDS)
        ConstsLds<Real, NA>::stage(lds_consts, o);
    Gen gen(w);
    for (uint32_t i = gtid; i < w.n_units; i += stride) {
        Real p;
        if constexpr (IN_LDS)
            p = basket_path<Real, NA, ANTI>(gen, o, ConstsLds<Real, NA>{lds_consts}, w, w.unit_lo + i);
        else
            p = basket_path<Real, NA, ANTI>(gen, o, ConstsArg<Real, NA>{o}, w, w.unit_lo + i);
        acc_s += (double)p;
        acc_q = __builtin_fma((double)p, (double)p, acc_q);
        if (out)  // wave-uniform: per-path dump for the parity tests
            out[i] = p;
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// ---- f32 basket: TWO paths per lane, one in each half of a packed-f32 register pair ----------
// Everything that is not Philox or a transcendental (uniform scaling, radius scaling, z = r*trig,
// the whole triangular mat-vec, the weighted sum, the sums) issues as v_pk_*_f32: two paths per
// 4-cycle slot.  The payoff's max(.,0) is the free [0,1] clamp of the subtract after an exact
// power-of-two rescale (coef and strike carry 2^-k, k from the generator's |z| < 6.77 bound);
// the finishing kernel scales the sums back.  Lane pairing: units i and i + stride of one trip.
template <int NA, bool ANTI, class Gen, class Consts>
__device__ __forceinline__ f2 basket_pair_f32(Gen &gen, const BasketArgs<float, NA> &o, Consts k, const Work &w, uint32_t cA, uint32_t cB)
{
    constexpr int NBLK = (NA + 3) / 4;
    f2 g[NBLK * 4];
    basket_normals_pk<NBLK>(gen, w, cA, cB, [&](int i) -> f2 & { return g[i]; });
    f2 basket = {0.0f, 0.0f}, mirror = {0.0f, 0.0f}, lg = bcast(o.cg), lgm = bcast(o.cg);
    k.fence(0, g[0].x);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const float base = k.base(a), coef = k.coef(a);
        f2 x = bcast(base);
#pragma unroll
        for (int b = 0; b <= a; ++b)
            x = pk_fma(bcast(k.m(a * (a + 1) / 2 + b)), g[b], x);
        basket_row_pk<ANTI>(x, base, coef, [&] { return k.wg(a); }, o.cv, basket, mirror, lg, lgm);
        k.fence(a + 1, x.x);
    }
    // payoff = the free [0,1] clamp of the subtract after the power-of-two rescale; the geometric mean never exceeds
    // the arithmetic one, so the same 2^-k scale keeps it in [0,1]
    f2 pay = {clamp01(basket.x - o.strike), clamp01(basket.y - o.strike)};
    if (o.cv)  // wave-uniform
        pay -= (f2){clamp01(__builtin_amdgcn_exp2f(lg.x) - o.strike), clamp01(__builtin_amdgcn_exp2f(lg.y) - o.strike)};
    if (ANTI) {  // sum of the two values; the 1/2 rides on the finishing step's scale
        pay += (f2){clamp01(mirror.x - o.strike), clamp01(mirror.y - o.strike)};
        if (o.cv)
            pay -= (f2){clamp01(__builtin_amdgcn_exp2f(lgm.x) - o.strike), clamp01(__builtin_amdgcn_exp2f(lgm.y) - o.strike)};
    }
    return pay;
}

constexpr uint32_t BASKET_F32_FLUSH = 8;

template <int NA, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void basket_f32_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketArgs<float, NA> o, const Work w, float *__restrict__ out,
                                                           float out_scale)
{
    const uint32_t stride = gridDim.x * GROUP;
    const uint32_t gtid = blockIdx.x * GROUP + threadIdx.x;
    const uint32_t full_trips = w.n_units / (2 * stride);  // trips in which every lane has both units
    double acc_s = 0.0, acc_q = 0.0;
    f2 s2 = {0.0f, 0.0f}, q2 = {0.0f, 0.0f};
    constexpr bool IN_LDS = basket_consts_in_lds<float, NA>();
    __shared__ float lds_consts[IN_LDS ? ConstsLds<float, NA>::COUNT : 1];
    if (IN_LDS)
        ConstsLds<float, NA>::stage(lds_consts, o);
    Gen gen(w);
    const auto pair = [&](uint32_t cA, uint32_t cB) __attribute__((always_inline)) {
        if constexpr (IN_LDS)
            return basket_pair_f32<NA, ANTI>(gen, o, ConstsLds<float, NA>{lds_consts}, w, cA, cB);
        else
            return basket_pair_f32<NA, ANTI>(gen, o, ConstsArg<float, NA>{o}, w, cA, cB);
    };
    uint32_t i = gtid;
    for (uint32_t trip = 0; trip < full_trips; ++trip, i += 2 * stride) {
        const f2 p = pair(w.unit_lo + i, w.unit_lo + i + stride);
        s2 += p;
        q2 = pk_fma(p, p, q2);
        if (out) {  // wave-uniform: per-path dump for the parity tests
            out[i] = p.x * out_scale;
            out[i + stride] = p.y * out_scale;
        }
        if ((trip & (BASKET_F32_FLUSH - 1)) == BASKET_F32_FLUSH - 1) {
            acc_s += (double)(s2.x + s2.y);
            acc_q += (double)(q2.x + q2.y);
            s2 = (f2){0.0f, 0.0f};
            q2 = (f2){0.0f, 0.0f};
        }
    }
    if (i < w.n_units) {  // the partial last trip: unit i, and unit i + stride where it exists
        const bool has_b = i + stride < w.n_units;
        f2 p = pair(w.unit_lo + i, w.unit_lo + (has_b ? i + stride : i));
        if (!has_b)
            p.y = 0.0f;
        s2 += p;
        q2 = pk_fma(p, p, q2);
        if (out) {
            out[i] = p.x * out_scale;
            if (has_b)
                out[i + stride] = p.y * out_scale;
        }
    }
    acc_s += (double)(s2.x + s2.y);
    acc_q += (double)(q2.x + q2.y);
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// ---- generic basket: runtime asset count beyond the compiled sizes (17 .. MC_MAX_ASSETS_GENERIC) ----
// The reference's N is any compile-time constant (MonteCarlo.h:16); sizes without a register-resident
// specialisation run here.  A lane's normals live in its own column of a dynamic-LDS array
// g[asset][lane] (consecutive lanes, consecutive addresses: conflict-free, and nobody else touches
// the column, so no barrier).  The triangular mat-vec is blocked 4 rows x 4 columns: one ds_read of
// g[b] feeds four rows' fmas (LDS traffic / 4), and the host lays the folded matrix out block-row by
// block-row, 4 row-values per column, zero-padded to whole blocks, so that each 4 x 4 tile is 16
// consecutive reals behind ONE wave-uniform (scalar) load.  Padded rows have coef = 0, padded columns
// multiply real normals by 0.  Same stream and estimator definitions as the specialised kernels.
template <class Real>
struct BasketDyn {
    const Real *consts;  // tiles (8 nb (nb + 1) reals, nb = ceil(n / 4)), then base[4 nb], coef[4 nb], wg[4 nb]
    int n;
    Real strike;
    Real cg;  // control variate constant (see BasketArgs)
    int cv;
};

template <class Real, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void basket_dyn_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketDyn<Real> o, const Work w, Real *__restrict__ out)
{
    stage_tables<Real>();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Real *g = reinterpret_cast<Real *>(lds_raw) + threadIdx.x;  // this lane's column, stride GROUP
    constexpr int NPB = Gen::template npb<Real>();
    const int nb = (o.n + 3) >> 2, np = nb * 4, nblk = (np + NPB - 1) / NPB;   // whole blocks: the column holds nblk * NPB normals
    // constant address space: wave-uniform reads become scalar loads (s_load_dwordx16 per tile) whatever the
    // compiler can or cannot prove about the kernel's global stores
    typedef const __attribute__((address_space(4))) Real *cptr;
    const cptr tiles = (cptr)o.consts, base = tiles + 8 * nb * (nb + 1), coef = base + np, wg = coef + np;
    const uint32_t stride = gridDim.x * GROUP;
    double acc_s = 0.0, acc_q = 0.0;
    Gen gen(w);
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += stride) {
        for (int b = 0; b < nblk; ++b) {
            Real z[NPB];
            gen.normals(w, w.unit_lo + i, (uint32_t)b, 2u /*MC_DOMAIN_BASKET*/, z);
#pragma unroll
            for (int j = 0; j < NPB; ++j)
                g[(b * NPB + j) * GROUP] = z[j];
        }
        Real basket = 0, mirror = 0, lg = o.cg, lgm = o.cg;
        cptr tile = tiles;
        for (int A = 0; A < nb; ++A) {
            Real x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[r] = base[4 * A + r];
            for (int c4 = 0; c4 <= A; ++c4, tile += 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const Real gb = g[(4 * c4 + j) * GROUP];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        x[r] = fma_r(tile[4 * j + r], gb, x[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)   // wg is zero without the control variate: its sums are formed unconditionally
                basket_row<ANTI>(x[r], (Real)base[4 * A + r], (Real)coef[4 * A + r], [&] { return (Real)wg[4 * A + r]; }, 1, basket, mirror, lg, lgm);
        }
        const Real p = basket_sample<ANTI>(basket, mirror, lg, lgm, o.strike, o.cv);
        acc_s += (double)p;
        acc_q = __builtin_fma((double)p, (double)p, acc_q);
        if (out)
            out[i] = p;
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// fp32 form of the generic kernel: TWO paths per lane (units i and i + stride), one in each half of a packed
// register pair, like basket_f32_kernel -- the uniforms' scaling, the radius scaling, z = r trig, every fma of
// the tiled mat-vec and the weighted sums issue as v_pk_*_f32, and each scalar-loaded tile value and each
// loop step serves two paths.  The lane's LDS column holds packed pairs (8 bytes per asset).
template <bool ANTI>
__global__ __launch_bounds__(GROUP) void basket_dyn_f32_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketDyn<float> o, const Work w, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    f2 *g = reinterpret_cast<f2 *>(lds_raw) + threadIdx.x;  // this lane's column, stride GROUP
    const int nb = (o.n + 3) >> 2, np = nb * 4;             // nb Philox blocks (4 normals each) per path
    typedef const __attribute__((address_space(4))) float *cptr;
    const cptr tiles = (cptr)o.consts, base = tiles + 8 * nb * (nb + 1), coef = base + np, wg = coef + np;
    const uint32_t stride = gridDim.x * GROUP;
    double acc_s = 0.0, acc_q = 0.0;
    GenPhilox gen(w);
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += 2 * stride) {
        const bool has_b = i + stride < w.n_units;  // false only in a range's last trip
        const uint32_t cA = w.unit_lo + i, cB = w.unit_lo + (has_b ? i + stride : i);
        for (int b = 0; b < nb; ++b)
            basket_normals_pk<1>(gen, w, cA, cB, [&](int k) -> f2 & { return g[(4 * b + k) * GROUP]; }, (uint32_t)b);
        f2 basket = {0.0f, 0.0f}, mirror = {0.0f, 0.0f}, lg = bcast(o.cg), lgm = bcast(o.cg);
        cptr tile = tiles;
        for (int A = 0; A < nb; ++A) {
            f2 x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[r] = bcast(base[4 * A + r]);
            for (int c4 = 0; c4 <= A; ++c4, tile += 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f2 gb = g[(4 * c4 + j) * GROUP];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        x[r] = pk_fma(bcast(tile[4 * j + r]), gb, x[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                basket_row_pk<ANTI>(x[r], base[4 * A + r], coef[4 * A + r], [&] { return wg[4 * A + r]; }, 1, basket, mirror, lg, lgm);
        }
        f2 p = basket_sample_pk<ANTI>(basket, mirror, lg, lgm, o.strike, o.cv);
        if (!has_b)
            p.y = 0.0f;
        acc_s += (double)p.x + (double)p.y;
        acc_q = __builtin_fma((double)p.x, (double)p.x, __builtin_fma((double)p.y, (double)p.y, acc_q));
        if (out) {
            out[i] = p.x;
            if (has_b)
                out[i + stride] = p.y;
        }
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// ---- tiled basket with register-resident normals: the larger fp64 sizes -----------------------
// Same constant buffer and tile layout as the generic kernel, but the asset count is bounded at compile
// time, the normals stay in registers and everything is unrolled (padding rows, columns and the zeros
// above the diagonal are skipped at compile time).  The folded matrix never sits in
// registers across a trip: each half tile (4 rows x 2 columns = 8 reals) is one scalar load from constant
// memory issued one half tile ahead of its use, so the constants cost no VALU slot (the SGPR path pays a
// v_readlane_b32 per use once they no longer fit: 329 of 1457 instructions at n=16) and no LDS latency.
// What keeps hipcc from hoisting these loop-invariant loads out of the path loop (and spilling them again)
// is an offset it cannot see through: an empty volatile asm that redefines `off` (always 0) and is tied
// to the value computed just before, which also fixes where in the trip each load is issued.
// A wave-uniform value (in an SGPR pair) into a vector register pair with ONE v_mov_b64 (hipcc emits two v_mov_b32).
__device__ __forceinline__ double scalar_to_vgpr(double x)
{
    double r;
    asm("v_mov_b64 %0, %1" : "=v"(r) : "s"(x));
    return r;
}
__device__ __forceinline__ float scalar_to_vgpr(float x) { return x; }

template <class Real, int NA, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void basket_tiled_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketDyn<Real> o, const Work w, Real *__restrict__ out)
{
    stage_tables<Real>();
    constexpr int NB = (NA + 3) / 4, NP = 4 * NB, NPB = Gen::template npb<Real>(), NBLK = (NA + NPB - 1) / NPB;
    constexpr int NH = NB * (NB + 1);  // half tiles in the buffer
    typedef const __attribute__((address_space(4))) Real *cptr;
    const cptr tiles = (cptr)o.consts, base = tiles + 8 * NH, coef = base + NP, wg = coef + NP;
    const uint32_t stride = gridDim.x * GROUP;
    double acc_s = 0.0, acc_q = 0.0;
    Gen gen(w);
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += stride) {
        Real g[NBLK * NPB];
        basket_normals(gen, w, w.unit_lo + i, g);
        Real basket = 0, mirror = 0, lg = o.cg, lgm = o.cg;
        Real tl[2][8];
        int off = 0;
        asm volatile("" : "+s"(off) : "v"(g[0]));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            tl[0][k] = tiles[off + k];
        int slot = 0;  // which of the two register tiles holds the half tile in use (compile-time after unrolling)
#pragma unroll
        for (int A = 0; A < NB; ++A) {
            constexpr int ROWS_LAST = NA - 4 * (NB - 1);
            const int rows = A == NB - 1 ? ROWS_LAST : 4;   // rows of this block row that exist
            const int halves = (4 * A + rows + 1) / 2;      // half tiles that hold a column <= the last row's diagonal
            Real x[4], cf[4], wr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < rows) {
                    x[r] = scalar_to_vgpr(base[off + 4 * A + r]);
                    cf[r] = coef[off + 4 * A + r];
                    wr[r] = wg[off + 4 * A + r];
                }
#pragma unroll
            for (int c2 = 0; c2 < 2 * (A + 1); ++c2) {
                if (c2 >= halves)
                    continue;
                // the half tile used next: the following one of this block row, or the first of the next block row
                const int h = A * (A + 1) + c2;
                const int h_next = c2 + 1 < halves ? h + 1 : (A + 1 < NB ? (A + 1) * (A + 2) : -1);
                if (h_next >= 0) {  // issued now, used after this half tile's fmas
                    asm volatile("" : "+s"(off) : "v"(x[0]));
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        tl[slot ^ 1][k] = tiles[off + 8 * h_next + k];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < rows && 2 * c2 + j <= 4 * A + r)   // inside the lower triangle
                            x[r] = fma_r(tl[slot][4 * j + r], g[2 * c2 + j], x[r]);
                slot ^= 1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < rows)
                    basket_row<ANTI>(x[r], (Real)base[off + 4 * A + r], cf[r], [&] { return wr[r]; }, 1, basket, mirror, lg, lgm);
        }
        const Real p = basket_sample<ANTI>(basket, mirror, lg, lgm, o.strike, o.cv);
        acc_s += (double)p;
        acc_q = __builtin_fma((double)p, (double)p, acc_q);
        if (out)
            out[i] = p;
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// fp32 form of the tiled kernel (12..32 assets): two paths per lane in packed halves (like basket_f32_kernel
// and basket_dyn_f32_kernel), normals in registers, each whole 4 x 4 tile (16 floats) one scalar load issued a
// tile ahead.  Plain max for the payoff (no power-of-two rescale: the host folds none for these sizes).
template <int NA, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void basket_tiled_f32_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketDyn<float> o, const Work w, float *__restrict__ out)
{
    constexpr int NB = (NA + 3) / 4, NP = 4 * NB, NT = NB * (NB + 1) / 2;
    typedef const __attribute__((address_space(4))) float *cptr;
    const cptr tiles = (cptr)o.consts, base = tiles + 16 * NT, coef = base + NP, wg = coef + NP;
    const uint32_t stride = gridDim.x * GROUP;
    double acc_s = 0.0, acc_q = 0.0;
    Gen gen(w);
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += 2 * stride) {
        const bool has_b = i + stride < w.n_units;  // false only in a range's last trip
        const uint32_t cA = w.unit_lo + i, cB = w.unit_lo + (has_b ? i + stride : i);
        f2 g[NP];
        basket_normals_pk<NB>(gen, w, cA, cB, [&](int k) -> f2 & { return g[k]; });
        f2 basket = {0.0f, 0.0f}, mirror = {0.0f, 0.0f}, lg = bcast(o.cg), lgm = bcast(o.cg);
        float tl[2][16];
        int off = 0;
        asm volatile("" : "+s"(off) : "v"(g[0].x));
#pragma unroll
        for (int k = 0; k < 16; ++k)
            tl[0][k] = tiles[off + k];
        int t = 0;  // tile counter (compile-time after unrolling)
#pragma unroll
        for (int A = 0; A < NB; ++A) {
            constexpr int ROWS_LAST = NA - 4 * (NB - 1);
            const int rows = A == NB - 1 ? ROWS_LAST : 4;  // rows of this block row that exist
            f2 x[4];
            float cf[4], wr[4], bs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < rows) {
                    bs[r] = base[off + 4 * A + r];
                    cf[r] = coef[off + 4 * A + r];
                    wr[r] = wg[off + 4 * A + r];
                    x[r] = bcast(bs[r]);
                }
#pragma unroll
            for (int c4 = 0; c4 <= A; ++c4, ++t) {
                if (t + 1 < NT) {  // next tile: issued now, used after this one's fmas
                    asm volatile("" : "+s"(off) : "v"(x[0].x));
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        tl[(t + 1) & 1][k] = tiles[off + 16 * (t + 1) + k];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < rows && 4 * c4 + j <= 4 * A + r)   // an existing row, inside the lower triangle
                            x[r] = pk_fma(bcast(tl[t & 1][4 * j + r]), g[4 * c4 + j], x[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < rows)
                    basket_row_pk<ANTI>(x[r], bs[r], cf[r], [&] { return wr[r]; }, 1, basket, mirror, lg, lgm);
        }
        f2 p = basket_sample_pk<ANTI>(basket, mirror, lg, lgm, o.strike, o.cv);
        if (!has_b)
            p.y = 0.0f;
        acc_s += (double)p.x + (double)p.y;
        acc_q = __builtin_fma((double)p.x, (double)p.x, __builtin_fma((double)p.y, (double)p.y, acc_q));
        if (out) {
            out[i] = p.x;
            if (has_b)
                out[i + stride] = p.y;
        }
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// ---- 16-asset fp64 basket with the correlation step on the matrix cores ------------------------
// The Cholesky step x = base + M g of 64 paths is a 16 x 16 by 16 x 64 product (BASELINE.json north_star: "MFMA only if
// the basket Cholesky step is cast as a dense small-matrix contraction"): four column blocks of 16 paths, each
// 4 x v_mfma_f64_16x16x4_f64, take the 136 v_fma_f64 per path of the mat-vec out of the vector pipe, which is what the
// kernel is bound by.  No transposition through LDS: Philox is counter-based, so a lane simply
// generates the normals the B operand wants from it.  Lane l = 16 q + j holds, for column block c (paths i0 + 16 c + j),
// the normals of Philox blocks q and q + 4 of that path, i.e. columns k(s, q) = 2 q, 2 q + 1, 2 q + 8, 2 q + 9 of M for
// the k-steps s = 0..3 -- the host lays M out as the A operand of each step in exactly that column order (BasketDyn
// consts, after wg: a4[s][lane] = M[j][k(s, q)]).  A lane does the same RNG work as before (8 blocks, 16 normals per
// trip), for 4 paths x 4 columns instead of 1 path x 16.
// The accumulators come back as X[asset q + 4 v][path i0 + 16 c + j], v = 0..3: 16 exponentials per lane as before,
// the weighted sum over assets is 4 in-lane fmas per column block plus a sum over the four lane groups, done as a
// reduce-scatter with v_permlane32_swap / v_permlane16_swap (6 swaps + 3 adds) that leaves lane l with the basket of
// path i0 + l: units, per-lane sums and the closing reduction are those of every other basket kernel.
// Differences from them: the order of the additions inside x and inside the basket sum (results agree to a few ulp,
// not bit for bit), and a wave prices all 64 paths of a trip or none (paths beyond the range are computed and dropped).
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// a's lanes 32..63 <-> b's lanes 0..31
__device__ __forceinline__ void swap_halves(double &a, double &b)
{
    uint64_t ua = __builtin_bit_cast(uint64_t, a), ub = __builtin_bit_cast(uint64_t, b);
    const u32x2 lo = __builtin_amdgcn_permlane32_swap((uint32_t)ua, (uint32_t)ub, false, false);
    const u32x2 hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32), false, false);
    a = __builtin_bit_cast(double, (uint64_t)lo.x | ((uint64_t)hi.x << 32));
    b = __builtin_bit_cast(double, (uint64_t)lo.y | ((uint64_t)hi.y << 32));
}
// a's odd rows of 16 lanes <-> b's even rows
__device__ __forceinline__ void swap_rows(double &a, double &b)
{
    uint64_t ua = __builtin_bit_cast(uint64_t, a), ub = __builtin_bit_cast(uint64_t, b);
    const u32x2 lo = __builtin_amdgcn_permlane16_swap((uint32_t)ua, (uint32_t)ub, false, false);
    const u32x2 hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32), false, false);
    a = __builtin_bit_cast(double, (uint64_t)lo.x | ((uint64_t)hi.x << 32));
    b = __builtin_bit_cast(double, (uint64_t)lo.y | ((uint64_t)hi.y << 32));
}
// p[c] = lane group q's share of the sum for path block c; returns the whole sum for path block q (this lane's own path)
__device__ __forceinline__ double sum_over_lane_groups(double p0, double p1, double p2, double p3)
{
    swap_halves(p0, p2);   // lower half keeps c = 0, upper half c = 2
    swap_halves(p1, p3);   //                  c = 1             c = 3
    double r0 = p0 + p2, r1 = p1 + p3;
    swap_rows(r0, r1);     // even rows keep r0's block, odd rows r1's
    return r0 + r1;
}

// Pair q (a lane-dependent 0..3) of fp64 block `block` of a path: words W[3q .. 3q + 2] of the block's twelve
// (mc_rng.hpp: words_to_normals).  They sit in Philox blocks 3 block + {0,0,1,2}[q] and 3 block + {0,1,2,2}[q]: two
// Philox blocks per pair here (the other kernels, which use all four pairs of a lane's block, pay three per four pairs) --
// one more reason this variant is not the default.
__device__ __forceinline__ void mfma_pair_normals(GenPhilox &gen, const Work &w, uint32_t unit, uint32_t block, int q, double (&z)[2])
{
    const uint32_t ba = (3u * (uint32_t)q) >> 2, bb = (3u * (uint32_t)q + 2u) >> 2;
    const u32x4 ra = gen.words(w, unit, 3u * block + ba, 2u /*MC_DOMAIN_BASKET*/);
    const u32x4 rb = gen.words(w, unit, 3u * block + bb, 2u);
    const uint32_t a = q == 0 ? ra.x : (q == 1 ? ra.w : (q == 2 ? ra.z : ra.y));   // W[3q]
    const uint32_t m = q == 0 ? ra.y : (q == 1 ? rb.x : (q == 2 ? ra.w : ra.z));   // W[3q + 1]
    const uint32_t c = q == 0 ? ra.z : (q == 1 ? rb.y : (q == 2 ? rb.x : rb.w));   // W[3q + 2]
    pair_normals_f64(a, m, c, z[0], z[1]);
}

template <bool ANTI>
__global__ __launch_bounds__(GROUP) void basket_mfma_f64_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketDyn<double> o, const Work w, double *__restrict__ out)
{
    stage_tables<double>();
    constexpr int NB = 4, NP = 16, NH = NB * (NB + 1);
    const double *base = o.consts + 8 * NH, *coef = base + NP, *wg = coef + NP, *a4 = wg + NP;
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    double A[4], cb[4], cf[4], wr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        A[s] = a4[64 * s + lane];
        cb[s] = base[q + 4 * s];
        cf[s] = coef[q + 4 * s];
        wr[s] = wg[q + 4 * s];
    }
    const uint32_t stride = gridDim.x * GROUP;
    double acc_s = 0.0, acc_q = 0.0;
    GenPhilox gen(w);
    // wave-uniform trip count: the matrix instructions want all 64 lanes
    for (uint32_t i0 = blockIdx.x * GROUP + (threadIdx.x & ~63u); i0 < w.n_units; i0 += stride) {
        double pb[4], pm[4], pl[4], plm[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t unit = w.unit_lo + i0 + 16u * c + j;
            double za[2], zb[2];
            mfma_pair_normals(gen, w, unit, 0u, q, za);   // columns 2q, 2q + 1:     pair q of the path's first fp64 block
            mfma_pair_normals(gen, w, unit, 1u, q, zb);   // columns 2q + 8, 2q + 9: pair q of its second
            d4 x = {cb[0], cb[1], cb[2], cb[3]};
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0], za[0], x, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1], za[1], x, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2], zb[0], x, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(A[3], zb[1], x, 0, 0, 0);
            double b = 0, m = 0, l = 0, lm = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                basket_row<ANTI>(x[v], cb[v], cf[v], [&] { return wr[v]; }, 1, b, m, l, lm);
            pb[c] = b, pm[c] = m, pl[c] = l, plm[c] = lm;
        }
        // the sums over the four lane groups (the control variate's only when it is on: wave-uniform)
        const double basket = sum_over_lane_groups(pb[0], pb[1], pb[2], pb[3]);
        const double mirror = ANTI ? sum_over_lane_groups(pm[0], pm[1], pm[2], pm[3]) : 0.0;
        const double lg = o.cv ? o.cg + sum_over_lane_groups(pl[0], pl[1], pl[2], pl[3]) : 0.0;
        const double lgm = (ANTI && o.cv) ? o.cg + sum_over_lane_groups(plm[0], plm[1], plm[2], plm[3]) : 0.0;
        double p = basket_sample<ANTI>(basket, mirror, lg, lgm, o.strike, o.cv);
        const uint32_t i = i0 + lane;
        if (i >= w.n_units)
            p = 0;
        acc_s += p;
        acc_q = __builtin_fma(p, p, acc_q);
        if (out && i < w.n_units)
            out[i] = p;
    }
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// =========================================================================================
// CVA of one call.  Reference device loop, dp/MonteCarloKernel.cu:241-262 (spot advanced
// first, exposure = Black-Scholes value at the NEW spot and residual maturity :125-129, with
// the Hastings CDF :110-123).  Everything that depends only on the date j is tabulated once
// per call on the host in fp64 (the reference recomputes it per path per step, :248):
//   W_j = z_1 + ... + z_j                       (the lane's only state)
//   d1 = W_j g_j + e1_j,  d2 = W_j g_j + e2_j   g_j = v sqrt(dt) / (v sqrt(tau_j))
//   s_j = E(W_j bx + xk_j)                      ln s_j = ln S0 + j a + v sqrt(dt) W_j
//   ee_j = s_j cnd(d1) - disc_j cnd(d2)         disc_j = K exp(-r tau_j)
//   cva  = LGD * sum_j dp_j ee_j
// A final date with residual maturity exactly 0 uses the intrinsic value.
// =========================================================================================
template <class Real>
struct CvaStep {
    Real g, e1, e2, xk, disc, dp;
};
template <class Real>
struct CvaArgs {
    const CvaStep<Real> *steps;  // n_bs Black-Scholes dates (+1 intrinsic date if last_intrinsic)
    const float *pairs;          // fp32 only: the same rows for date pairs (2q, 2q + 1), field by field:
                                 // {g, g', e1, e1', e2, e2', xk, xk', disc, disc', dp, dp'}, floor(n_bs / 2) pairs
    int n_bs;                    // dates priced with the closed form
    int last_intrinsic;          // 1: one more date, residual maturity == 0
    Real bx;                     // v sqrt(dt) (times log2 e in f32)
    Real lgd, strike;
    const Real *extra;           // Greeks only: sqrt(tau_j) for every date, then sigma t_j for every date (2 (n_bs + last_intrinsic) reals)
    int pairs_in_lds;            // fp32 one-lane-per-path loop: the launch carries dynamic LDS for `pairs` (host: cva_enqueue), rows come as ds_read_b128
};

// Black-Scholes exposure at one date, from the lane's state W and the date's table row.
// Reference: device_bsCall + cnd, dp/MonteCarloKernel.cu:110-129.  With T(d) = phi(d) P(1/(1+c|d|))
// (Hastings 26.2.17, same constants) cnd(d) = d > 0 ? 1 - T(d) : T(d), and the Black-Scholes
// identity  S phi(d1) = K e^{-r tau} phi(d2)  lets ONE exponential serve both terms:
//     A = S phi(d1) = C exp(ln S - d1^2 / 2)
//     S cnd(d1) = d1 > 0 ? S - A P1 : A P1,      K e^{-r tau} cnd(d2) = d2 > 0 ? disc - A P2 : A P2
// -- evaluated without the selects by signed_tails below --
// (the reference evaluates three exponentials per date here: :106,:118 twice,:128).
// ln S is the value the spot's own exponential is taken of, so A costs one fma + one exponential.
// S cnd(d1) - K' cnd(d2) from the two upper tails t1 = S tail(|d1|), t2 = K' tail(|d2|), without selects:
//     cnd(d) = 1/2 + sgn(d) (1/2 - tail(|d|))   =>   (S - K')/2 + sgn(d1) (S/2 - t1) - sgn(d2) (K'/2 - t2)
// sgn(d) x is x with d's sign bit xor-ed into its (high) word: ONE v_bitop3_b32 (a ^ (b & c), truth table 0x78).
// fp64: 7 instructions per date instead of 9 (two subtractions, two compares, four v_cndmask, one subtraction): -1.3 ... -2.4 %
// kernel time on the 256-date CVA, in-process (profiles/r03_ab_cva_signed_tails.log); same bits as the select form up to
// the roundings of S/2 - t (per-path difference from the oracle unchanged: 1.5e-14 against 1.4e-14).
__device__ __forceinline__ float flip_by_sign(float x, float d)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, d), 0x80000000u, 0x78));
}
__device__ __forceinline__ double flip_by_sign(double x, double d)
{
    return __hiloint2double((int)__builtin_amdgcn_bitop3_b32((uint32_t)__double2hiint(x), (uint32_t)__double2hiint(d), 0x80000000u, 0x78),
                            __double2loint(x));
}
template <class Real>
__device__ __forceinline__ Real signed_tails(Real spot, Real disc, Real d1, Real d2, Real t1, Real t2)
{
    const Real w1 = flip_by_sign(fma_r((Real)0.5, spot, -t1), d1);
    const Real w2 = flip_by_sign(fma_r((Real)0.5, disc, -t2), d2);
    return fma_r((Real)0.5, spot - disc, w1 - w2);
}

__device__ __forceinline__ float bs_exposure(float ln2_spot, float W, const CvaStep<float> &st)
{
    const float spot = __builtin_amdgcn_exp2f(ln2_spot);
    // d1, d2 and everything downstream as one packed pair
    const f2 d = __builtin_elementwise_fma((f2){W, W}, (f2){st.g, st.g}, (f2){st.e1, st.e2});
    const float A = 0.3989422804014327f * __builtin_amdgcn_exp2f(__builtin_fmaf(d.x * -0.72134752044448170f, d.x, ln2_spot));
    const f2 den = __builtin_elementwise_fma((f2){0.2316419f, 0.2316419f}, __builtin_elementwise_abs(d), (f2){1.0f, 1.0f});
    const f2 k = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f2 poly = __builtin_elementwise_fma(k, (f2){1.330274429f, 1.330274429f}, (f2){-1.821255978f, -1.821255978f});
    poly = __builtin_elementwise_fma(k, poly, (f2){1.781477937f, 1.781477937f});
    poly = __builtin_elementwise_fma(k, poly, (f2){-0.356563782f, -0.356563782f});
    poly = __builtin_elementwise_fma(k, poly, (f2){0.31938153f, 0.31938153f});
    const f2 t = poly * k * (f2){A, A};
    return signed_tails(spot, st.disc, d.x, d.y, t.x, t.y);
}

// fp32, TWO consecutive dates of one path in the halves of every packed op (the form above packs d1, d2 of one
// date): the same operations per date, but the spot's exponent, A, the selects' subtractions and the dp-weighted
// accumulation now issue packed as well.  `row` = 12 floats {g, e1, e2, xk, disc, dp} x {date, next date}.
__device__ __forceinline__ f2 bs_exposure_dates(f2 ln2_spot, f2 W, f2 g, f2 e1, f2 e2, f2 disc)
{
    const f2 spot = {__builtin_amdgcn_exp2f(ln2_spot.x), __builtin_amdgcn_exp2f(ln2_spot.y)};
    const f2 d1 = __builtin_elementwise_fma(W, g, e1), d2 = __builtin_elementwise_fma(W, g, e2);
    const f2 a = __builtin_elementwise_fma(d1 * (f2){-0.72134752044448170f, -0.72134752044448170f}, d1, ln2_spot);
    const f2 A = (f2){0.3989422804014327f, 0.3989422804014327f} * (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f2 c = {0.2316419f, 0.2316419f}, one = {1.0f, 1.0f};
    const f2 den1 = __builtin_elementwise_fma(c, __builtin_elementwise_abs(d1), one);
    const f2 den2 = __builtin_elementwise_fma(c, __builtin_elementwise_abs(d2), one);
    const f2 k1 = {__builtin_amdgcn_rcpf(den1.x), __builtin_amdgcn_rcpf(den1.y)};
    const f2 k2 = {__builtin_amdgcn_rcpf(den2.x), __builtin_amdgcn_rcpf(den2.y)};
    const f2 c4 = {1.330274429f, 1.330274429f}, c3 = {-1.821255978f, -1.821255978f}, c2 = {1.781477937f, 1.781477937f},
             c1 = {-0.356563782f, -0.356563782f}, c0 = {0.31938153f, 0.31938153f};
    f2 p1 = __builtin_elementwise_fma(k1, c4, c3), p2 = __builtin_elementwise_fma(k2, c4, c3);
    p1 = __builtin_elementwise_fma(k1, p1, c2);
    p2 = __builtin_elementwise_fma(k2, p2, c2);
    p1 = __builtin_elementwise_fma(k1, p1, c1);
    p2 = __builtin_elementwise_fma(k2, p2, c1);
    p1 = __builtin_elementwise_fma(k1, p1, c0);
    p2 = __builtin_elementwise_fma(k2, p2, c0);
    const f2 t1 = p1 * k1 * A, t2 = p2 * k2 * A;
    // signed_tails, both dates packed: the sign flips are the only per-lane instructions
    const f2 half = {0.5f, 0.5f};
    const f2 v1 = __builtin_elementwise_fma(half, spot, -t1), v2 = __builtin_elementwise_fma(half, disc, -t2);
    const f2 w1 = {flip_by_sign(v1.x, d1.x), flip_by_sign(v1.y, d1.y)}, w2 = {flip_by_sign(v2.x, d2.x), flip_by_sign(v2.y, d2.y)};
    return __builtin_elementwise_fma(half, spot - disc, w1 - w2);
}
__device__ __forceinline__ f2 bs_exposure_dates(f2 ln2_spot, f2 W, const float *row)
{
    return bs_exposure_dates(ln2_spot, W, (f2){row[0], row[1]}, (f2){row[2], row[3]}, (f2){row[4], row[5]}, (f2){row[8], row[9]});
}

// a * b + c with c read from its SGPR pair by the three-operand instruction (c must be wave-uniform)
__device__ __forceinline__ double fma_scalar_addend(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
// a wave-uniform value in a vector register pair (one copy, where hipcc would make one per use)
__device__ __forceinline__ double to_vgpr(double x)
{
    asm("" : "+v"(x));
    return x;
}

__device__ __forceinline__ double hastings_poly(double k)
{
    double poly = __builtin_fma(k, 1.330274429, -1.821255978);
    poly = __builtin_fma(k, poly, 1.781477937);
    poly = __builtin_fma(k, poly, -0.356563782);
    poly = __builtin_fma(k, poly, 0.31938153);
    return poly * k;
}

// the same polynomial times 1/sqrt(2 pi): coefficients pre-multiplied (each rounded once, at compile time)
__device__ __forceinline__ double hastings_poly_phi(double k)
{
    constexpr double C = 0.39894228040143267793994605993438;
    double poly = __builtin_fma(k, C * 1.330274429, C * -1.821255978);
    poly = __builtin_fma(k, poly, C * 1.781477937);
    poly = __builtin_fma(k, poly, C * -0.356563782);
    poly = __builtin_fma(k, poly, C * 0.31938153);
    return poly * k;
}

__device__ __forceinline__ double bs_exposure(double ln_spot, double W, const CvaStep<double> &st)
{
    const double spot = exp_f64(ln_spot);
    const double d1 = __builtin_fma(W, st.g, st.e1), d2 = __builtin_fma(W, st.g, st.e2);
    const double A = 0.39894228040143267793994605993438 * exp_f64(fmax(__builtin_fma(-0.5 * d1, d1, ln_spot), -800.0));  // d1 runs away as tau -> 0
    double k1, k2;
    recip2_pos(__builtin_fma(0.2316419, fabs(d1), 1.0), __builtin_fma(0.2316419, fabs(d2), 1.0), k1, k2);
    const double t1 = A * hastings_poly(k1);
    const double t2 = A * hastings_poly(k2);
    return signed_tails(spot, st.disc, d1, d2, t1, t2);
}

// fp64: two consecutive dates of a path together, so that their four Hastings reciprocals share one v_rcp_f64 (a
// 16-cycle instruction: -3.7 % kernel time for sharing in pairs, a further -1.3 % for four).  Same operations per date
// as bs_exposure except for that shared reciprocal and 1/sqrt(2 pi) folded into the Hastings coefficients.
template <bool SGPR_ROWS = true>
__device__ __forceinline__ void bs_exposure2(double ln_a, double W_a, const CvaStep<double> &sa, double ln_b, double W_b,
                                             const CvaStep<double> &sb, double &ee_a, double &ee_b)
{
    const double spot_a = exp_f64(ln_a), spot_b = exp_f64(ln_b);
    // The table row sits in SGPRs and a VALU instruction reads at most one scalar operand.  hipcc turns fma(W, g, e)
    // with g and e both scalar into v_fmac_f64 and copies the ADDEND into the destination pair first (two v_mov_b32 per
    // fma: 6 of the ~124 instructions per date).  Spelled out instead: g into a vector pair once (it serves d1 and d2),
    // the addends read straight from their SGPRs by the three-operand form -- same fma, same bits, 3 instructions per
    // date instead of 6 (-3.1 % instructions, -2.4 % time: profiles/r02_ab_cva_scalar_addend.log).
    // (SGPR_ROWS = false: the date differs between lanes -- cva_dates_kernel -- and the rows are ordinary vector operands)
    double d1a, d2a, d1b, d2b;
    if constexpr (SGPR_ROWS) {
        const double g_a = scalar_to_vgpr(sa.g), g_b = scalar_to_vgpr(sb.g);
        d1a = fma_scalar_addend(W_a, g_a, sa.e1), d2a = fma_scalar_addend(W_a, g_a, sa.e2);
        d1b = fma_scalar_addend(W_b, g_b, sb.e1), d2b = fma_scalar_addend(W_b, g_b, sb.e2);
    } else {
        d1a = __builtin_fma(W_a, sa.g, sa.e1), d2a = __builtin_fma(W_a, sa.g, sa.e2);
        d1b = __builtin_fma(W_b, sb.g, sb.e1), d2b = __builtin_fma(W_b, sb.g, sb.e2);
    }
    // A = C exp(.) with C = 1/sqrt(2 pi) folded into the Hastings coefficients (hastings_poly_phi): one multiply less per date
    const double A_a = exp_f64(fmax(__builtin_fma(-0.5 * d1a, d1a, ln_a), -800.0));  // d1 runs away as tau -> 0
    const double A_b = exp_f64(fmax(__builtin_fma(-0.5 * d1b, d1b, ln_b), -800.0));
    double k1a, k2a, k1b, k2b;
    recip4_pos(__builtin_fma(0.2316419, fabs(d1a), 1.0), __builtin_fma(0.2316419, fabs(d2a), 1.0),
               __builtin_fma(0.2316419, fabs(d1b), 1.0), __builtin_fma(0.2316419, fabs(d2b), 1.0), k1a, k2a, k1b, k2b);
    const double t1a = A_a * hastings_poly_phi(k1a), t2a = A_a * hastings_poly_phi(k2a);
    const double t1b = A_b * hastings_poly_phi(k1b), t2b = A_b * hastings_poly_phi(k2b);
    ee_a = signed_tails(spot_a, sa.disc, d1a, d2a, t1a, t2a);
    ee_b = signed_tails(spot_b, sb.disc, d1b, d2b, t1b, t2b);
}

// One date the slow way: any date of a block that the pair forms below do not cover (a block's tail at the end of the
// grid, the intrinsic-value date).  j is wave-uniform: the table row comes through scalar loads.
template <class Real, bool ANTI>
__device__ __forceinline__ void cva_single_date(const CvaArgs<Real> &o, int j, int n_dates, Real z, Real &W, Real &acc)
{
    if (j >= n_dates)
        return;
    const CvaStep<Real> st = o.steps[j];
    W += z;
    const Real ln_spot = fma_r(W, o.bx, st.xk);     // natural log in f64, log2 in f32
    const Real ln_mirror = fma_r(-W, o.bx, st.xk);  // the path driven by -z (ANTI only)
    Real ee;
    if (j < o.n_bs) {
        ee = bs_exposure(ln_spot, W, st);
        if (ANTI)
            ee += bs_exposure(ln_mirror, -W, st);
    } else {
        const Real iv = exp_model(ln_spot) - o.strike;
        ee = iv > 0 ? iv : 0;
        if (ANTI) {
            const Real ivm = exp_model(ln_mirror) - o.strike;
            ee += ivm > 0 ? ivm : 0;
        }
    }
    acc = fma_r(st.dp, ee, acc);
}

// fp32: a block of four normals = two packed date pairs per trip of the date loop.
__device__ __forceinline__ f2 exposure_pair_rows(float bx, f2 Wp, const float (&row)[12]);   // below, next to the date-parallel role that shares it
// `lds_pairs`: the workgroup's LDS copy of o.pairs, or nullptr.  From scalar registers a pair's row costs six v_mov_b32 per pair (a
// packed fma takes ONE scalar operand, so every addend and second factor is copied into vector registers first: 12 of the 169
// instructions per four dates); from LDS the twelve floats arrive in vector registers as three ds_read_b128 at a wave-uniform
// address, which issue beside the VALU (44 instead of 49 vector instructions per pair; -2.5 % at 19 trips, -2.3 % at 1e7 paths in one
// process: profiles/r06_ab_cva_f32_rows_in_lds.log).  Same operations on the same values either way.
template <bool ANTI, class Gen>
__device__ __forceinline__ float cva_path(Gen &gen, const CvaArgs<float> &o, const Work &w, uint32_t c0, const float *lds_pairs = nullptr)
{
    constexpr int NPB = Gen::template npb<float>();
    static_assert(NPB == 4, "two packed date pairs per block");
    float W = 0, acc = 0;
    f2 acc2 = {0.0f, 0.0f};  // even / odd dates of the packed date pairs
    float z[NPB];
    const int n_dates = o.n_bs + o.last_intrinsic;
    for (int j0 = 0; j0 < n_dates; j0 += NPB) {
        gen.normals(w, c0, (uint32_t)(j0 / NPB), 3u /*MC_DOMAIN_CVA*/, z);
#pragma unroll
        for (int h = 0; h < NPB / 2; ++h) {
            const int j = j0 + 2 * h;
            if (j + 1 < o.n_bs) {  // wave-uniform: both dates of this pair have a closed-form exposure
                // the pair's rows once more, field by field ({g, g'} ... {dp, dp'} adjacent: mc_api.hip), so the two dates
                // ride in the halves of every packed instruction (-8 % against packing d1, d2 of one date)
                const f2 Wp = {W + z[2 * h], (W + z[2 * h]) + z[2 * h + 1]};
                W = Wp.y;
                if (lds_pairs) {   // wave-uniform
                    float row[12];
                    const float4 *src = reinterpret_cast<const float4 *>(lds_pairs + 12 * (j / 2));
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const float4 v = src[q];
                        row[4 * q] = v.x, row[4 * q + 1] = v.y, row[4 * q + 2] = v.z, row[4 * q + 3] = v.w;
                    }
                    f2 ee = exposure_pair_rows(o.bx, Wp, row);
                    if (ANTI)
                        ee += exposure_pair_rows(o.bx, -Wp, row);
                    acc2 = pk_fma((f2){row[10], row[11]}, ee, acc2);
                } else {
                    const float *row = o.pairs + 12 * (j / 2);
                    const f2 bx = {o.bx, o.bx}, xk = {row[6], row[7]}, dp = {row[10], row[11]};
                    f2 ee = bs_exposure_dates(pk_fma(Wp, bx, xk), Wp, row);
                    if (ANTI)
                        ee += bs_exposure_dates(pk_fma(-Wp, bx, xk), -Wp, row);
                    acc2 = pk_fma(dp, ee, acc2);
                }
            } else {
                cva_single_date<float, ANTI>(o, j, n_dates, z[2 * h], W, acc);
                cva_single_date<float, ANTI>(o, j + 1, n_dates, z[2 * h + 1], W, acc);
            }
        }
    }
    acc += acc2.x + acc2.y;
    return acc * (ANTI ? o.lgd * 0.5f : o.lgd);
}

// fp64: dates in Box-Muller pairs drawn through the generator's pair cursor (mc_rng.hpp): the fp64 stream hands out eight
// normals per block, and a date loop that draws all eight at once keeps four pairs of table rows in SGPRs (they spill into
// VGPR lanes) and eight normals in VGPRs.  The cursor draws one pair at a time; the loop runs FOUR pairs per trip with the
// cursor's phase a compile-time constant in each copy (round 6: 793 vector instructions per eight dates instead of 4 x 216,
// -3.4 % at C5's shard and -4.2 % at C5 in one process, profiles/r06_ab_cva_four_pairs.log; 129 VGPRs = 3 waves per SIMD,
// and forcing 4 with amdgpu_waves_per_eu changes nothing), and one pair per trip for what is left of the grid.
template <bool ANTI, class Gen>
__device__ __forceinline__ double cva_path(Gen &gen, const CvaArgs<double> &o, const Work &w, uint32_t c0)
{
    double W = 0, acc = 0;
    const int n_dates = o.n_bs + o.last_intrinsic;
    const double bx_v = to_vgpr(o.bx);   // in a vector register for the whole path: ln s = fma(W, bx, xk_j) then reads ONE scalar (xk_j)
    // the polynomial coefficients of the pair's Box-Muller as opaque vector registers: one three-operand v_fma_f64 per Horner step
    // instead of hipcc's v_mov + v_fmac (mc_math_f64.hpp: F64K; 9 of the 225 instructions per two dates -- the exponentials and
    // Hastings tails of the exposure gain nothing from the same treatment and stay on literals)
    F64K K;
    K.load();
    typename Gen::Carry carry;
    carry.K = &K;
    // dates j, j + 1 (both with a closed-form exposure) from one pair of normals
    auto closed_pair = [&](int j, double z0, double z1) {
        const CvaStep<double> sa = o.steps[j], sb = o.steps[j + 1];
        const double W_a = W + z0, W_b = W_a + z1;
        W = W_b;
        double ee_a, ee_b;
        bs_exposure2(fma_scalar_addend(W_a, bx_v, sa.xk), W_a, sa, fma_scalar_addend(W_b, bx_v, sb.xk), W_b, sb, ee_a, ee_b);
        if (ANTI) {
            double em_a, em_b;
            bs_exposure2(fma_scalar_addend(-W_a, bx_v, sa.xk), -W_a, sa, fma_scalar_addend(-W_b, bx_v, sb.xk), -W_b, sb, em_a, em_b);
            ee_a += em_a;
            ee_b += em_b;
        }
        acc = fma_r(sa.dp, ee_a, acc);
        acc = fma_r(sb.dp, ee_b, acc);
    };
    // FOUR pairs per trip while eight closed-form dates remain: P = 4g + {0, 1, 2, 3} makes the cursor's phase a compile-time
    // constant in each of the four copies -- no four-way switch, and the carried words are plain values instead of loop-carried
    // copies (the one-pair loop below spends 8 v_mov_b32 + 3 v_mov_b64 of its 216 instructions per trip on them).
    // (the Philox cursors only: a sequential or external stream has no phase switch, and four copies cost it 13 to 90 VGPRs)
    int j = 0;
    if constexpr (Gen::cursor_phases > 1) {
#pragma unroll 1
        for (; j + 8 <= o.n_bs; j += 8) {
            const uint32_t g4 = (uint32_t)(j >> 3) << 2;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                double z0, z1;
                gen.pair(w, c0, 3u /*MC_DOMAIN_CVA*/, g4 | k, carry, z0, z1);
                closed_pair(j + 2 * (int)k, z0, z1);
            }
        }
    }
#pragma unroll 1
    for (; j < n_dates; j += 2) {
        double z0, z1;
        gen.pair(w, c0, 3u /*MC_DOMAIN_CVA*/, (uint32_t)(j >> 1), carry, z0, z1);
        if (j + 1 < o.n_bs) {  // wave-uniform: both dates of this pair have a closed-form exposure
            closed_pair(j, z0, z1);
        } else {
            cva_single_date<double, ANTI>(o, j, n_dates, z0, W, acc);
            cva_single_date<double, ANTI>(o, j + 1, n_dates, z1, W, acc);
        }
    }
    gen.pairs_done((uint32_t)((n_dates + 1) >> 1));
    return acc * (ANTI ? o.lgd * 0.5 : o.lgd);
}

// one lane per path, grid-stride over the segment's paths: workgroup `block` of `n_blocks` (the whole grid for cva_kernel, the
// main part of it for cva_split_kernel)
// `lds_raw`: the launch's dynamic LDS.  fp32 with o.pairs_in_lds: the pair rows are staged there once per workgroup (all threads; a barrier)
template <class Real, bool ANTI, class Gen>
__device__ __forceinline__ void cva_paths_role(const CvaArgs<Real> &o, const Work &w, Real *__restrict__ out, uint32_t block, uint32_t n_blocks,
                                               unsigned char *lds_raw, double &acc_s, double &acc_q)
{
    const uint32_t stride = n_blocks * GROUP;
    const uint32_t gtid = block * GROUP + threadIdx.x;
    [[maybe_unused]] const float *lds_pairs = nullptr;
    if constexpr (sizeof(Real) == 4 && !ANTI) {   // (the antithetic form holds a row across two exposures: measured 0.5 % slower from LDS)
        if (o.pairs_in_lds) {   // uniform over the launch
            float *dst = reinterpret_cast<float *>(lds_raw);
            for (int k = threadIdx.x; k < 12 * (o.n_bs / 2); k += GROUP)
                dst[k] = o.pairs[k];
            __syncthreads();
            lds_pairs = dst;
        }
    }
    Gen gen(w);
    for (uint32_t i = gtid; i < w.n_units; i += stride) {
        Real p;
        if constexpr (sizeof(Real) == 4 && !ANTI)
            p = cva_path<ANTI>(gen, o, w, w.unit_lo + i, lds_pairs);
        else
            p = cva_path<ANTI>(gen, o, w, w.unit_lo + i);
        acc_s += (double)p;
        acc_q = __builtin_fma((double)p, (double)p, acc_q);
        if (out)  // wave-uniform: per-path dump for the parity tests
            out[i] = p;
    }
}

template <class Real, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void cva_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const CvaArgs<Real> o, const Work w, Real *__restrict__ out)
{
    stage_tables<Real>();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double acc_s = 0.0, acc_q = 0.0;
    cva_paths_role<Real, ANTI, Gen>(o, w, out, blockIdx.x, gridDim.x, lds_raw, acc_s, acc_q);
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// =========================================================================================
// CVA, parallel in the DATE axis.  Reference loop: dp/MonteCarloKernel.cu:241-262 -- one thread walks all N_GRID dates of
// its path.  With the reformulation above the lane's only state is W_j = z_1 + ... + z_j and everything else is a table
// row of the date, so the walk is a prefix sum followed by independent work: here a path's dates are shared by
// L = 2^log2_lanes ADJACENT lanes (L = 2 ... 64).  Per round of CH * L dates, lane `sub` of the path owns the CH dates
// [r0 + sub CH, r0 + (sub + 1) CH):
//   1. it draws ITS dates' normals -- the generator is counter-based (block = date / NPB): nothing is handed over --
//      and keeps them in registers (CH of them);
//   2. one DPP scan over the L lanes (row_shr inside rows of 16, row_bcast across rows: pure VALU) turns the chunk sums into every
//      lane's W offset;
//   3. it prices its CH dates with the same per-date operations as cva_kernel, the table row read per LANE from an LDS copy
//      of the table (the date is no longer wave-uniform, so the scalar loads of cva_kernel are not available);
// and after the last round one DPP reduction over the L lanes forms the path's sum_j dp_j ee_j BEFORE it is squared.  The
// values differ from cva_kernel's only by the association of two sums (W, and the sum over dates): 1e-16-level in fp64.
// What it is for (mc_api.hip: cva_enqueue picks L): a unit of work is 1/L of a path, so
//   * the LAST PARTIAL WAVE-TRIP of a large call (cva_kernel's time is a staircase in steps of 64 lanes x 1024 SIMDs = 65 536
//     paths: 1.25e6 paths -- C5's shard of 8 -- pay 20 trips for 19.07) is priced by date-parallel workgroups of the SAME launch
//     (cva_split_kernel below);
//   * a SMALL call (the reference driver's own 131 072 paths, dp/cvaOpt.cu:12-15: two waves per SIMD) fills the chip
//     (cva_dates_kernel).
// Generators: the counter-based ones and the external array (XORWOW is one sequence per lane: a path cannot be entered in
// the middle).  Grid-stride over path slots; a wave's lanes all run the same number of trips (lanes beyond the range price
// a valid path and contribute nothing), because the scans need every lane.
// =========================================================================================
__device__ __forceinline__ float lane_fetch(float v, uint32_t src_lane)   // v of lane `src_lane` (mod 64)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double lane_fetch(double v, uint32_t src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// DPP moves inside the wave (pure VALU: no LDS round trip on the scan's dependent chain).  Lanes without a source lane, or in a
// row masked off, receive 0.  Controls (gfx9 encoding): row_shr:n = 0x110 + n (shift right by n inside a row of 16),
// wave_shr:1 = 0x138 (the whole wave by one lane); the others are mc_reduce.hpp's.
constexpr int DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138;
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_fetch(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// Inclusive scan of `v` over aligned groups of L = 2^k <= 64 adjacent lanes (`sub` = lane & (L - 1), L wave-uniform): first inside
// the rows of 16 (row_shr 1, 2, 4, 8), then row totals across rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
template <class Real>
__device__ __forceinline__ Real group_scan_incl(Real v, uint32_t L, uint32_t sub)
{
    const uint32_t r = sub & 15u;   // position inside the row of 16
    if (L > 1) { const Real t = dpp_fetch<DPP_ROW_SHR + 1>(v); v += r >= 1u ? t : (Real)0; }
    if (L > 2) { const Real t = dpp_fetch<DPP_ROW_SHR + 2>(v); v += r >= 2u ? t : (Real)0; }
    if (L > 4) { const Real t = dpp_fetch<DPP_ROW_SHR + 4>(v); v += r >= 4u ? t : (Real)0; }
    if (L > 8) { const Real t = dpp_fetch<DPP_ROW_SHR + 8>(v); v += r >= 8u ? t : (Real)0; }
    if (L > 16) { const Real t = dpp_fetch<DPP_ROW_BCAST15, 0xA>(v); v += (sub & 16u) ? t : (Real)0; }
    if (L > 32) { const Real t = dpp_fetch<DPP_ROW_BCAST31, 0xC>(v); v += (sub & 32u) ? t : (Real)0; }
    return v;
}
// Sum of `v` over the same groups, valid in the group's LAST lane (sub == L - 1; for L <= 16 in every lane of the group).
template <class Real>
__device__ __forceinline__ Real group_total_in_last_lane(Real v, uint32_t L)
{
    if (L > 1) v += dpp_fetch<DPP_QUAD_XOR1>(v);
    if (L > 2) v += dpp_fetch<DPP_QUAD_XOR2>(v);
    if (L > 4) v += dpp_fetch<DPP_ROW_HALF_MIRROR>(v);
    if (L > 8) v += dpp_fetch<DPP_ROW_MIRROR>(v);
    if (L > 16) v += dpp_fetch<DPP_ROW_BCAST15, 0xA>(v);
    if (L > 32) v += dpp_fetch<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// exposures of two consecutive dates of one path, table rows in vector registers; same operations per date as the pair
// forms of cva_path (fp32: the two dates in the halves of every packed instruction; fp64: one shared reciprocal)
// fp32: `row` = the pair's 12 floats {g, g', e1, e1', e2, e2', xk, xk', disc, disc', dp, dp'} (date, next date) read per lane from the
// LDS copy -- adjacent registers are ready-made packed operands (two separate 6-float rows made the compiler transpose them
// through scratch memory: 28 bytes per lane and twice the time of the one-lane-per-path kernel)
__device__ __forceinline__ f2 exposure_pair_rows(float bx, f2 Wp, const float (&row)[12])
{
    return bs_exposure_dates(pk_fma(Wp, (f2){bx, bx}, (f2){row[6], row[7]}), Wp, (f2){row[0], row[1]}, (f2){row[2], row[3]}, (f2){row[4], row[5]},
                             (f2){row[8], row[9]});
}
__device__ __forceinline__ void exposure_pair_rows(double bx, double W_a, double W_b, const CvaStep<double> &sa, const CvaStep<double> &sb,
                                                   double &ee_a, double &ee_b)
{
    bs_exposure2<false>(__builtin_fma(W_a, bx, sa.xk), W_a, sa, __builtin_fma(W_b, bx, sb.xk), W_b, sb, ee_a, ee_b);
}
template <class Real>
__device__ __forceinline__ Real intrinsic_exposure(Real bx, Real W, Real xk, Real strike)
{
    const Real iv = exp_model(fma_r(W, bx, xk)) - strike;
    return iv > 0 ? iv : (Real)0;
}

// workgroup `block` of `n_blocks` date-parallel workgroups; lds_raw: room for the per-date table (n_dates rows)
template <class Real, int CH, bool ANTI, class Gen>
__device__ __forceinline__ void cva_dates_role(const CvaArgs<Real> &o, const Work &w, const uint32_t log2_lanes, Real *__restrict__ out, uint32_t block,
                                               uint32_t n_blocks, unsigned char *lds_raw, double &acc_s, double &acc_q)
{
    constexpr int NPB = Gen::template npb<Real>();
    static_assert(CH % NPB == 0 && CH % 2 == 0, "a lane's chunk is whole generator blocks and whole date pairs");
    CvaStep<Real> *rows = reinterpret_cast<CvaStep<Real> *>(lds_raw);
    const int n_dates = o.n_bs + o.last_intrinsic;
    {   // the per-date table, once per workgroup (n_dates * 6 reals, rounded up to whole date pairs; the host checked that it fits).
        // fp64: the rows as they are.  fp32: two dates per row, field by field -- {g, g', e1, e1', ...} -- the packed operands of a pair
        const Real *src = reinterpret_cast<const Real *>(o.steps);
        Real *dst = reinterpret_cast<Real *>(lds_raw);
        for (int k = threadIdx.x; k < 6 * (n_dates + (n_dates & 1)); k += GROUP) {
            const int date = k / 6, field = k - 6 * date;
            const Real x = date < n_dates ? src[k] : (Real)0;
            if constexpr (sizeof(Real) == 4)
                dst[12 * (date >> 1) + 2 * field + (date & 1)] = x;
            else
                dst[k] = x;
        }
    }
    __syncthreads();
    const uint32_t L = 1u << log2_lanes, lane = threadIdx.x & 63u, sub = lane & (L - 1u);
    const uint32_t slots = (uint32_t)GROUP >> log2_lanes;                  // paths a workgroup holds at a time
    const uint32_t stride = n_blocks * slots;
    const uint32_t slot = block * slots + (threadIdx.x >> log2_lanes);
    Gen gen(w);
    for (uint32_t i = slot, i0 = __builtin_amdgcn_readfirstlane(slot); i0 < w.n_units; i += stride, i0 += stride) {   // i0: the wave's first slot
        const bool live = i < w.n_units;
        const uint32_t unit = w.unit_lo + (live ? i : i0);
        Real W_done = 0, acc = 0;   // W_done: the path's W at the end of the previous round
        f2 acc2 = {0.0f, 0.0f};     // fp32: even / odd dates of the packed pairs
        for (int r0 = 0; r0 < n_dates; r0 += CH << log2_lanes) {
            const int j0 = r0 + (int)sub * CH;   // this lane's first date (0-based) of the round: a multiple of CH, so every pair starts on an even date
            Real z[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t)
                z[t] = 0;
            if (j0 < n_dates) {
#pragma unroll
                for (int q = 0; q < CH / NPB; ++q) {
                    Real zz[NPB];
                    gen.normals(w, unit, (uint32_t)(j0 / NPB + q), 3u /*MC_DOMAIN_CVA*/, zz);
#pragma unroll
                    for (int t = 0; t < NPB; ++t)
                        z[q * NPB + t] = (j0 + q * NPB + t < n_dates) ? zz[t] : (Real)0;
                }
            }
            Real incl = z[0];
#pragma unroll
            for (int t = 1; t < CH; ++t)
                incl += z[t];
            incl = group_scan_incl(incl, L, sub);                    // inclusive scan of the chunk sums over the path's L lanes (DPP)
            const Real before = dpp_fetch<DPP_WAVE_SHR1>(incl);      // the lane below's inclusive sum
            Real Wl = sub ? W_done + before : W_done;                 // W at the end of the date before this lane's chunk
            W_done += lane_fetch(incl, lane | (L - 1u));              // the round's total, for the NEXT round: off the dependent chain
#pragma unroll
            for (int t = 0; t < CH; t += 2) {
                const int ja = j0 + t, jb = ja + 1;
                const Real W_a = Wl + z[t], W_b = W_a + z[t + 1];
                Wl = W_b;
                const bool intrinsic_here = o.last_intrinsic && (ja == o.n_bs || jb == o.n_bs);   // one lane of one wave of the path
                if constexpr (sizeof(Real) == 4) {
                    const int pair = (ja < n_dates ? ja : n_dates - 1) >> 1;
                    float row[12];
                    const float *pr = reinterpret_cast<const float *>(lds_raw) + 12 * pair;
#pragma unroll
                    for (int f = 0; f < 12; ++f)
                        row[f] = pr[f];
                    const f2 Wp = {W_a, W_b};
                    f2 ee = exposure_pair_rows(o.bx, Wp, row);
                    if (ANTI)
                        ee += exposure_pair_rows(o.bx, -Wp, row);
                    ee.x = ja < o.n_bs ? ee.x : 0.0f;   // beyond the closed-form dates: the intrinsic-value date below, or nothing
                    ee.y = jb < o.n_bs ? ee.y : 0.0f;
                    if (intrinsic_here) {
                        const bool first = ja == o.n_bs;
                        const float Wi = first ? W_a : W_b, xk = first ? row[6] : row[7];
                        float iv = intrinsic_exposure(o.bx, Wi, xk, o.strike);
                        if (ANTI)
                            iv += intrinsic_exposure(o.bx, -Wi, xk, o.strike);
                        ee.x = first ? iv : ee.x;
                        ee.y = first ? ee.y : iv;
                    }
                    const f2 dp = {ja < n_dates ? row[10] : 0.0f, jb < n_dates ? row[11] : 0.0f};
                    acc2 = pk_fma(dp, ee, acc2);
                } else {
                    const CvaStep<Real> sa = rows[ja < n_dates ? ja : n_dates - 1], sb = rows[jb < n_dates ? jb : n_dates - 1];
                    Real ee_a, ee_b;
                    exposure_pair_rows(o.bx, W_a, W_b, sa, sb, ee_a, ee_b);
                    if (ANTI) {
                        Real em_a, em_b;
                        exposure_pair_rows(o.bx, -W_a, -W_b, sa, sb, em_a, em_b);
                        ee_a += em_a, ee_b += em_b;
                    }
                    ee_a = ja < o.n_bs ? ee_a : (Real)0;   // beyond the closed-form dates: the intrinsic-value date below, or nothing
                    ee_b = jb < o.n_bs ? ee_b : (Real)0;
                    if (intrinsic_here) {
                        const bool first = ja == o.n_bs;
                        const Real Wi = first ? W_a : W_b, xk = first ? sa.xk : sb.xk;
                        Real iv = intrinsic_exposure(o.bx, Wi, xk, o.strike);
                        if (ANTI)
                            iv += intrinsic_exposure(o.bx, -Wi, xk, o.strike);
                        ee_a = first ? iv : ee_a;   // (selects: a reference picked at run time would put both in scratch)
                        ee_b = first ? ee_b : iv;
                    }
                    acc = fma_r(sa.dp, ee_a, acc);
                    acc = fma_r(sb.dp, ee_b, acc);
                }
            }
        }
        if constexpr (sizeof(Real) == 4)
            acc += acc2.x + acc2.y;
        acc = group_total_in_last_lane(acc, L);   // the path's sum over its lanes, in the group's last lane
        const Real p = acc * (ANTI ? o.lgd * (Real)0.5 : o.lgd);
        if (live && sub == L - 1u) {
            acc_s += (double)p;
            acc_q = __builtin_fma((double)p, (double)p, acc_q);
            if (out)
                out[i] = p;
        }
    }
}

template <class Real, int CH, bool ANTI, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void cva_dates_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const CvaArgs<Real> o, const Work w,
                                                          const uint32_t log2_lanes, Real *__restrict__ out)
{
    stage_tables<Real>();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double acc_s = 0.0, acc_q = 0.0;
    cva_dates_role<Real, CH, ANTI, Gen>(o, w, log2_lanes, out, blockIdx.x, gridDim.x, lds_raw, acc_s, acc_q);
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}

// ONE launch for a call that ends in a partial wave-trip: the first `tail_groups` workgroups price the call's last `wt.n_units`
// paths date-parallel, the others the leading whole trips one lane per path.  (Two launches would have to run side by side:
// on one stream the queue's barrier bit serialises them -- gfx950 ignores hipExtAnyOrderLaunch, profiles/r06_any_order_probe.log
// -- and a second stream forked from and joined to the first costs 21-26 us per call whatever the remainder's size,
// half of the trip it saves: profiles/r06_shard_clock_split_two_streams.log.)  The date-parallel workgroups come FIRST in
// the grid: they are dispatched with the first round of the others and are long gone when the last full trip ends.  The
// kernel's registers are the date-parallel role's (fp64: 4 waves per SIMD instead of cva_kernel's 6, which costs cva_kernel's
// loop nothing measurable: profiles/r06_occupancy_probe.log).
template <class Real, int CH, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void cva_split_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const CvaArgs<Real> o, const Work w,
                                                          const Work wt, const uint32_t tail_groups, const uint32_t log2_lanes,
                                                          Real *__restrict__ out, Real *__restrict__ out_tail)
{
    stage_tables<Real>();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double acc_s = 0.0, acc_q = 0.0;
    if (blockIdx.x < tail_groups)   // workgroup-uniform
        cva_dates_role<Real, CH, false, Gen>(o, wt, log2_lanes, out_tail, blockIdx.x, tail_groups, lds_raw, acc_s, acc_q);
    else
        cva_paths_role<Real, false, Gen>(o, w, out, blockIdx.x - tail_groups, gridDim.x - tail_groups, lds_raw, acc_s, acc_q);
    group_sum2(acc_s, acc_q);
    finish_group(acc_s, acc_q);
}


// =========================================================================================
// Greeks of the basket call (the reference prices only).  Per path, in the reference's own (unfolded) device
// formulas dp/MonteCarloKernel.cu:74-101 on the pricing kernels' normals:
//     bt_a = sum_{b<=a} L_ab g_b + d_a,   s_a = S_a exp(mu_a + v_a bt_a sqrt T),   B = sum_a w_a s_a,   I = [B > K]
// PATHWISE (LR = false):
//     payoff = I (B - K),   d payoff / d S_a = I w_a s_a / S_a,   d payoff / d v_a = I w_a s_a (bt_a sqrt T - v_a T)
// LIKELIHOOD RATIO (LR = true): the payoff times the score of the terminal prices' joint lognormal density.  With
// x = ln S(T) ~ N(m, Sigma), Sigma = T D L L' D, D = diag(v), the whitened vector y = L^-T g (y_a = sum_{b>=a} M_ab g_b,
// M = L^-T on the host) gives Sigma^-1 (x - m) = D^-1 y / sqrt T and
//     score_{S_a} = y_a / (S_a v_a sqrt T),
//     score_{v_a} = (y_a (L g)_a - 1) / v_a + (sqrt T d_a - v_a T) y_a / (v_a sqrt T)
// (one asset: z / (S sigma sqrt T) and (z^2 - 1) / sigma - z sqrt T, the vanilla kernel's).  No indicator is differentiated;
// needs a non-singular factor (every L_aa > 0).
// ONE PASS per BASKET_GREEKS_CHUNK assets: a lane keeps the (sum, sum2) pairs of the price and of the two derivatives of the
// A = 8 assets of its workgroup's chunk (grid y index) in registers -- 2 + 4 A doubles -- so a basket of n assets is simulated
// ceil(n / 8) times instead of n times (round 5: one pass per asset; n = 16: profiles/r06_basket_greeks_one_pass.log).  Generic
// in n (normals in the lane's LDS column, constants through scalar loads).  Planes of the pair buffer:
// 0 = price (published by y = 0 only), 1 + a = delta_a, 1 + n + a = vega_a.
// Constant table (Real): L[n*n] row-major | d[n] | mu[n] | v[n] | w[n] | s[n] | inv_s[n] | vt[n] (= v_a T)
//                        | LR only: M[n*n] | inv_svt[n] (= 1 / (S_a v_a sqrt T)) | inv_v[n] | mcoef[n] (= (sqrt T d_a - v_a T) / (v_a sqrt T)).
// =========================================================================================
template <class Real>
struct BasketGreeks {
    const Real *consts;
    int n;
    Real strike, sqrt_t;
};
#ifndef MC_BASKET_GREEKS_CHUNK
#define MC_BASKET_GREEKS_CHUNK 8   // -DMC_BASKET_GREEKS_CHUNK=1 rebuilds round 5's one pass per asset (the A/B of profiles/r06_basket_greeks_one_pass.log)
#endif
constexpr int BASKET_GREEKS_CHUNK = MC_BASKET_GREEKS_CHUNK;

__device__ __forceinline__ float exp_nat(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ double exp_nat(double x) { return exp_f64(x); }

template <class Real, bool LR>
__global__ __launch_bounds__(GROUP) void basket_greeks_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const BasketGreeks<Real> o, const Work w)
{
    constexpr int A = BASKET_GREEKS_CHUNK;
    stage_tables<Real>();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Real *g = reinterpret_cast<Real *>(lds_raw) + threadIdx.x;  // this lane's column, stride GROUP
    constexpr int NPB = GenPhilox::npb<Real>();
    const int n = o.n, nblk = (n + NPB - 1) / NPB, chunk = blockIdx.y, a0 = chunk * A;
    GenPhilox gen(w);
    typedef const __attribute__((address_space(4))) Real *cptr;
    const cptr L = (cptr)o.consts, d = L + n * n, mu = d + n, v = mu + n, wt = v + n, s0 = wt + n, inv_s = s0 + n, vt = inv_s + n;
    const cptr M = vt + n, inv_svt = M + n * n, inv_v = inv_svt + n, mcoef = inv_v + n;   // LR only
    const uint32_t stride = gridDim.x * GROUP;
    double acc[2 + 4 * A];
#pragma unroll
    for (int q = 0; q < 2 + 4 * A; ++q)
        acc[q] = 0;
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += stride) {
        for (int b = 0; b < nblk; ++b) {
            Real z[NPB];
            gen.normals(w, w.unit_lo + i, (uint32_t)b, 2u /*MC_DOMAIN_BASKET*/, z);
#pragma unroll
            for (int j = 0; j < NPB; ++j)
                g[(b * NPB + j) * GROUP] = z[j];
        }
        Real basket = 0, term[A], bts[A];
#pragma unroll
        for (int k = 0; k < A; ++k)
            term[k] = bts[k] = 0;
        for (int c = 0; c * A < n; ++c) {
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const int a = c * A + k;
                if (a < n) {   // workgroup-uniform
                    Real bt = 0;
                    for (int b = 0; b <= a; ++b)
                        bt = fma_r(L[a * n + b], g[b * GROUP], bt);
                    const Real bt0 = bt;   // (L g)_a, before the drift
                    bt += d[a];
                    const Real t_a = s0[a] * exp_nat(fma_r(v[a] * bt, o.sqrt_t, mu[a])) * wt[a];
                    basket += t_a;
                    if (c == chunk) {   // workgroup-uniform: the assets whose derivatives this workgroup accumulates
                        term[k] = t_a;
                        bts[k] = LR ? bt0 : bt;
                    }
                }
            }
        }
        const bool itm = basket > o.strike;
        const Real payoff = itm ? basket - o.strike : (Real)0;
        const double pay = (double)payoff;
        acc[0] += pay, acc[1] = __builtin_fma(pay, pay, acc[1]);
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const int a = a0 + k;
            if (a < n) {   // workgroup-uniform
                double dl, vg;
                if constexpr (LR) {
                    Real y = 0;
                    for (int b = a; b < n; ++b)
                        y = fma_r(M[a * n + b], g[b * GROUP], y);
                    dl = (double)(payoff * (y * inv_svt[a]));
                    vg = (double)(payoff * fma_r(fma_r(y, bts[k], (Real)-1), inv_v[a], mcoef[a] * y));
                } else {
                    dl = itm ? (double)(term[k] * inv_s[a]) : 0.0;
                    vg = itm ? (double)(term[k] * (bts[k] * o.sqrt_t - vt[a])) : 0.0;
                }
                acc[2 + 4 * k] += dl, acc[3 + 4 * k] = __builtin_fma(dl, dl, acc[3 + 4 * k]);
                acc[4 + 4 * k] += vg, acc[5 + 4 * k] = __builtin_fma(vg, vg, acc[5 + 4 * k]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 1 + 2 * A; ++q)
        group_sum2(acc[2 * q], acc[2 * q + 1]);
    const tail_ptr t = late_tail(acc[0]);
    if (chunk == 0)
        publish_pair(t, 0, acc[0], acc[1]);
#pragma unroll
    for (int k = 0; k < A; ++k)
        if (a0 + k < n) {
            publish_pair(t, 1 + a0 + k, acc[2 + 4 * k], acc[3 + 4 * k]);
            publish_pair(t, 1 + n + a0 + k, acc[4 + 4 * k], acc[5 + 4 * k]);
        }
    arrive_and_finish(t);
}

// =========================================================================================
// CVA with its pathwise delta.  CVA = LGD sum_j dp_j C(S_j, tau_j) and S_j is proportional to S_0, so
//     d CVA / d S_0 = LGD sum_j dp_j Delta_j S_j / S_0,   Delta_j = cnd(d1_j)   (I[S_j > K] on an intrinsic-value date)
// and S_j cnd(d1_j) is the first term of the exposure the pricing kernel computes anyway.  Vega: sigma moves the closed
// form (Black-Scholes vega S phi(d1) sqrt(tau) = A sqrt(tau), A being the shared exponential of bs_exposure) and the
// path (d S_j / d sigma = S_j (W_j sqrt(dt) - sigma t_j)):
//     d CVA / d sigma = LGD sum_j dp_j [ A_j sqrt(tau_j) + S_j cnd(d1_j) (W_j sqrt(dt) - sigma t_j) ]
// Plain estimator, one date at a time (a secondary kernel); planes of the pair buffer: 0 = CVA, 1 = delta, 2 = vega.
// =========================================================================================
__device__ __forceinline__ void bs_exposure_delta(float ln2_spot, float W, const CvaStep<float> &st, float &ee, float &s_delta, float &s_phi)
{
    const float spot = __builtin_amdgcn_exp2f(ln2_spot);
    const float d1 = __builtin_fmaf(W, st.g, st.e1), d2 = __builtin_fmaf(W, st.g, st.e2);
    const float A = 0.3989422804014327f * __builtin_amdgcn_exp2f(__builtin_fmaf(d1 * -0.72134752044448170f, d1, ln2_spot));
    const float k1 = __builtin_amdgcn_rcpf(__builtin_fmaf(0.2316419f, fabsf(d1), 1.0f));
    const float k2 = __builtin_amdgcn_rcpf(__builtin_fmaf(0.2316419f, fabsf(d2), 1.0f));
    const float t1 = A * (k1 * (0.31938153f + k1 * (-0.356563782f + k1 * (1.781477937f + k1 * (-1.821255978f + k1 * 1.330274429f)))));
    const float t2 = A * (k2 * (0.31938153f + k2 * (-0.356563782f + k2 * (1.781477937f + k2 * (-1.821255978f + k2 * 1.330274429f)))));
    s_delta = d1 > 0 ? spot - t1 : t1;            // S cnd(d1)
    ee = s_delta - (d2 > 0 ? st.disc - t2 : t2);  // - K e^{-r tau} cnd(d2)
    s_phi = A;                                    // S phi(d1)
}
__device__ __forceinline__ void bs_exposure_delta(double ln_spot, double W, const CvaStep<double> &st, double &ee, double &s_delta, double &s_phi)
{
    const double spot = exp_f64(ln_spot);
    const double d1 = __builtin_fma(W, st.g, st.e1), d2 = __builtin_fma(W, st.g, st.e2);
    const double A = 0.39894228040143267793994605993438 * exp_f64(fmax(__builtin_fma(-0.5 * d1, d1, ln_spot), -800.0));
    double k1, k2;
    recip2_pos(__builtin_fma(0.2316419, fabs(d1), 1.0), __builtin_fma(0.2316419, fabs(d2), 1.0), k1, k2);
    const double t1 = A * hastings_poly(k1), t2 = A * hastings_poly(k2);
    s_delta = d1 > 0 ? spot - t1 : t1;
    ee = s_delta - (d2 > 0 ? st.disc - t2 : t2);
    s_phi = A;
}

// LR = true: the likelihood-ratio forms on the same stream.  Only the first transition's density depends on S_0, every
// transition's on sigma, and the closed-form exposure depends on sigma explicitly (its own vega stays pathwise):
//     d CVA / d S_0   = E[ CVA_path z_1 ] / (S_0 sigma sqrt(dt))
//     d CVA / d sigma = E[ LGD sum_j dp_j S_j phi(d1_j) sqrt(tau_j)  +  CVA_path sum_j ((z_j^2 - 1) / sigma - z_j sqrt(dt)) ]
// (the sums over the dates that draw a normal).  lr_delta = 1 / (S_0 sigma sqrt(dt)), inv_sigma = 1 / sigma.
template <class Real, bool LR>
__global__ __launch_bounds__(GROUP) void cva_greeks_kernel(const Tail /* first argument, read late: mc_reduce.hpp */, const CvaArgs<Real> o, const Work w, Real inv_spot,
                                                           Real sqrt_dt, Real lr_delta, Real inv_sigma)
{
    stage_tables<Real>();
    constexpr int NPB = GenPhilox::npb<Real>();
    const uint32_t stride = gridDim.x * GROUP;
    const int n_dates = o.n_bs + o.last_intrinsic;
    GenPhilox gen(w);
    typedef const __attribute__((address_space(4))) Real *cptr;
    const cptr sqrt_tau = (cptr)o.extra, sig_t = sqrt_tau + n_dates;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += stride) {
        Real W = 0, cva = 0, delta = 0, vega = 0, score = 0, z_first = 0, z[NPB];
        for (int j = 0; j < n_dates; ++j) {   // wave-uniform: table rows through scalar loads
            if (j % NPB == 0)
                gen.normals(w, w.unit_lo + i, (uint32_t)(j / NPB), 3u /*MC_DOMAIN_CVA*/, z);
            const CvaStep<Real> st = o.steps[j];
            Real zz = z[0];
#pragma unroll
            for (int q = 1; q < NPB; ++q)
                zz = (j % NPB == q) ? z[q] : zz;
            W += zz;
            if (LR) {
                z_first = j == 0 ? zz : z_first;
                score += fma_r(fma_r(zz, zz, (Real)-1), inv_sigma, -zz * sqrt_dt);
            }
            const Real ln_spot = fma_r(W, o.bx, st.xk);
            Real ee, sd, sphi;
            if (j < o.n_bs) {
                bs_exposure_delta(ln_spot, W, st, ee, sd, sphi);
            } else {   // residual maturity exactly 0: intrinsic value, derivative I[S > K] S, no closed-form vega
                const Real spot = exp_model(ln_spot);
                const bool itm = spot > o.strike;
                ee = itm ? spot - o.strike : (Real)0;
                sd = itm ? spot : (Real)0;
                sphi = 0;
            }
            cva = fma_r(st.dp, ee, cva);
            if (LR) {
                vega = fma_r(st.dp, sphi * sqrt_tau[j], vega);
            } else {
                delta = fma_r(st.dp, sd, delta);
                vega = fma_r(st.dp, fma_r(sphi, sqrt_tau[j], sd * fma_r(W, sqrt_dt, -sig_t[j])), vega);
            }
        }
        const Real cva_path = cva * o.lgd;
        const double c = (double)cva_path;
        const double dl = LR ? (double)(cva_path * (z_first * lr_delta)) : (double)(delta * o.lgd * inv_spot);
        const double vg = LR ? (double)fma_r(cva_path, score, vega * o.lgd) : (double)(vega * o.lgd);
        acc[0] += c, acc[1] = __builtin_fma(c, c, acc[1]);
        acc[2] += dl, acc[3] = __builtin_fma(dl, dl, acc[3]);
        acc[4] += vg, acc[5] = __builtin_fma(vg, vg, acc[5]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        group_sum2(acc[2 * q], acc[2 * q + 1]);
    const tail_ptr t = late_tail(acc[0]);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        publish_pair(t, q, acc[2 * q], acc[2 * q + 1]);
    arrive_and_finish(t);
}

// =========================================================================================
// Normals dump (parity tests of the generator itself): Gen::npb<Real>() normals per unit, unit-major.
// =========================================================================================
template <class Real, class Gen = GenPhilox>
__global__ __launch_bounds__(GROUP) void normals_kernel(const Work w, uint32_t block, uint32_t domain,
                                                        Real *__restrict__ out)
{
    stage_tables<Real>();
    constexpr int NPB = Gen::template npb<Real>();
    const uint32_t stride = gridDim.x * GROUP;
    Gen gen(w);
    for (uint32_t i = blockIdx.x * GROUP + threadIdx.x; i < w.n_units; i += stride) {
        Real z[NPB];
        gen.normals(w, w.unit_lo + i, block, domain, z);
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            out[(uint64_t)i * NPB + j] = z[j];
    }
}

}  // namespace mc
