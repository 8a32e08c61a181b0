// mc_rng.hpp -- counter-based random normals for gfx950 (device code).
//
// Replaces the reference's cuRAND XORWOW state array (dp/MonteCarloKernel.cu:285-290 set-up
// kernel, 48 B of state read per thread at :143,189,232): there is NO generator state in HBM.
// A path's normals are a pure function of (seed, counter):
//
//     Philox4x32-10( counter = {unit_hi, unit_lo, block, domain}, key = {seed_lo, seed_hi} )
//
// (unit = 64-bit index of the unit of work; why the LOW word sits in counter word 1: philox_unit below.)
//
// One Philox block gives four 32-bit words = 4 normals in f32 (one word per uniform); in f64 three Philox blocks
// give 8 normals (96 bits per Box-Muller pair: words_to_normals below), by two-branch Box-Muller.  The oracle twin of
// this file is oracle/mc_oracle_impl.h:orc_dev_normals (tests compare them word for word).
//
// WHERE a kernel's normals come from is a policy class (the `Gen` template parameter of every simulation kernel,
// "generator policies" below): Philox (default), XORWOW (the reference's generator), Philox with fp32 normals widened
// to double (the reference's own dp arithmetic, opt-in), or an array in HBM (the from-normals test hooks; the launch-geometry compatibility mode, mc_grid.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mc_math_f64.hpp"

namespace mc {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

struct u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c in ONE VALU instruction: gfx950's v_bitop3_b32 evaluates any 3-input boolean
// function from an 8-bit truth table (0x96 = three-way xor).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Ten rounds.  The key schedule is wave-uniform (seed is a kernel argument) and lives in
// SGPRs; each round is two v_mad_u64_u32 (full 64-bit product: hi and lo in one instruction)
// and two v_bitop3_b32 (hi ^ counter ^ key in one instruction instead of two v_xor_b32: on
// MI355X every VALU instruction next to the multiplies costs a full 4-cycle issue slot, so
// halving the xors takes a fifth off the generator).  Counter words that are wave-uniform
// (unit_hi, block, domain) let the compiler move the first rounds' work to the scalar unit (philox_unit).
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return {c0, c1, c2, c3};
}

// The engine's counter layout: {unit_hi, unit_lo, block, domain}.  Only unit_lo differs between the lanes of
// a wave.  Word 1 is not multiplied in the first round (it is only xor-ed into the new word 0), so with the
// lane-varying word there, both of round 1's multiplies and one each of rounds 2 and 3 act on wave-uniform
// values and run on the scalar unit: 16 v_mad_u64_u32 + 18 v_bitop3_b32 per block instead of 19 + 19 with the
// index in word 0 (vanilla fp32 -4.6 % time, in-process A/B).  Any placement is an equally good stream --
// Philox is a bijection of the 128-bit counter for each key.
__device__ __forceinline__ u32x4 philox_unit(uint32_t unit_lo, uint32_t unit_hi, uint32_t block, uint32_t domain,
                                             uint32_t k0, uint32_t k1)
{
    return philox4x32_10(unit_hi, unit_lo, block, domain, k0, k1);
}

// ---- f32 ---------------------------------------------------------------------------------
// u = x * 2^-32 + 2^-33  in (0, 1]   (one v_cvt_f32_u32 + one v_fma_f32)
__device__ __forceinline__ float u01_f32(uint32_t x)
{
    return __builtin_fmaf((float)x, 0x1p-32f, 0x1p-33f);
}

// Angle uniform, in REVOLUTIONS, for v_sin_f32 / v_cos_f32: the top 23 bits of the word as the mantissa of a
// float in [1, 2) -- one v_alignbit_b32, no conversion and no scaling.  sin and cos are periodic in whole
// revolutions, so 1 + f and f give the same point of the circle; f = (x >> 9) 2^-23 takes 2^23 equally spaced
// angles.  (The radius uniform keeps the (0, 1] form above: it goes through a logarithm.)
__device__ __forceinline__ float angle_f32(uint32_t x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_alignbit(0x7fu, x, 9));
}

// Box-Muller on hardware transcendentals: v_log_f32 is log2, v_sin/v_cos_f32 take their
// argument in revolutions, so 2*pi never appears.  `scale2` multiplies the squared radius:
//   scale2 = -2 ln2        ->  plain N(0,1) pair
//   scale2 = -2 ln2 * b^2  ->  pair already multiplied by b (saves the multiply per normal)
__device__ __forceinline__ void box_muller_f32(uint32_t xa, uint32_t xb, float scale2, float &z_cos,
                                               float &z_sin)
{
    const float ua = u01_f32(xa);
    const float ub = angle_f32(xb);
    const float radius = __builtin_amdgcn_sqrtf(scale2 * __builtin_amdgcn_logf(ua));
    z_cos = radius * __builtin_amdgcn_cosf(ub);
    z_sin = radius * __builtin_amdgcn_sinf(ub);
}

constexpr float NEG_2LN2_F32 = -1.3862943611198906f;

// ---- f64 ---------------------------------------------------------------------------------
// 52-bit uniform strictly inside (0,1): ((hi:lo >> 12) + 0.5) * 2^-52, exact in double.
// Built from bits: [1,2) mantissa fill, then one exact add of -(1 - 2^-53)
// (two v_alignbit + v_add_f64 instead of two int->double conversions and two fmas).
__device__ __forceinline__ double u01_f64(uint32_t lo, uint32_t hi)
{
    const uint32_t mant_lo = __builtin_amdgcn_alignbit(hi, lo, 12);
    const uint32_t mant_hi = __builtin_amdgcn_alignbit(0x3ffu, hi, 12);  // (hi >> 12) | 0x3ff00000 in one instruction
    return __hiloint2double((int)mant_hi, (int)mant_lo) + (-1.0 + 0x1p-53);
}

// Kernels that draw fp64 normals call stage_tables<double>() first (LDS tables of mc_math_f64.hpp).
__device__ __forceinline__ void box_muller_f64(const u32x4 r, double &z_cos, double &z_sin)
{
    const double radius = sqrt_pos(neg2log_unit_tab(u01_f64(r.x, r.y)));
    double s, c;
    sincos_turns_tab(r.z, r.w, s, c);  // angle 2*pi*u01_f64(r.z, r.w)
    z_cos = radius * c;
    z_sin = radius * s;
}
template <class Real> __device__ __forceinline__ void stage_tables()
{
    if constexpr (sizeof(Real) == 8)
        stage_f64_tables();
}

// All normals of one fp32 block: four words, one word per uniform
__device__ __forceinline__ void words_to_normals(const u32x4 r, float (&out)[4])
{
    box_muller_f32(r.x, r.y, NEG_2LN2_F32, out[0], out[1]);
    box_muller_f32(r.z, r.w, NEG_2LN2_F32, out[2], out[3]);
}

// fp64 (stream version 2, MC_STREAM_VERSION in mc_mi355x.h): a "block" of the fp64 stream is THREE consecutive Philox
// blocks = 12 words = four Box-Muller pairs of 96 bits each, EIGHT normals (version 1 spent one Philox block of 128 bits
// per pair and used 104 of them: 8 normals cost 4 Philox blocks, 35 VALU instructions each).  Pair p takes words
// (a, m, c) = W[3p], W[3p + 1], W[3p + 2]:
//     radius uniform   52 bits: (a : top 20 bits of m)                     u_a = (J + 1/2) 2^-52
//     angle uniform    44 bits: (c : low 12 bits of m), as the top 44 bits of a 52-bit fraction with 8 zero bits below:
//                               u_b = (J' 2^8 + 1/2) 2^-52 -- no bit is shared with the radius
// so that both go through the same table-driven forms as before; the only extra instruction per pair is the shift that
// moves m's low 12 bits into place.  z_{2p} = r cos 2 pi u_b, z_{2p+1} = r sin 2 pi u_b.
__device__ __forceinline__ void pair_normals_f64(uint32_t a, uint32_t m, uint32_t c, double &z_cos, double &z_sin)
{
    box_muller_f64((u32x4){m, a, m << 20, c}, z_cos, z_sin);   // (lo, hi) of the radius, (lo, hi) of the angle
}
// the same pair on register-resident coefficients (mc_math_f64.hpp: F64K): same operations, same bits
__device__ __forceinline__ void pair_normals_f64(uint32_t a, uint32_t m, uint32_t c, const F64K &K, double &z_cos, double &z_sin)
{
    const double radius = sqrt_pos(neg2log_unit_tab(u01_f64(m, a), K));
    double s, co;
    sincos_turns_tab(m << 20, c, K, s, co);
    z_cos = radius * co;
    z_sin = radius * s;
}
__device__ __forceinline__ void words_to_normals(const u32x4 r0, const u32x4 r1, const u32x4 r2, double (&out)[8])
{
    pair_normals_f64(r0.x, r0.y, r0.z, out[0], out[1]);
    pair_normals_f64(r0.w, r1.x, r1.y, out[2], out[3]);
    pair_normals_f64(r1.z, r1.w, r2.x, out[4], out[5]);
    pair_normals_f64(r2.y, r2.z, r2.w, out[6], out[7]);
}

// ---- the unit of work a launch strides over (host side: mc_api.hip make_work) ----------------------------
// `Work` describes one segment: units [unit_lo, unit_lo + n_units) with a common high word
// (the host splits a range so that unit_lo + n_units <= 2^32 and n_units <= 2^31; only the low
// word differs between lanes, which moves part of Philox's first rounds to the scalar unit: philox_unit).
struct Work {
    uint32_t seed_lo, seed_hi;  // Philox key
    uint32_t unit_lo, unit_hi;  // first unit of the segment (64-bit counter, split)
    uint32_t n_units;           // units in the segment
    // path window for masked launches (vanilla edges, per-path dumps): a path is live iff
    // first_path <= p < end_path.  Ignored by the unmasked kernels.
    uint64_t first_path, end_path;
    const uint32_t *xorwow;     // GenXorwow only: start states of the launch's lanes, 6 words each
    const void *ext;            // GenExternal only: normals of the segment's units, `ext_per_unit` Reals per unit
    uint32_t ext_per_unit;
};

typedef float f2 __attribute__((ext_vector_type(2)));  // one VGPR pair: v_pk_{fma,mul,add}_f32 operands

// ---- generator policies -------------------------------------------------------------------------------
// A policy object lives in a lane's registers for the whole launch.  Interface:
//   Gen(const Work &)                                   per-lane set-up (XORWOW: load the lane's state)
//   npb<Real>()                                         normals one "block" yields in that precision
//   normals(w, unit_lo, block, domain, Real (&z)[npb])  the normals of block `block` of unit `unit_lo`
//   words(w, unit_lo, block, domain)                    the block's four raw words (word-level kernels: the packed-fp32
//                                                       kernels build two paths' normals from words themselves);
//                                                       absent when `external`
//   external                                            normals come from memory, there are no words
//
// GenPhilox: the engine's generator.  Stateless: block (unit, block, domain) is a pure function of the counter.
//
// GenXorwow: the reference's generator (cuRAND XORWOW, dp/MonteCarloKernel.cu:285-290 curand_init, :68,78,250
// curand_normal) as a SECOND, selectable generator (mc_context_set_generator).  Marsaglia's xorwow:
// 160 bits of xorshift state + a Weyl word, ~9 integer instructions per 32-bit word.  One sequence per LANE of the
// launch, exactly like the reference's one curandState per thread: lane l starts from rocRAND's
// rocrand_init(seed, subsequence = base + l, offset 0) -- the subsequences are 2^67 words apart -- and draws four
// words whenever a kernel asks for a "block", in the order it asks (the counter arguments are ignored).  The start
// states come from a context-owned array in HBM (24 B per lane, read once per launch: the reference reads 48 B per
// thread, :189), filled by xorwow_init_kernel below when (seed, base) change; nothing is written back, so a call is
// reproducible -- but, unlike Philox, WHICH normals a path gets depends on the launch geometry.
//
// GenPhiloxF32N (fp64 kernels only, opt-in: mc_context_set_normals): the reference's own "double precision" --
// `double z = curand_normal(...)`, a FLOAT normal widened to double (dp/MonteCarloKernel.cu:68,78,250).
// One Philox block then yields FOUR normals through the hardware transcendentals of the fp32 path instead of two
// through mc_math_f64.hpp; everything downstream of the normal stays fp64.  Its own stream layout (4 normals per
// block in fp64), mirrored by the oracle (orc_set_normals_f32).
//
// GenExternal (mc_*_from_normals_*, mc_*_run_grid_*): the normals are read from an array in HBM, `ext_per_unit`
// Reals per unit, so that the reference's own normal stream (glibc rand() + Box-Muller, MonteCarloHost.c:117-121) can be
// pushed through the very payoff / accumulation / final-reduction code of the hot kernels and compared with the
// compiled reference's outputs (tests/test_gpu_from_normals.py) -- and so that the launch-geometry compatibility mode can
// price the sample a (numBlocks x numThreads) launch of the reference draws (mc_grid.hpp writes it in path order).
// Indices beyond a unit's count read as 0.
struct GenPhilox {
    static constexpr bool external = false;
    // the fp64 pair cursor below has a four-phase state: a date loop that draws FOUR pairs per trip makes the phase a compile-time
    // constant (mc_kernels.hpp: cva_path<double>); a cursor without phases (cursor_phases = 1) only pays registers for it
    static constexpr int cursor_phases = 4;
    template <class Real> static constexpr int npb() { return sizeof(Real) == 4 ? 4 : 8; }
    __device__ __forceinline__ explicit GenPhilox(const Work &) {}
    __device__ __forceinline__ u32x4 words(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain)
    {
        return philox_unit(unit_lo, w.unit_hi, block, domain, w.seed_lo, w.seed_hi);
    }
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain, float (&z)[4])
    {
        words_to_normals(words(w, unit_lo, block, domain), z);
    }
    // fp64: block b of the stream = Philox blocks 3b, 3b + 1, 3b + 2 (a kernel that uses only some of the eight normals
    // pays only for the Philox blocks those need: the rest is dead code after unrolling)
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain, double (&z)[8])
    {
        words_to_normals(words(w, unit_lo, 3u * block, domain), words(w, unit_lo, 3u * block + 1u, domain),
                         words(w, unit_lo, 3u * block + 2u, domain), z);
    }
    // fp64 PAIR CURSOR: the same stream one Box-Muller pair at a time, for a kernel that consumes a unit's pairs strictly in
    // order (the CVA date loop: pair P = dates 2P + 1, 2P + 2) and cannot afford eight normals and twelve words live at once.
    // P (wave-uniform) = 0, 1, 2, ... ; the words a Philox block yields beyond the current pair wait in `carry`:
    // pairs 4b .. 4b + 3 cost Philox blocks 3b, 3b + 1, 3b + 2 and nothing.
    // `K`: the caller's register-resident coefficients (the CVA date loop binds them once per kernel), or nullptr for the literal forms
    struct Carry {
        uint32_t c0, c1, c2;
        const F64K *K = nullptr;
    };
    __device__ __forceinline__ void pair(const Work &w, uint32_t unit_lo, uint32_t domain, uint32_t P, Carry &k, double &z0, double &z1)
    {
        const uint32_t b3 = 3u * (P >> 2);
        uint32_t a, m, c;
        switch (P & 3u) {
        case 0: { const u32x4 r = words(w, unit_lo, b3, domain); a = r.x, m = r.y, c = r.z, k.c0 = r.w; break; }
        case 1: { const u32x4 r = words(w, unit_lo, b3 + 1u, domain); a = k.c0, m = r.x, c = r.y, k.c0 = r.z, k.c1 = r.w; break; }
        case 2: { const u32x4 r = words(w, unit_lo, b3 + 2u, domain); a = k.c0, m = k.c1, c = r.x, k.c0 = r.y, k.c1 = r.z, k.c2 = r.w; break; }
        default: a = k.c0, m = k.c1, c = k.c2; break;
        }
        if (k.K)   // compile-time after inlining: the caller either bound the coefficients or did not
            pair_normals_f64(a, m, c, *k.K, z0, z1);
        else
            pair_normals_f64(a, m, c, z0, z1);
    }
    // after the last pair of a unit (`pairs` of them were drawn): nothing to do for a counter-based stream
    __device__ __forceinline__ void pairs_done(uint32_t) {}
};

struct GenXorwow {
    static constexpr bool external = false;
    static constexpr int cursor_phases = 1;
    template <class Real> static constexpr int npb() { return sizeof(Real) == 4 ? 4 : 8; }
    uint32_t x0, x1, x2, x3, x4, d;
    __device__ __forceinline__ explicit GenXorwow(const uint32_t *states)
    {
        const uint32_t *p = states + 6u * (blockIdx.x * blockDim.x + threadIdx.x);
        x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3], x4 = p[4], d = p[5];
    }
    __device__ __forceinline__ explicit GenXorwow(const Work &w) : GenXorwow(w.xorwow) {}
    struct Row {};   // tag: `p` already points at this lane's own six words (the launch-geometry kernels index by (block, thread))
    __device__ __forceinline__ GenXorwow(const uint32_t *p, Row) { x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3], x4 = p[4], d = p[5]; }
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t t = x0 ^ (x0 >> 2);
        x0 = x1, x1 = x2, x2 = x3, x3 = x4;
        x4 = xor3(x4, x4 << 4, t ^ (t << 1));
        d += 362437u;
        return d + x4;
    }
    __device__ __forceinline__ u32x4 words(const Work &, uint32_t, uint32_t, uint32_t)
    {
        u32x4 r;
        r.x = next(), r.y = next(), r.z = next(), r.w = next();
        return r;
    }
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain, float (&z)[4])
    {
        words_to_normals(words(w, unit_lo, block, domain), z);
    }
    // fp64: the lane's next TWELVE words (always the whole block, used or not: the sequence must move on the same way)
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain, double (&z)[8])
    {
        const u32x4 r0 = words(w, unit_lo, block, domain), r1 = words(w, unit_lo, block, domain), r2 = words(w, unit_lo, block, domain);
        words_to_normals(r0, r1, r2, z);
    }
    struct Carry { const F64K *K = nullptr; };
    __device__ __forceinline__ void pair(const Work &, uint32_t, uint32_t, uint32_t, Carry &, double &z0, double &z1)
    {
        const uint32_t a = next(), m = next(), c = next();   // the next three words: twelve per four pairs, as normals() draws them
        pair_normals_f64(a, m, c, z0, z1);
    }
    // a unit always consumes whole blocks of twelve words, like normals(): skip the pairs of the last block that were not drawn
    __device__ __forceinline__ void pairs_done(uint32_t pairs)
    {
        for (uint32_t p = pairs; (p & 3u) != 0; ++p) {
            (void)next();
            (void)next();
            (void)next();
        }
    }
};

struct GenPhiloxF32N : GenPhilox {
    static constexpr int cursor_phases = 2;   // one Philox block per two pairs
    template <class Real> static constexpr int npb() { return 4; }
    __device__ __forceinline__ explicit GenPhiloxF32N(const Work &w) : GenPhilox(w) {}
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain, double (&z)[4])
    {
        float f[4];
        words_to_normals(words(w, unit_lo, block, domain), f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            z[j] = (double)f[j];   // the widening of `double z = curand_normal(...)`
    }
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t domain, float (&z)[4])
    {
        words_to_normals(words(w, unit_lo, block, domain), z);
    }
    struct Carry { float z2, z3; const F64K *K = nullptr; };
    __device__ __forceinline__ void pair(const Work &w, uint32_t unit_lo, uint32_t domain, uint32_t P, Carry &k, double &z0, double &z1)
    {
        if ((P & 1u) == 0) {
            float f[4];
            words_to_normals(words(w, unit_lo, P >> 1, domain), f);
            z0 = (double)f[0], z1 = (double)f[1], k.z2 = f[2], k.z3 = f[3];
        } else {
            z0 = (double)k.z2, z1 = (double)k.z3;
        }
    }
    __device__ __forceinline__ void pairs_done(uint32_t) {}
};

struct GenExternal {
    static constexpr bool external = true;
    static constexpr int cursor_phases = 1;
    template <class Real> static constexpr int npb() { return sizeof(Real) == 4 ? 4 : 8; }   // the native block sizes
    __device__ __forceinline__ explicit GenExternal(const Work &) {}
    template <class Real, int N>
    __device__ __forceinline__ void normals(const Work &w, uint32_t unit_lo, uint32_t block, uint32_t, Real (&z)[N])
    {
        const Real *p = static_cast<const Real *>(w.ext) + (size_t)(unit_lo - w.unit_lo) * w.ext_per_unit;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint32_t idx = block * (uint32_t)N + (uint32_t)j;
            z[j] = idx < w.ext_per_unit ? p[idx] : (Real)0;
        }
    }
    struct Carry { const F64K *K = nullptr; };
    __device__ __forceinline__ void pair(const Work &w, uint32_t unit_lo, uint32_t, uint32_t P, Carry &, double &z0, double &z1)
    {
        const double *p = static_cast<const double *>(w.ext) + (size_t)(unit_lo - w.unit_lo) * w.ext_per_unit;
        z0 = 2u * P < w.ext_per_unit ? p[2u * P] : 0.0;
        z1 = 2u * P + 1u < w.ext_per_unit ? p[2u * P + 1u] : 0.0;
    }
    __device__ __forceinline__ void pairs_done(uint32_t) {}
};

// Start states of XORWOW lanes: the seeded state (5 xorshift words + Weyl word) jumped ahead by sub * 2^67 steps.
// jump[i] = A^(2^67 * 2^i) over GF(2), A = one xorshift step, as 160 columns of 5 words (column c = image of state
// bit c); the host computes them by repeated squaring (mc_api.hip).  A jump is one conditional xor of a column per set
// state bit, per set bit of the subsequence number.
constexpr int XORWOW_JUMP_BITS = 48;   // subsequence numbers below 2^48
// ... followed in the same table by XORWOW_OFFSET_BITS matrices A^(2^i), i = 0 .. 31: a jump by `offset` STEPS inside a
// sequence (the launch-geometry kernels split a reference thread's stream into sub-streams, mc_grid.hpp).
constexpr int XORWOW_OFFSET_BITS = 32;
constexpr int XORWOW_JUMP_MATRICES = XORWOW_JUMP_BITS + XORWOW_OFFSET_BITS;
// v <- (product of table[i] over the set bits i < bits of `value`) v
__device__ inline void xorwow_apply_jumps(uint32_t (&v)[5], uint64_t value, int bits, const uint32_t *__restrict__ table)
{
    for (int i = 0; i < bits; ++i) {
        if (!((value >> i) & 1u))
            continue;
        const uint32_t *m = table + (size_t)i * 160 * 5;
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (int c = 0; c < 160; ++c) {
            const uint32_t mask = 0u - ((v[c >> 5] >> (c & 31)) & 1u);
#pragma unroll
            for (int k = 0; k < 5; ++k)
                r[k] ^= mask & m[c * 5 + k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k)
            v[k] = r[k];
    }
}
__device__ inline void xorwow_jump(uint32_t (&v)[5], uint64_t sub, const uint32_t *__restrict__ jump)
{
    xorwow_apply_jumps(v, sub, XORWOW_JUMP_BITS, jump);
}
// `steps` words further in the same sequence: the xorshift words by the offset matrices, the Weyl word by steps * 362437
__device__ inline void xorwow_skip(uint32_t (&v)[5], uint32_t &weyl, uint32_t steps, const uint32_t *__restrict__ jump)
{
    xorwow_apply_jumps(v, steps, XORWOW_OFFSET_BITS, jump + (size_t)XORWOW_JUMP_BITS * 160 * 5);
    weyl += 362437u * steps;
}

// rocRAND's seeding (rocrand_xorwow.h, xorwow_engine constructor): fixed start words scrambled with the seed halves
__host__ __device__ inline void xorwow_seed(uint64_t seed, uint32_t (&x)[5], uint32_t &weyl)
{
    const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fu, s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbu;
    const uint32_t t0 = 1228688033u * s0, t1 = 2073658381u * s1;
    x[0] = 123456789u + t0, x[1] = 362436069u ^ t0, x[2] = 521288629u + t1, x[3] = 88675123u ^ t1, x[4] = 5783321u + t0;
    weyl = 6615241u + t1 + t0;
}

// lanes [0, lanes) of ONE seed: lane l = subsequence base + l
__global__ __launch_bounds__(256) void xorwow_init_kernel(const uint32_t *__restrict__ jump, uint64_t seed, uint64_t base, uint32_t lanes,
                                                          uint32_t *__restrict__ states)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes)
        return;
    uint32_t v[5], weyl;
    xorwow_seed(seed, v, weyl);
    xorwow_jump(v, base + lane, jump);
#pragma unroll
    for (int k = 0; k < 5; ++k)
        states[6 * (size_t)lane + k] = v[k];
    states[6 * (size_t)lane + 5] = weyl;
}

// `count` consecutive words of each of XORWOW lanes [0, lanes): the generator alone, for the word-for-word
// comparison with rocRAND's engine (tests).  out[lane * count + k].
__global__ __launch_bounds__(256) void xorwow_words_kernel(const uint32_t *__restrict__ states, uint32_t lanes, uint32_t count,
                                                           uint32_t *__restrict__ out)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes)
        return;
    GenXorwow rng(states);
    for (uint32_t k = 0; k < count; ++k)
        out[(size_t)lane * count + k] = rng.next();
}

}  // namespace mc
