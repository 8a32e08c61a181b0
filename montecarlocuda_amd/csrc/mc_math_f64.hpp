// mc_math_f64.hpp -- double-precision elementary functions cut to the Monte Carlo kernels' domains.
//
// Why not the library ones: on gfx950 every fp64 VALU instruction costs a 4-cycle issue slot and
// the kernels are VALU-issue-bound, so instruction count is time.  The general
// ocml routines pay for double-double intermediates and for arguments these kernels never see
// (measured VALU instructions: log 98, sincospi 70, sqrt 22, 1/x 12, exp 23 + ~10 v_mov_b64).  The versions below
// assume what the generator guarantees -- a uniform strictly inside (0,1), a positive finite radicand -- and stay
// within 1-2 ulp of the correctly rounded result (checked against glibc through the oracle's normals:
// tests/test_gpu_parity.py, bound 2e-14 absolute on |z| < 8.3; measured 2.9e-15).  LDS reads do not occupy the VALU,
// so the three hot functions trade polynomial length for one LDS read each:
//
//   sqrt_pos(x)          ~7 instructions   v_rsq_f64 + one coupled Newton step + one residual step (no rescaling)
//   recip_pos(d)         ~5                v_rcp_f64 + two Newton steps (recip2_pos / recip4_pos: several from one v_rcp)
//   exp_f64(x)           ~13 + ds_read     256-entry table of 2^(j/256), Taylor to r^4
//   neg2log_unit_tab(u)  ~16 + ds_read     128-entry table over the reduced mantissa: -2 ln m = t_i + S(m c2_i + 2),
//                                          S a 7-term series on |r'| <= 2^-7
//   sincos_turns_tab     ~18 + ds_read     256-entry table of (sin, cos) at the slot centres, angle addition with
//                                          2-/3-term series on |b| <= 2 pi / 512
// tools/check_f64_tables.c (host twin, vs 80-bit libm over 4e7 inputs): -2 ln u within 2.8e-16 relative,
// sin/cos within 1.7e-16 absolute, the normal within 1.8e-15 absolute, exp within 1.09 ulp.
//
// History of the variants that lost their in-process A/B runs (polynomial-only log 37 / sincos 36 / exp 19 instructions,
// ocml exp, coefficients in constant memory, two v_rcp_f64 instead of one shared, a second sqrt residual step, the
// three-operation sincos combine): DESIGN.md 4.1 and profiles/r01_ab_f64_*.log, r02_ab_exp256.log.  They are no longer
// in the tree; tools/build_ab.sh builds a variant from a patch when one is needed again.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

// ---- lookup tables (LDS) --------------------------------------------------------------------------
struct alignas(16) F64Pair { double x, y; };
// rows 0..127: {c2_i, t_i} of the log table; rows 128..383: {sin, cos} at the centre of angle slot i;
// rows 384..511: 2^(j/256), j = 0..255, two per row
// (tools/gen_f64_tables.py; 60-digit arithmetic, rounded once)
__device__ const F64Pair F64_TABLES_ROM[512] = {
#include "mc_tables_f64.inc"
};
__shared__ F64Pair f64_tables[512];  // 8 KB of LDS per workgroup, filled by stage_f64_tables()

// Every fp64 kernel calls this first (all threads; ends in a barrier).
__device__ __forceinline__ void stage_f64_tables()
{
    for (int i = threadIdx.x; i < 512; i += blockDim.x)
        f64_tables[i] = F64_TABLES_ROM[i];
    __syncthreads();
}

// e^x for finite x (underflows to 0 through v_ldexp_f64; these kernels never overflow it).
// Table-driven: x = (256 e + j) ln2/256 + r, |r| <= ln2/512;  e^x = 2^e * T_j * (1 + r + r^2/2 + r^3/6 + r^4/24).
// 13 instructions + one 8-byte LDS read; the dropped term r^5/120 is below 4e-17 relative.  (Round 1 and most of round 2:
// 64 entries and one more Horner step; the 2 KB table buys one fp64 instruction per exponential -- 16 per 16-asset basket
// path, 2 per CVA date: profiles/r02_ab_exp256.log.)  Max error: tools/check_f64_tables.c.
// PRECONDITION of exp_f64: |x| < EXP_F64_ARG_LIMIT.  n = rint(256 x / ln 2) must fit 31 bits (|x| < 5.8e6); the host-side
// range guards (mc_api.hip) and this comment share the one constant.
constexpr double EXP_F64_ARG_LIMIT = 5.0e6;
__device__ __forceinline__ double exp_f64(double x)
{
    // n = rint(256 x / ln 2) by the 1.5 * 2^52 trick: after the fma the integer sits in the low mantissa bits,
    // so the int conversion is a register read and the rounding is the fma's own.  Beyond EXP_F64_ARG_LIMIT the low
    // word is garbage and so is the result -- callers whose argument can run away (CVA's exp(-d1^2/2) next to
    // maturity) clamp it first.
    const double shifted = __builtin_fma(x, 369.32993046757462751, 0x1.8p52);
    const double n = shifted - 0x1.8p52;
    double r = __builtin_fma(n, -6.93147180369123816490e-01 / 256, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10 / 256, r);
    const int ni = __double2loint(shifted);
    const double T = reinterpret_cast<const double *>(f64_tables + 384)[ni & 255];
    double p = __builtin_fma(r, 1.0 / 24, 1.0 / 6);
    p = __builtin_fma(r, p, 0.5);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_amdgcn_ldexp(__builtin_fma(T, r * p, T), ni >> 8);
}

__device__ __forceinline__ double recip_pos(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}

// 1/a and 1/b from ONE v_rcp_f64 (a 16-cycle instruction): r = 1/(a b), 1/a = r b, 1/b = r a.
// For operands whose product stays far from over/underflow (CVA: both in [1, 3]).
__device__ __forceinline__ void recip2_pos(double a, double b, double &ra, double &rb)
{
    const double r = recip_pos(a * b);
    ra = r * b;
    rb = r * a;
}

// Four reciprocals from ONE v_rcp_f64: r = 1/(a1 a2 b1 b2), then products (14 instructions for four results
// instead of 16 for two pairs, and one 16-cycle v_rcp_f64 instead of two).  Operands in [1, 3]: no range issue.
__device__ __forceinline__ void recip4_pos(double a1, double a2, double b1, double b2, double &ra1, double &ra2, double &rb1,
                                           double &rb2)
{
    const double pa = a1 * a2, pb = b1 * b2;
    const double r = recip_pos(pa * pb);
    const double qa = r * pb, qb = r * pa;  // 1/(a1 a2), 1/(b1 b2)
    ra1 = qa * a2;
    ra2 = qa * a1;
    rb1 = qb * b2;
    rb2 = qb * b1;
}

// sqrt of a positive normal double (no rescaling: the callers' radicands lie in (1e-16, 1e3))
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// ---- table-driven forms of the Box-Muller pieces -------------------------------------------------
// -2 ln(u) for a positive normal double.
// u = 2^k m, m in [sqrt(1/2), sqrt(2)) (the exponent boundary is shifted by subtracting the bit pattern of sqrt(1/2)
// from the high word); slot i = top 7 bits of the shifted mantissa;
// r' = m c2_i + 2 = -2 (m c_i - 1), |r'| <= 2^-7;  -2 ln m = t_i + r' + r'^2/4 + r'^3/12 + ... + r'^7/448.
// The slot that contains m = 1 has c = 1, t = 0, so u -> 1 keeps its relative accuracy.
__device__ __forceinline__ double neg2log_unit_tab(double u)
{
    const int h = __double2hiint(u) - 0x3fe6a09e;
    const int k = h >> 20;
    const double m = __hiloint2double(__double2hiint(u) - (k << 20), __double2loint(u));
    const F64Pair e = f64_tables[(h >> 13) & 0x7f];
    const double r = __builtin_fma(m, e.x, 2.0);
    double p = __builtin_fma(r, 1.0 / 448, 1.0 / 192);
    p = __builtin_fma(r, p, 1.0 / 80);
    p = __builtin_fma(r, p, 1.0 / 32);
    p = __builtin_fma(r, p, 1.0 / 12);
    p = __builtin_fma(r, p, 0.25);
    const double small = __builtin_fma(r * r, p, r);
    const double big = __builtin_fma((double)k, -1.3862943611198906188, e.y);
    return big + small;
}

// (sin, cos) of 2 pi u for the 52-bit uniform u = (J + 1/2) 2^-52 given by its Philox words (J = hi:lo >> 12).
// Slot i = top 8 bits of J with centre a_i = 2 pi (i + 1/2)/256; b = 2 pi u - a_i, |b| <= 2 pi/512, is formed
// exactly up to one rounding from the low 44 bits; then sin(a+b), cos(a+b) by angle addition.
__device__ __forceinline__ void sincos_turns_tab(uint32_t lo, uint32_t hi, double &sin_out, double &cos_out)
{
    const F64Pair e = f64_tables[128 + (hi >> 24)];
    const uint32_t mant_lo = __builtin_amdgcn_alignbit(hi, lo, 12);
    const uint32_t mant_hi = ((hi >> 12) & 0xfffu) | 0x3ff00000u;
    const double y = __hiloint2double((int)mant_hi, (int)mant_lo) - (1.0 + 0x1p-9);   // (J mod 2^44) 2^-52 - 2^-9
    const double b = __builtin_fma(y, 6.283185307179586477, 6.283185307179586477 * 0x1p-53);
    const double z = b * b;
    const double ps = __builtin_fma(z, 1.0 / 120, -1.0 / 6);   // sin b = b + b z ps; the next term is < 1e-17
    const double sb = __builtin_fma(z * b, ps, b);       // sin b
    double pc = __builtin_fma(z, -1.0 / 720, 1.0 / 24);
    pc = __builtin_fma(z, pc, -0.5);
    const double cm = z * pc;                            // cos b - 1
    // two fmas each (max error 1.65e-16: tools/check_f64_tables.c)
    sin_out = __builtin_fma(e.y, sb, __builtin_fma(e.x, cm, e.x));
    cos_out = __builtin_fma(-e.x, sb, __builtin_fma(e.y, cm, e.y));
}


// ---- the same functions on REGISTER-RESIDENT constants ------------------------------------------------------------
// For a kernel whose scalar register file is full (the CVA date loop: two table rows, the Philox key schedule, ...) hipcc keeps the
// polynomial coefficients in vector registers, shares the equal halves of different coefficients (1/192, 1/12, 1/6, 1/24 all end in
// 0x55555555) and re-assembles a register pair before every use that needs the coefficient as the ADDEND of a two-address v_fmac_f64:
// v_mov_b32 + v_mov_b64 + v_fmac_f64 for one Horner step.  28 of the 225 VALU instructions per two fp64 CVA dates were such moves.
// F64K holds the coefficients that appear as addends (and the first-step multipliers next to them) as OPAQUE 64-bit vector registers
// (an empty asm the compiler cannot see through, executed once per kernel) and fma3() is ONE three-operand v_fma_f64 on them.  Same
// operations in the same order as the functions above: bit-identical results (tests/test_gpu_parity.py compares both with the oracle,
// tests/test_gpu_cva_dates.py the two kernels with each other).  Only the Box-Muller pieces: the exponentials and Hastings tails of the
// CVA exposure showed no gain from the same treatment (their remaining moves are table rows coming from scalar registers).
__device__ __forceinline__ double k_vgpr(double x)
{
    asm("" : "+v"(x));
    return x;
}
__device__ __forceinline__ double fma3(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct F64K {
    double l448, l192, l80, l32, l12, lk;          // -2 ln u
    double s2pi53, s120, sm6, sm720, s24;          // sincos
    __device__ __forceinline__ void load()
    {
        l448 = k_vgpr(1.0 / 448), l192 = k_vgpr(1.0 / 192), l80 = k_vgpr(1.0 / 80), l32 = k_vgpr(1.0 / 32), l12 = k_vgpr(1.0 / 12);
        lk = k_vgpr(-1.3862943611198906188);
        s2pi53 = k_vgpr(6.283185307179586477 * 0x1p-53), s120 = k_vgpr(1.0 / 120), sm6 = k_vgpr(-1.0 / 6), sm720 = k_vgpr(-1.0 / 720), s24 = k_vgpr(1.0 / 24);
    }
};

__device__ __forceinline__ double neg2log_unit_tab(double u, const F64K &K)
{
    const int h = __double2hiint(u) - 0x3fe6a09e;
    const int k = h >> 20;
    const double m = __hiloint2double(__double2hiint(u) - (k << 20), __double2loint(u));
    const F64Pair e = f64_tables[(h >> 13) & 0x7f];
    const double r = __builtin_fma(m, e.x, 2.0);
    double p = fma3(r, K.l448, K.l192);
    p = fma3(r, p, K.l80);
    p = fma3(r, p, K.l32);
    p = fma3(r, p, K.l12);
    p = __builtin_fma(r, p, 0.25);
    const double small = __builtin_fma(r * r, p, r);
    const double big = fma3((double)k, K.lk, e.y);
    return big + small;
}

__device__ __forceinline__ void sincos_turns_tab(uint32_t lo, uint32_t hi, const F64K &K, double &sin_out, double &cos_out)
{
    const F64Pair e = f64_tables[128 + (hi >> 24)];
    const uint32_t mant_lo = __builtin_amdgcn_alignbit(hi, lo, 12);
    const uint32_t mant_hi = ((hi >> 12) & 0xfffu) | 0x3ff00000u;
    const double y = __hiloint2double((int)mant_hi, (int)mant_lo) - (1.0 + 0x1p-9);
    const double b = fma3(y, 6.283185307179586477, K.s2pi53);
    const double z = b * b;
    const double ps = fma3(z, K.s120, K.sm6);
    const double sb = __builtin_fma(z * b, ps, b);
    double pc = fma3(z, K.sm720, K.s24);
    pc = __builtin_fma(z, pc, -0.5);
    const double cm = z * pc;
    sin_out = __builtin_fma(e.y, sb, __builtin_fma(e.x, cm, e.x));
    cos_out = __builtin_fma(-e.x, sb, __builtin_fma(e.y, cm, e.y));
}

}  // namespace mc
