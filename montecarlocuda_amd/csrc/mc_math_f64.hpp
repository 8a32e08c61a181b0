// mc_math_f64.hpp -- double-precision elementary functions cut to the Monte Carlo kernels' domains.
//
// Why not the library ones: on gfx950 every fp64 VALU instruction costs a 4-cycle issue slot and
// the kernels are VALU-issue-bound, so instruction count is time.  The general
// ocml routines pay for double-double intermediates and for arguments these kernels never see
// (measured VALU instructions: log 98, sincospi 70, sqrt 22, 1/x 12).  The versions below assume
// what the generator guarantees -- a uniform strictly inside (0,1), a positive finite radicand --
// and stay within 1-2 ulp of the correctly rounded result (checked against glibc through the
// oracle's normals: tests/test_gpu_parity.py, bound 2e-14 absolute on |z| < 8.3; measured 2.9e-15):
//
//   log_unit(u)        ~36 instructions   fdlibm's e_log scheme: u = 2^k (1+f), s = f/(2+f),
//                                         7-term minimax in s^2 (max error 0.8 ulp)
//   sqrt_pos(x)        ~7                 v_rsq_f64 + one coupled Newton step + one residual step (no rescaling)
//   sincos_turns(u)    ~36                quadrant from rint(4u), Taylor to y^15 / y^16 on |y| <= 1/2
//   recip_pos(d)       ~5                 v_rcp_f64 + two Newton steps
//   exp_f64(x)         ~19                n = rint(x log2 e), two-step reduction, Taylor to r^13, v_ldexp_f64
//                      ~13 + ds_read      (default) 256-entry table of 2^(j/256), Taylor to r^4
//
// The two functions inside Box-Muller additionally have table-driven forms (the default), which trade
// polynomial length for one 16-byte LDS read each -- LDS reads do not occupy the VALU:
//   neg2log_unit_tab(u) ~16 + ds_read     128-entry table over the reduced mantissa: -2 ln m = t_i + S(m c2_i + 2),
//                                         S a 7-term series on |r'| <= 2^-7            (replaces log_unit + a multiply)
//   sincos_turns_tab    ~18 + ds_read     256-entry table of (sin, cos) at the slot centres, angle-addition with
//                                         2-/3-term series on |b| <= 2 pi / 512          (replaces sincos_turns)
// tools/check_f64_tables.c (host twin, vs 80-bit libm over 4e7 inputs): -2 ln u within 2.8e-16 relative,
// sin/cos within 1.7e-16 absolute, the normal within 1.8e-15 absolute -- the same as the polynomial forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

// Build switches for in-process A/B runs (tools/ab_f64.py builds one .so per combination and times them
// interleaved on one device).  Measured on MI355X, medians, default build = 1.00:
//                     vanilla f64   basket n=4 / n=16 f64   CVA 256 dates f64
//   default                1.00          1.00 / 1.00              1.00
//   MC_AB_OCML_EXP         1.04          1.04 / 1.03              1.04      (ocml exp: ~10 extra v_mov_b64)
//   MC_AB_CONST_COEF       1.00          1.19 / 1.03              1.11      (coefficients in constant memory:
//                                                              scalar reloads in the loop cost more than VGPRs)
//   both                   1.02          1.20 / 1.07              1.13      (profiles/r01_ab_f64_math.log)
#ifdef MC_AB_CONST_COEF
#define MC_COEF_STORAGE __constant__
#else
#define MC_COEF_STORAGE static constexpr
#endif
MC_COEF_STORAGE double LOG_LG[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
                                 2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
                                 1.479819860511658591e-01};
MC_COEF_STORAGE double SIN_Q[8] = {1.5707963267948966,     -0.6459640975062463,    0.07969262624616705,   -0.004681754135318688,
                                0.00016044118478735983, -3.598843235212085e-06, 5.692172921967927e-08, -6.688035109811468e-10};
MC_COEF_STORAGE double COS_Q[8] = {-1.2337005501361697,     0.25366950790104803,   -0.02086348076335296,   0.0009192602748394266,
                                -2.5202042373060607e-05, 4.710874778818172e-07, -6.386603083791852e-09, 6.565963114979473e-11};

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

// e^x for finite x (underflows to 0 through v_ldexp_f64; these kernels never overflow it):
// x = n ln2 + r, |r| <= ln2/2, Taylor to r^13: 19 instructions, max error 0.86 ulp.
#if !defined(MC_AB_OCML_EXP) && !defined(MC_AB_EXP_POLY) && !defined(MC_AB_NO_TABLES)
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
#elif !defined(MC_AB_OCML_EXP)
__device__ __forceinline__ double exp_f64(double x)
{
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = __builtin_fma(r, 1.6059043836821613e-10, 2.08767569878681e-09);
    p = __builtin_fma(r, p, 2.505210838544172e-08);
    p = __builtin_fma(r, p, 2.755731922398589e-07);
    p = __builtin_fma(r, p, 2.7557319223985893e-06);
    p = __builtin_fma(r, p, 2.48015873015873e-05);
    p = __builtin_fma(r, p, 0.0001984126984126984);
    p = __builtin_fma(r, p, 0.001388888888888889);
    p = __builtin_fma(r, p, 0.008333333333333333);
    p = __builtin_fma(r, p, 0.041666666666666664);
    p = __builtin_fma(r, p, 0.16666666666666666);
    p = __builtin_fma(r, p, 0.5);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}
#else
__device__ __forceinline__ double exp_f64(double x) { return exp(x); }
#endif

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
#ifdef MC_AB_TWO_RCP
    ra = recip_pos(a);
    rb = recip_pos(b);
#else
    const double r = recip_pos(a * b);
    ra = r * b;
    rb = r * a;
#endif
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

// natural log of a positive normal double
__device__ __forceinline__ double log_unit(double x)
{
    // x = 2^k * m with m in [sqrt(1/2), sqrt(2)): shift the exponent boundary by adding the
    // distance between the bit patterns of 1.0 and sqrt(1/2) to the high word
    const int hi = __double2hiint(x) + (0x3ff00000 - 0x3fe6a09e);
    const int k = (hi >> 20) - 0x3ff;
    const double m = __hiloint2double((hi & 0x000fffff) + 0x3fe6a09e, __double2loint(x));
    const double f = m - 1.0;
    // s = f / (2 + f), correctly rounded by one residual step
    const double d = 2.0 + f;
    const double r = recip_pos(d);
    double s = f * r;
    s = __builtin_fma(__builtin_fma(-s, d, f), r, s);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, LOG_LG[5], LOG_LG[3]), LOG_LG[1]);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, LOG_LG[6], LOG_LG[4]), LOG_LG[2]), LOG_LG[0]);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
    const double inner = __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10);
    return __builtin_fma(dk, 6.93147180369123816490e-01, -((hfsq - inner) - f));
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
#ifdef MC_AB_SQRT_LONG   // ocml's second residual step (correct rounding); the first already gives < 1 ulp
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
#endif
    return __builtin_fma(d, h, g);
}

// ---- table-driven forms of the Box-Muller pieces -------------------------------------------------
// -2 ln(u) for a positive normal double.
// u = 2^k m, m in [sqrt(1/2), sqrt(2)) as in log_unit; slot i = top 7 bits of the shifted mantissa;
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
#ifdef MC_AB_LONG_COMBINE
    sin_out = e.x + __builtin_fma(e.x, cm, e.y * sb);
    cos_out = e.y + __builtin_fma(e.y, cm, -(e.x * sb));
#else   // two fmas each instead of mul + fma + add: max error 1.65e-16 instead of 1.10e-16 (tools/check_f64_tables.c)
    sin_out = __builtin_fma(e.y, sb, __builtin_fma(e.x, cm, e.x));
    cos_out = __builtin_fma(-e.x, sb, __builtin_fma(e.y, cm, e.y));
#endif
}

// (sin, cos) of 2*pi*u for u in [0, 1]: quadrant q = rint(4u), y = 4u - q in [-1/2, 1/2]
// (both exact), then sin(pi/2 y) and cos(pi/2 y) by Taylor polynomials (remainders 1.2e-17, 2.3e-18).
__device__ __forceinline__ void sincos_turns(double u, double &sin_out, double &cos_out)
{
    const double t = 4.0 * u;
    const double q = __builtin_rint(t);
    const double y = t - q;
    const double z = y * y;
    double ps = __builtin_fma(z, SIN_Q[7], SIN_Q[6]);
#pragma unroll
    for (int i = 5; i >= 0; --i)
        ps = __builtin_fma(z, ps, SIN_Q[i]);
    const double sy = y * ps;
    double pc = __builtin_fma(z, COS_Q[7], COS_Q[6]);
#pragma unroll
    for (int i = 5; i >= 0; --i)
        pc = __builtin_fma(z, pc, COS_Q[i]);
    const double cy = __builtin_fma(z, pc, 1.0);
    // angle = (pi/2)(q + y):  q mod 4 = 0: (sy, cy)  1: (cy, -sy)  2: (-sy, -cy)  3: (-cy, sy)
    const int qi = (int)q;
    const bool odd = (qi & 1) != 0;
    const double s0 = odd ? cy : sy;
    const double c0 = odd ? sy : cy;
    const uint32_t uq = (uint32_t)qi;
    const uint32_t sin_flip = (uq & 2u) << 30;         // bit 31 set for q mod 4 in {2, 3}
    const uint32_t cos_flip = ((uq + 1u) & 2u) << 30;  // bit 31 set for q mod 4 in {1, 2}
    sin_out = __hiloint2double((int)((uint32_t)__double2hiint(s0) ^ sin_flip), __double2loint(s0));
    cos_out = __hiloint2double((int)((uint32_t)__double2hiint(c0) ^ cos_flip), __double2loint(c0));
}

}  // namespace mc
