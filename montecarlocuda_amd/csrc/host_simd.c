/*
 * host_simd.c -- the CPU twin's hot loops (vanilla, basket, CVA), written so that the compiler vectorises them.
 *
 * Same stream, same formulas as vanilla_chunk / basket_chunk / cva_chunk in host_path.c (the scalar forms, kept for
 * remainders and as the readable statement): Philox4x32-10, two-branch Box-Muller, the reference's device formulas
 * MonteCarloKernel.cu:67-129,241-262.
 * What differs is only the shape: paths are processed in batches of BATCH, structure-of-arrays, every stage a plain
 * counted loop, so gcc turns them into AVX2 / AVX-512 code and calls glibc's vector math (libmvec: expf, logf, sinf,
 * cosf and the double forms, <= 4 ulp).  This translation unit alone is compiled with -O3 -ffast-math (the vector math
 * variants are only offered under it); host_path.c keeps -O2 -ffp-contract=off because host_bsCall and Chol must
 * reproduce the reference bit for bit.  One binary: the widest ISA the CPU has is picked at run time (host_path.c).
 * Results: per-path values within a few ulp of the scalar form; chunk sums are added in a different (fixed) order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "mc_mi355x.h"

#ifdef MC_SINGLE_PRECISION
typedef float real;
#define NPB 4
#else
typedef double real;
#define NPB 8   /* fp64 stream version 2: eight normals per block of three Philox blocks (mc_rng.hpp) */
#endif

#define BATCH 256
/* Built three times per precision (Makefile): -march=x86-64, haswell (AVX2 + FMA), skylake-avx512 with 512-bit vectors
 * preferred; MC_SIMD_SUFFIX names the copy and host_path.c picks one at run time from what the CPU reports. */
#ifndef MC_SIMD_SUFFIX
#define MC_SIMD_SUFFIX base
#endif
#define MC_CAT2(a, b) a##_##b
#define MC_CAT(a, b) MC_CAT2(a, b)
#define SIMD_NAME(stem) MC_CAT(stem, MC_SIMD_SUFFIX)
#ifndef N
#define N 3   /* the reference's asset count (MonteCarlo.h:16); the Makefile passes the library's */
#endif

#ifdef MC_SINGLE_PRECISION
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#else
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#endif

/* Philox4x32-10 on BATCH counters {unit_hi, unit_lo + i, block, domain}; rounds outside, lanes inside,
 * the 32 x 32 -> 64 products as a high-part and a low-part multiply: the shape the vectoriser recognises */
static inline void philox_batch(uint64_t seed, uint64_t unit0, uint32_t block, uint32_t domain, uint32_t *restrict c0,
                                uint32_t *restrict c1, uint32_t *restrict c2, uint32_t *restrict c3)
{
    for (int i = 0; i < BATCH; ++i) {
        const uint64_t unit = unit0 + (uint64_t)i;
        c0[i] = (uint32_t)(unit >> 32), c1[i] = (uint32_t)unit, c2[i] = block, c3[i] = domain;
    }
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int round = 0; round < 10; ++round) {
        for (int i = 0; i < BATCH; ++i) {
            const uint32_t a = c0[i], b = c2[i];
            const uint32_t hi0 = (uint32_t)(((uint64_t)0xD2511F53u * a) >> 32), lo0 = 0xD2511F53u * a;
            const uint32_t hi1 = (uint32_t)(((uint64_t)0xCD9E8D57u * b) >> 32), lo1 = 0xCD9E8D57u * b;
            c0[i] = hi1 ^ c1[i] ^ k0;
            c2[i] = hi0 ^ c3[i] ^ k1;
            c1[i] = lo1;
            c3[i] = lo0;
        }
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}

/* the NPB normals of BATCH consecutive units (block `block` of each): z[j][i] = normal j of unit unit0 + i */
static inline void normals_batch(uint64_t seed, uint64_t unit0, uint32_t block, uint32_t domain, real z[NPB][BATCH])
{
    real radius[BATCH], ang[BATCH];
#ifdef MC_SINGLE_PRECISION
    uint32_t w0[BATCH], w1[BATCH], w2[BATCH], w3[BATCH];
    philox_batch(seed, unit0, block, domain, w0, w1, w2, w3);
    for (int h = 0; h < 2; ++h) {
        const uint32_t *restrict wa = h ? w2 : w0, *restrict wb = h ? w3 : w1;
        real *restrict zc = z[2 * h], *restrict zs = z[2 * h + 1];
        for (int i = 0; i < BATCH; ++i) {
            const float ua = (float)wa[i] * 0x1p-32f + 0x1p-33f;
            radius[i] = sqrtf(-2.0f * logf(ua));
            /* angle in revolutions: the word's top 23 bits (mc_rng.hpp angle_f32); reduced to [-1/2, 1/2) exactly */
            const float rev = (float)(wb[i] >> 9) * 0x1p-23f;
            ang[i] = 6.283185307179586f * (rev - (rev >= 0.5f ? 1.0f : 0.0f));
        }
        /* cos and sin in loops of their own: together gcc fuses them into a complex exponential it cannot vectorise */
        for (int i = 0; i < BATCH; ++i)
            zc[i] = radius[i] * cosf(ang[i]);
        for (int i = 0; i < BATCH; ++i)
            zs[i] = radius[i] * sinf(ang[i]);
    }
#else
    /* block b = Philox blocks 3b .. 3b + 2 = twelve words W[0..11] per unit; pair p = (a, m, c) = W[3p .. 3p + 2]:
     * 52-bit radius uniform (a : top 20 of m), 44-bit angle (c : low 12 of m) on top of a 52-bit fraction */
    static _Thread_local uint32_t W[12][BATCH];
    for (int s = 0; s < 3; ++s)
        philox_batch(seed, unit0, 3 * block + (uint32_t)s, domain, W[4 * s], W[4 * s + 1], W[4 * s + 2], W[4 * s + 3]);
    for (int p = 0; p < 4; ++p) {
        const uint32_t *restrict wa = W[3 * p], *restrict wm = W[3 * p + 1], *restrict wc = W[3 * p + 2];
        for (int i = 0; i < BATCH; ++i) {
            const double ua = ((double)(((uint64_t)wa[i] << 20) | (wm[i] >> 12)) + 0.5) * 0x1p-52;
            const double ub = ((double)((((uint64_t)wc[i] << 12) | (wm[i] & 0xfffu)) << 8) + 0.5) * 0x1p-52;
            radius[i] = sqrt(-2.0 * log(ua));
            ang[i] = 6.283185307179586477 * ub;
        }
        real *restrict zc = z[2 * p], *restrict zs = z[2 * p + 1];
        for (int i = 0; i < BATCH; ++i)
            zc[i] = radius[i] * cos(ang[i]);
        for (int i = 0; i < BATCH; ++i)
            zs[i] = radius[i] * sin(ang[i]);
    }
#endif
}

/* vanilla: sum and sum of squares of the payoffs of n_units whole units starting at unit0 (n_units a multiple of BATCH) */
void SIMD_NAME(mc_host_vanilla_units)(uint64_t seed, uint64_t unit0, long long n_units, real spot, real strike, real drift, real vol,
                                      int antithetic, double out[2])
{
    double s = 0, s2 = 0;
    real z[NPB][BATCH];
    for (long long u = 0; u < n_units; u += BATCH) {
        normals_batch(seed, unit0 + (uint64_t)u, 0, MC_DOMAIN_VANILLA, z);
        for (int j = 0; j < NPB; ++j) {
            double bs = 0, bs2 = 0;
            if (antithetic) {
                for (int i = 0; i < BATCH; ++i) {
                    const real up = spot * R_EXP(drift + vol * z[j][i]) - strike, dn = spot * R_EXP(drift - vol * z[j][i]) - strike;
                    const double pay = (double)((real)0.5 * ((up > 0 ? up : 0) + (dn > 0 ? dn : 0)));
                    bs += pay, bs2 += pay * pay;
                }
            } else {
                for (int i = 0; i < BATCH; ++i) {
                    const real v = spot * R_EXP(drift + vol * z[j][i]) - strike;
                    const double pay = (double)(v > 0 ? v : 0);
                    bs += pay, bs2 += pay * pay;
                }
            }
            s += bs, s2 += bs2;
        }
    }
    out[0] = s, out[1] = s2;
}

/* basket (MonteCarloKernel.cu:74-101, host_path.c basket_chunk): n_paths a multiple of BATCH, paths first .. first + n - 1.
 * p = the N x N factor, row-major (lower triangle used). */
void SIMD_NAME(mc_host_basket_paths)(uint64_t seed, uint64_t first, long long n_paths, const real *p, const real *d, const real *v,
                                     const real *s0, const real *w, real strike, real t, real r, int antithetic, int control,
                                     double out[2])
{
    enum { NBLK = (N + NPB - 1) / NPB };
    real g[NBLK * NPB][BATCH];
    const real sqrt_t = (real)sqrt((double)t);
    double wsum = 0;
    for (int a = 0; a < N; ++a)
        wsum += (double)w[a];
    real mu[N], wn[N], ls[N];
    for (int a = 0; a < N; ++a) {
        mu[a] = (real)(((double)r - 0.5 * (double)v[a] * (double)v[a]) * (double)t);
        wn[a] = (real)((double)w[a] / wsum);
        ls[a] = R_LOG(s0[a]);
    }
    const real lg0 = (real)log(wsum);
    double sum = 0, sum2 = 0;
    real payoff[BATCH], basket[BATCH], lg[BATCH], bt[BATCH];
    for (long long u = 0; u < n_paths; u += BATCH) {
        for (int b = 0; b < NBLK; ++b)
            normals_batch(seed, first + (uint64_t)u, (uint32_t)b, MC_DOMAIN_BASKET, (real(*)[BATCH])g[b * NPB]);
        for (int i = 0; i < BATCH; ++i)
            payoff[i] = 0;
        for (int sign = 1; sign >= (antithetic ? -1 : 1); sign -= 2) {
            const real sg = (real)sign;
            for (int i = 0; i < BATCH; ++i)
                basket[i] = 0, lg[i] = lg0;
            for (int a = 0; a < N; ++a) {
                for (int i = 0; i < BATCH; ++i)
                    bt[i] = 0;
                for (int b = 0; b <= a; ++b) {
                    const real pab = p[a * N + b];
                    const real *restrict gb = g[b];
                    for (int i = 0; i < BATCH; ++i)
                        bt[i] += pab * (sg * gb[i]);
                }
                const real da = d[a], va = v[a] , ma = mu[a], cf = s0[a], wa = w[a], wna = wn[a], lsa = ls[a];
                for (int i = 0; i < BATCH; ++i) {
                    const real x = ma + va * (bt[i] + da) * sqrt_t;
                    basket[i] += cf * R_EXP(x) * wa;
                    lg[i] += wna * (lsa + x);
                }
            }
            if (control) {
                for (int i = 0; i < BATCH; ++i) {
                    const real vv = basket[i] - strike, gv = R_EXP(lg[i]) - strike;
                    payoff[i] += (vv > 0 ? vv : 0) - (gv > 0 ? gv : 0);
                }
            } else {
                for (int i = 0; i < BATCH; ++i) {
                    const real vv = basket[i] - strike;
                    payoff[i] += vv > 0 ? vv : 0;
                }
            }
        }
        double bs = 0, bs2 = 0;
        const real half = antithetic ? (real)0.5 : (real)1;
        for (int i = 0; i < BATCH; ++i) {
            const double pay = (double)(payoff[i] * half);
            bs += pay, bs2 += pay * pay;
        }
        sum += bs, sum2 += bs2;
    }
    out[0] = sum, out[1] = sum2;
}

/* Hastings CDF and Black-Scholes call with the casts of host_path.c (hastings_cdf, bs_call), one element */
static inline real cdf_h(real d)
{
    const real kk = (real)(1.0 / (1.0 + 0.2316419 * fabs((double)d)));
    real poly = (real)1.330274429;
    poly = (real)-1.821255978 + kk * poly;
    poly = (real)1.781477937 + kk * poly;
    poly = (real)-0.356563782 + kk * poly;
    poly = (real)0.31938153 + kk * poly;
    poly *= kk;
    const real tail = (real)0.39894228040143267793994605993438 * R_EXP((real)(-0.5 * (double)d * (double)d)) * poly;
    return d > 0 ? (real)(1.0 - (double)tail) : tail;
}

/* CVA, device ordering (MonteCarloKernel.cu:241-262, host_path.c cva_chunk): n_paths a multiple of BATCH.  Everything that
 * depends only on the date -- default-probability increment, residual maturity, vol sqrt(tau), K e^{-r tau} -- is a
 * scalar per date; the lanes carry spot, mirrored spot and the running sum. */
void SIMD_NAME(mc_host_cva_paths)(uint64_t seed, uint64_t first, long long n_paths, real s0, real strike, real r, real v, real t,
                                  int n_dates, real defint, real lgd, int antithetic, double out[2])
{
    const real dt = t / n_dates;
    const real step_drift = (real)(((double)r - 0.5 * (double)v * (double)v) * (double)dt);
    const real step_vol = (real)((double)v * sqrt((double)dt));
    double sum = 0, sum2 = 0;
    real z[NPB][BATCH], spot[BATCH], mirror[BATCH], acc[BATCH], ee[BATCH];
    for (long long u = 0; u < n_paths; u += BATCH) {
        for (int i = 0; i < BATCH; ++i)
            spot[i] = mirror[i] = s0, acc[i] = 0;
        real ttm = t;
        for (int j = 1; j <= n_dates; ++j) {
            const double t_prev = (double)dt * (j - 1), t_now = (double)dt * j;
            const real dpd = (real)(-exp(-(double)defint * t_prev) * expm1(-(double)defint * (t_now - t_prev)));
            ttm -= dt;
            if (!(ttm >= 0))
                continue;
            const int idx = j - 1;
            if (idx % NPB == 0)
                normals_batch(seed, first + (uint64_t)u, (uint32_t)(idx / NPB), MC_DOMAIN_CVA, z);
            const real *restrict zj = z[idx % NPB];
            for (int i = 0; i < BATCH; ++i)
                spot[i] = spot[i] * R_EXP(step_drift + step_vol * zj[i]);
            if (antithetic)
                for (int i = 0; i < BATCH; ++i)
                    mirror[i] = mirror[i] * R_EXP(step_drift - step_vol * zj[i]);
            for (int i = 0; i < BATCH; ++i)
                ee[i] = 0;
            for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
                const real *restrict sx = leg ? mirror : spot;
                if (ttm == 0) {
                    for (int i = 0; i < BATCH; ++i)
                        ee[i] += sx[i] > strike ? sx[i] - strike : 0;
                } else {
                    const real vol = v * R_SQRT(ttm), kdisc = strike * R_EXP(-r * ttm);
                    const double c1 = ((double)r + 0.5 * (double)v * (double)v) * (double)ttm;
                    for (int i = 0; i < BATCH; ++i) {
                        const real d1 = (real)(((double)R_LOG(sx[i] / strike) + c1) / (double)vol);
                        const real d2 = d1 - vol;
                        ee[i] += sx[i] * cdf_h(d1) - kdisc * cdf_h(d2);
                    }
                }
            }
            const real scale = antithetic ? (real)0.5 * dpd : dpd;
            for (int i = 0; i < BATCH; ++i)
                acc[i] += scale * ee[i];
        }
        double bs = 0, bs2 = 0;
        for (int i = 0; i < BATCH; ++i) {
            const double pay = (double)(acc[i] * lgd);
            bs += pay, bs2 += pay * pay;
        }
        sum += bs, sum2 += bs2;
    }
    out[0] = sum, out[1] = sum2;
}
