/*
 * host_simd.c -- the CPU twin's vanilla hot loop, written so that the compiler vectorises it.
 *
 * Same stream, same formulas as vanilla_chunk in host_path.c (the scalar form, kept for partial units and as the
 * readable statement): Philox4x32-10 per unit, two-branch Box-Muller, reference payoff MonteCarloKernel.cu:67-71.
 * What differs is only the shape: units are processed in batches of BATCH, structure-of-arrays, every stage a plain
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
#define NPB 2
#endif

#define BATCH 256
/* Built three times per precision (Makefile): -march=x86-64, haswell (AVX2 + FMA), skylake-avx512 with 512-bit vectors
 * preferred; MC_SIMD_NAME names the copy and host_path.c picks one at run time from what the CPU reports. */
#ifndef MC_SIMD_NAME
#define MC_SIMD_NAME mc_host_vanilla_units_base
#endif

/* Philox4x32-10 on BATCH counters {unit_hi, unit_lo + i, 0, domain}; rounds outside, lanes inside,
 * the 32 x 32 -> 64 products as a high-part and a low-part multiply: the shape the vectoriser recognises */
static inline void philox_batch(uint64_t seed, uint64_t unit0, uint32_t domain, uint32_t *restrict c0, uint32_t *restrict c1,
                                uint32_t *restrict c2, uint32_t *restrict c3)
{
    for (int i = 0; i < BATCH; ++i) {
        const uint64_t unit = unit0 + (uint64_t)i;
        c0[i] = (uint32_t)(unit >> 32), c1[i] = (uint32_t)unit, c2[i] = 0, c3[i] = domain;
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

/* sum and sum of squares of the payoffs of n_units whole units starting at unit0 (n_units a multiple of BATCH) */
void MC_SIMD_NAME(uint64_t seed, uint64_t unit0, long long n_units, real spot, real strike, real drift, real vol,
                                     int antithetic, double out[2])
{
    double s = 0, s2 = 0;
    uint32_t w0[BATCH], w1[BATCH], w2[BATCH], w3[BATCH];
    real z[NPB][BATCH], radius[BATCH], ang[BATCH];
    for (long long u = 0; u < n_units; u += BATCH) {
        philox_batch(seed, unit0 + (uint64_t)u, MC_DOMAIN_VANILLA, w0, w1, w2, w3);
#ifdef MC_SINGLE_PRECISION
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
        for (int i = 0; i < BATCH; ++i) {
            const double ua = ((double)(((uint64_t)w1[i] << 20) | (w0[i] >> 12)) + 0.5) * 0x1p-52;
            const double ub = ((double)(((uint64_t)w3[i] << 20) | (w2[i] >> 12)) + 0.5) * 0x1p-52;
            radius[i] = sqrt(-2.0 * log(ua));
            ang[i] = 6.283185307179586477 * ub;
        }
        for (int i = 0; i < BATCH; ++i)
            z[0][i] = radius[i] * cos(ang[i]);
        for (int i = 0; i < BATCH; ++i)
            z[1][i] = radius[i] * sin(ang[i]);
#endif
        for (int j = 0; j < NPB; ++j) {
            double bs = 0, bs2 = 0;
            if (antithetic) {
                for (int i = 0; i < BATCH; ++i) {
#ifdef MC_SINGLE_PRECISION
                    const real up = spot * expf(drift + vol * z[j][i]) - strike, dn = spot * expf(drift - vol * z[j][i]) - strike;
#else
                    const real up = spot * exp(drift + vol * z[j][i]) - strike, dn = spot * exp(drift - vol * z[j][i]) - strike;
#endif
                    const double pay = (double)((real)0.5 * ((up > 0 ? up : 0) + (dn > 0 ? dn : 0)));
                    bs += pay, bs2 += pay * pay;
                }
            } else {
                for (int i = 0; i < BATCH; ++i) {
#ifdef MC_SINGLE_PRECISION
                    const real v = spot * expf(drift + vol * z[j][i]) - strike;
#else
                    const real v = spot * exp(drift + vol * z[j][i]) - strike;
#endif
                    const double pay = (double)(v > 0 ? v : 0);
                    bs += pay, bs2 += pay * pay;
                }
            }
            s += bs, s2 += bs2;
        }
    }
    out[0] = s, out[1] = s2;
}
