/*
 * mc_oracle.c -- TEST INFRASTRUCTURE ONLY (see mc_oracle.h for the contract and the
 * parity-pinning statement).  Instantiates mc_oracle_impl.h for f32 and f64 and holds the
 * precision-independent pieces: Philox4x32-10 and the fp64 closing formulas.
 */
#define _GNU_SOURCE
#include <math.h>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>

#include "mc_oracle.h"

/* ------------------------------------------------------------------------------------- */
/* Philox4x32-10, as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
 * easy as 1, 2, 3", SC'11; Random123 v1.09 philox.h).  rocRAND / cuRAND ship the same
 * generator (the reference draws from cuRAND: dp/MonteCarloKernel.cu:68,78,250,289; its
 * XORWOW stream is not reproduced -- BASELINE.json north_star allows Philox).
 * Pinned by the Random123 known-answer vectors in tests/golden/philox_kat.json.           */
/* ------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u; /* round multipliers */
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u; /* Weyl key increments */
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += W0, k1 += W1;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* price = discount * sum/n;  s^2 = (n sum2 - sum^2) / (n (n-1));  CI = 1.96 s / sqrt(n).
 * dp/MonteCarloHost.c:220-228, dp/MonteCarloKernel.cu:420-423 (and :466-468 for CVA). */
void orc_closing(double sum, double sum2, long long n, double discount, double *expected,
                 double *confidence)
{
    double dn = (double)n;
    *expected = discount * (sum / dn);
    double dev = sqrt((dn * sum2 - sum * sum) / (dn * (double)(n - 1)));
    *confidence = 1.96 * dev / sqrt(dn);
}

/* ---- f32 instantiation ---- */
#define REAL float
#define X f32
#define SQRT_R sqrtf
#define LOG_R logf
#define EXP_R expf
#define ORC_IS_F32 1
#define ORC_NPB 4
#include "mc_oracle_impl.h"
#undef REAL
#undef X
#undef SQRT_R
#undef LOG_R
#undef EXP_R
#undef ORC_IS_F32
#undef ORC_NPB

/* ---- f64 instantiation ---- */
#define REAL double
#define X f64
#define SQRT_R sqrt
#define LOG_R log
#define EXP_R exp
#define ORC_IS_F32 0
#define ORC_NPB 2
#include "mc_oracle_impl.h"
