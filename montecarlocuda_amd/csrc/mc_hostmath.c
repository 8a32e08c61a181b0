/*
 * mc_hostmath.c -- the host-only part of the C ABI (include/mc_mi355x.h "host-side helpers"): plain C, no HIP.
 *
 * Compiled ONCE into an object that is linked into both libmc_mi355x.so (the GPU engine) and libmchost_*.so (the
 * OpenMP CPU twin), so that the CPU twin loads on a machine without the ROCm runtime (SURVEY 8f-3 "no-GPU fallback";
 * tests/test_build_deps.py checks its dynamic section).  What lives here:
 *   mc_closing               closing formulas, dp/MonteCarloKernel.cu:420-423 and :466-468
 *   mc_shard_range           SURVEY 8e partitioning
 *   mc_chol_*                dp/MonteCarloHost.c:90-105 semantics
 *   mc_factor_from_cov_*     SURVEY 8f-2
 *   mc_basket_control_mean_* closed-form mean of the geometric-basket control variate
 *   mc_last_error / mc_internal_fail   the per-thread error text of whichever library this object is linked into
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/mc_mi355x.h"
#include "mc_hostmath.h"

static _Thread_local char g_last_error[640];

const char *mc_last_error(void) { return g_last_error; }

int mc_internal_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
    return code;
}

void mc_closing(double sum, double sum2, uint64_t n, double discount, double *expected, double *confidence)
{
    const double dn = (double)n;
    if (expected) *expected = discount * (sum / dn);
    if (confidence) {
        const double dev = sqrt((dn * sum2 - sum * sum) / (dn * (double)(n - 1)));
        *confidence = 1.96 * dev / sqrt(dn);
    }
}

void mc_shard_range(uint64_t total, int rank, int world, uint64_t *first, uint64_t *count)
{
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    if (rank >= world) rank = world - 1;
    /* floor(rank * total / world) without overflowing 64 bits */
    const unsigned __int128 t = total;
    const uint64_t lo = (uint64_t)(t * (unsigned)rank / (unsigned)world);
    const uint64_t hi = (uint64_t)(t * (unsigned)(rank + 1) / (unsigned)world);
    if (first) *first = lo;
    if (count) *count = hi - lo;
}

/* One body per precision: REAL, SQRT_R, X set by the includer below. */
#define MC_HM_CAT_(a, b) a##_##b
#define MC_HM_CAT(a, b) MC_HM_CAT_(a, b)
#define FN(name) MC_HM_CAT(name, X)

#define REAL float
#define SQRT_R sqrtf
#define X f32
#define BASKET mc_basket_f32
#include "mc_hostmath_impl.h"
#undef REAL
#undef SQRT_R
#undef X
#undef BASKET

#define REAL double
#define SQRT_R sqrt
#define X f64
#define BASKET mc_basket_f64
#include "mc_hostmath_impl.h"
#undef REAL
#undef SQRT_R
#undef X
#undef BASKET
