/*
 * MonteCarlo.h -- drop-in type header for the MI355X-native engine.
 *
 * Layout-compatible with the reference's MonteCarlo.h (marcomatteo/MonteCarloCUDA,
 * double_precision/MonteCarlo.h:32-73 and the single_precision/ twin): same struct names, same
 * member names, same member order, therefore the same sizeof/offsetof under the x86-64 SysV
 * ABI (checked by the static asserts at the bottom and by tests/test_abi.py).  A driver
 * written against the reference header compiles unchanged against this one.
 *
 * Differences, all supersets:
 *   - one header for both precisions: define MC_SINGLE_PRECISION for the float layout
 *     (the reference keeps two copies of the file that differ only in the scalar type);
 *   - `N` (asset count of MultiOptionData) may be set with -DN=<n>; the reference hard-codes
 *     `#define N 3` without a guard (MonteCarlo.h:16);
 *   - the entry points the reference drivers re-declare by hand (vanillaOpt.cu:17-20,
 *     basketOpt.cu:17-21, cvaOpt.cu:17-20) are prototyped here;
 *   - no CUDA error macro: the HIP engine reports errors itself (see mc_mi355x.h).
 */
#ifndef MONTECARLO_H_
#define MONTECARLO_H_

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#ifndef N
#define N 3 /* assets in a basket; must match the libmcgpu build */
#endif

#ifdef MC_SINGLE_PRECISION
typedef float mc_real;
#else
typedef double mc_real;
#endif

/* One vanilla option under Black-Scholes dynamics. */
typedef struct {
    mc_real s; /* spot                   */
    mc_real k; /* strike                 */
    mc_real r; /* risk-free rate         */
    mc_real v; /* volatility             */
    mc_real t; /* maturity in years      */
} OptionData;

/* A basket of N assets.  At call time `p` holds the lower-triangular Cholesky factor of the
 * correlation matrix (the reference driver overwrites it in place, basketOpt.cu:96-99). */
typedef struct {
    mc_real s[N];    /* spots                                   */
    mc_real v[N];    /* volatilities                            */
    mc_real p[N][N]; /* correlation matrix / its Cholesky factor */
    mc_real d[N];    /* drift added to the correlated normals   */
    mc_real w[N];    /* basket weights                          */
    mc_real k;
    mc_real t;
    mc_real r;
} MultiOptionData;

/* Result of a simulation: estimate and 95 % half-width.  (The reference also re-uses this
 * struct for per-block partial sums on the device; this engine does not.) */
typedef struct {
    mc_real Expected;
    mc_real Confidence;
} OptionValue;

/* Credit valuation adjustment of one vanilla call. */
typedef struct {
    mc_real defInt, lgd; /* default intensity, loss given default */
    int ns;              /* unused by the reference, kept for layout */
    OptionData option;
    int n;               /* exposure dates on [0, option.t]          */
} CVA;

/* CPU-side scratch of the reference host path; kept for source compatibility. */
typedef struct {
    OptionValue callValue;
    MultiOptionData mopt;
    OptionData sopt;
    int numOpt, path;
} MonteCarloData;

#ifdef __cplusplus
extern "C" {
#endif

/* GPU entry points (reference MonteCarloKernel.cu:483,500,517).  numBlocks/numThreads are the
 * reference's launch-geometry arguments: this engine uses them only for the reference's path
 * count rule, paths = numBlocks * (sims / numBlocks) (MonteCarloKernel.cu:491,508,524,413). */
OptionValue dev_vanillaOpt(OptionData *opt, int numBlocks, int numThreads, int sims);
OptionValue dev_basketOpt(MultiOptionData *option, int numBlocks, int numThreads, int sims);
OptionValue dev_cvaEquityOption(CVA *cva, int numBlocks, int numThreads, int sims);

/* Not in the reference (its API has no seed parameter; the GPU path uses fixed seeds, MonteCarloKernel.cu:289, the CPU path
 * the wall clock, MonteCarloHost.c:189): the same calls with an explicit 64-bit seed instead of MC_SEED / the default. */
OptionValue dev_vanillaOpt_ex(OptionData *opt, int numBlocks, int numThreads, int sims, uint64_t seed);
OptionValue dev_basketOpt_ex(MultiOptionData *option, int numBlocks, int numThreads, int sims, uint64_t seed);
OptionValue dev_cvaEquityOption_ex(CVA *cva, int numBlocks, int numThreads, int sims, uint64_t seed);

/* Host entry points (reference MonteCarloHost.c:139,282,292,302,90,42,51; this repo:
 * libmchost_f64/_f32, montecarlocuda_amd/csrc/host_path.c -- a many-core CPU twin of the GPU
 * estimator on the same Philox stream, see that file's header). */
mc_real host_bsCall(OptionData option);
OptionValue host_vanillaOpt(OptionData option, int path);
OptionValue host_basketOpt(MultiOptionData *option, int path);
OptionValue host_cvaEquityOption(CVA *cva, int path);
OptionValue host_vanillaOpt_ex(OptionData option, int path, uint64_t seed);
OptionValue host_basketOpt_ex(MultiOptionData *option, int path, uint64_t seed);
OptionValue host_cvaEquityOption_ex(CVA *cva, int path, uint64_t seed);
void Chol(mc_real c[N][N], mc_real a[N][N]);
void printOption(OptionData o);
void printMultiOpt(MultiOptionData *o);
/* helpers the reference's host file leaves visible (MonteCarloHost.c:20,31,67,111; basketOpt.cu:21 declares
 * randMinMax): row-major matrices, randMinMax draws from libc's rand() */
void printVect(mc_real *vect, int c);
void printMat(mc_real *mat, int r, int c);
void prodMat(mc_real *first, mc_real *second, mc_real *result, int f_rows, int f_cols, int s_cols);
mc_real randMinMax(mc_real min, mc_real max);
/* Not in the reference: the number of threads the next host_* call will use -- MC_HOST_THREADS, else the hardware
 * threads OpenMP sees capped by the container's cgroup CPU quota (host_path.c). */
int mc_host_threads(void);

#ifdef __cplusplus
}
#endif

#if defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
_Static_assert(sizeof(OptionData) == 5 * sizeof(mc_real), "OptionData layout");
_Static_assert(sizeof(MultiOptionData) == (4 * N + N * N + 3) * sizeof(mc_real), "MultiOptionData layout");
_Static_assert(sizeof(OptionValue) == 2 * sizeof(mc_real), "OptionValue layout");
#endif

#endif /* MONTECARLO_H_ */
