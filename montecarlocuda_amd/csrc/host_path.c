/*
 * host_path.c -- the reference's HOST entry points (libmchost_f64.so / libmchost_f32.so).
 *
 * Product code, plain C + OpenMP, built once per precision like legacy_abi.c.  It provides the
 * symbols the reference drivers take from MonteCarloHost.c (SURVEY 8b "host side"):
 *
 *   host_bsCall  (MonteCarloHost.c:139)   Chol          (:90)
 *   host_vanillaOpt (:282)  host_basketOpt (:292)  host_cvaEquityOption (:302)
 *   printOption  (:42)   printMultiOpt (:51)
 *
 * a many-core CPU twin of the GPU engine, NOT a fallback of it (the dev_*
 * entry points never call into this file) and NOT the oracle (tests check this file against
 * the oracle like any other product code).  Differences from the reference CPU path, by design:
 *   - same estimator as the GPU: the reference DEVICE formulas (MonteCarloKernel.cu:67-129,
 *     241-262) on the engine's Philox4x32-10 stream, so for one seed the CPU and GPU results
 *     agree path by path to rounding (the reference's CPU and GPU use unrelated streams, its dp
 *     CPU basket drops the volatility, and its CPU CVA lags the exposure: SURVEY 2.3 #1,#7,#12);
 *   - all host cores (OpenMP over fixed 65536-path chunks, chunk sums added in index order:
 *     the result does not depend on the thread count); MC_HOST_THREADS caps the threads;
 *   - sums in double for both precisions; seed = MC_SEED or MC_DEFAULT_SEED.
 * host_bsCall and Chol keep the reference's arithmetic (Hastings CDF, zero-pivot rule).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "MonteCarlo.h"
#include "mc_mi355x.h"

#ifdef MC_SINGLE_PRECISION
#define NPB 4
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#else
#define NPB 8   /* fp64 stream version 2: eight normals per block of three Philox blocks (mc_rng.hpp) */
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- closed forms ----------------------------------------------------------------------- */
static mc_real hastings_cdf(mc_real d)
{
    /* Abramowitz-Stegun 26.2.17, the constants of MonteCarloHost.c:125-130 */
    static const double c[5] = {0.31938153, -0.356563782, 1.781477937, -1.821255978, 1.330274429};
    const mc_real k = (mc_real)(1.0 / (1.0 + 0.2316419 * fabs((double)d)));
    mc_real poly = (mc_real)c[4];
    for (int i = 3; i >= 0; --i)
        poly = (mc_real)c[i] + k * poly;
    poly *= k;
    const mc_real tail = (mc_real)0.39894228040143267793994605993438 * R_EXP((mc_real)(-0.5 * (double)d * (double)d)) * poly;
    return d > 0 ? (mc_real)(1.0 - (double)tail) : tail;
}

static mc_real bs_call(mc_real s, mc_real k, mc_real r, mc_real v, mc_real t)
{
    const mc_real vol = v * R_SQRT(t);
    const mc_real d1 = (mc_real)(((double)R_LOG(s / k) + ((double)r + 0.5 * (double)v * (double)v) * (double)t) / (double)vol);
    const mc_real d2 = d1 - vol;
    return s * hastings_cdf(d1) - k * R_EXP(-r * t) * hastings_cdf(d2);
}

mc_real host_bsCall(OptionData option) { return bs_call(option.s, option.k, option.r, option.v, option.t); }

void Chol(mc_real c[N][N], mc_real a[N][N])
{
#ifdef MC_SINGLE_PRECISION
    mc_chol_f32(N, &c[0][0], &a[0][0]);
#else
    mc_chol_f64(N, &c[0][0], &a[0][0]);
#endif
}

/* ---- small helpers the reference's host file also exports (MonteCarloHost.c:20,31,67,111) -----
 * Not used by the estimators here; kept so that a driver declaring them (basketOpt.cu:21 declares
 * randMinMax) links against this library unchanged. */
static void print_rows(const mc_real *a, int rows, int cols)
{
    for (int i = 0; i < rows; ++i) {
        printf("\n!\t");
        for (int j = 0; j < cols; ++j)
            printf("\t%f\t", (double)a[(size_t)i * cols + j]);
        printf("\t!");
    }
    printf("\n\n");
}
void printVect(mc_real *vect, int c) { print_rows(vect, 1, c); }
void printMat(mc_real *mat, int r, int c) { print_rows(mat, r, c); }

/* result (f_rows x s_cols) = first (f_rows x f_cols) * second (f_cols x s_cols), row-major, accumulated in mc_real */
void prodMat(mc_real *first, mc_real *second, mc_real *result, int f_rows, int f_cols, int s_cols)
{
    for (int i = 0; i < f_rows; ++i)
        for (int j = 0; j < s_cols; ++j) {
            mc_real acc = 0;
            for (int k = 0; k < f_cols; ++k)
                acc += first[(size_t)i * f_cols + k] * second[(size_t)k * s_cols + j];
            result[(size_t)i * s_cols + j] = acc;
        }
}

/* uniform in [min, max] from libc's rand(), the reference's convention x = rand()/RAND_MAX */
mc_real randMinMax(mc_real min, mc_real max)
{
    const mc_real x = (mc_real)rand() / (mc_real)RAND_MAX;
    return max * x + ((mc_real)1 - x) * min;
}

/* ---- printers (same content as MonteCarloHost.c:42-65, own wording) ------------------------ */
void printOption(OptionData o)
{
    printf("\n-\tOption data\t-\n\n");
    printf("Underlying asset price:\t %.2f\nStrike price:\t\t %.2f\n", (double)o.s, (double)o.k);
    printf("Risk free interest rate: %.2f %%\nVolatility:\t\t %.2f %%\n", (double)o.r * 100, (double)o.v * 100);
    printf("Time to maturity:\t %.2f %s\n", (double)o.t, o.t > 1 ? "years" : "year");
}

void printMultiOpt(MultiOptionData *o)
{
    printf("\n-\tBasket Option data\t-\n\nNumber of assets: %d\n", N);
    const mc_real *rows[3] = {o->s, o->v, o->w};
    const char *names[3] = {"Underlying assets prices:", "Volatility:", "Weights:"};
    for (int k = 0; k < 3; ++k) {
        printf("%s\n!\t", names[k]);
        for (int i = 0; i < N; ++i)
            printf("\t%f\t", (double)rows[k][i]);
        printf("\t!\n");
    }
    printf("Correlation matrix (Cholesky factor at call time):\n");
    for (int i = 0; i < N; ++i) {
        printf("!\t");
        for (int j = 0; j < N; ++j)
            printf("\t%f\t", (double)o->p[i][j]);
        printf("\t!\n");
    }
    printf("Strike price:\t\t %.2f\nRisk free interest rate: %.2f\nTime to maturity:\t %.2f %s\n", (double)o->k,
           (double)o->r, (double)o->t, o->t > 1 ? "years" : "year");
}

/* ---- the engine's random stream on the CPU ---------------------------- */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1, c3 = (uint32_t)p0, c0 = n0, c2 = n2;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* MC_F64_NORMALS=f32 (fp64 build only; read once per call): the GPU engine's MC_NORMALS_F32 mode -- FOUR fp32 normals per
 * Philox block, widened to double, the reference's own dp arithmetic (dp/MonteCarloKernel.cu:68,78,250).  The scalar loops
 * below then run with 4 normals per block (g_npb); the vectorised loops of host_simd.c are compiled for the native layout
 * and are bypassed in this mode. */
static int g_normals_f32;
static int g_npb = NPB;   /* normals one block yields under the current mode */
#define MAX_NPB 8

static void block_normals_f32(const uint32_t x[4], float z[4])
{
    for (int h = 0; h < 2; ++h) {
        const float ua = fmaf((float)x[2 * h], 0x1p-32f, 0x1p-33f);
        const uint32_t ub_bits = (x[2 * h + 1] >> 9) | 0x3f800000u;  /* angle in revolutions, in [1, 2): mc_rng.hpp angle_f32 */
        float ub;
        memcpy(&ub, &ub_bits, 4);
        const float radius = sqrtf(-1.3862943611198906f * log2f(ua));
        const double ang = 2.0 * M_PI * (double)ub;
        z[2 * h] = radius * (float)cos(ang);
        z[2 * h + 1] = radius * (float)sin(ang);
    }
}

/* z has room for MAX_NPB values; g_npb of them are written */
static void block_normals(uint64_t seed, uint32_t domain, uint64_t unit, uint32_t block, mc_real *z)
{
#ifdef MC_SINGLE_PRECISION
    uint32_t x[4];
    philox4x32_10((uint32_t)(unit >> 32), (uint32_t)unit, block, domain, (uint32_t)seed, (uint32_t)(seed >> 32), x);  /* counter = {unit_hi, unit_lo, block, domain} */
    block_normals_f32(x, z);
#else
    if (g_normals_f32) {
        uint32_t x[4];
        float f[4];
        philox4x32_10((uint32_t)(unit >> 32), (uint32_t)unit, block, domain, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        block_normals_f32(x, f);
        for (int j = 0; j < 4; ++j)
            z[j] = (double)f[j];
        return;
    }
    /* native fp64: block b = Philox blocks 3b .. 3b + 2 = twelve words, four Box-Muller pairs of 96 bits:
     * pair p = (a, m, c) = W[3p .. 3p + 2]; radius from the 52 bits (a : top 20 of m), angle from the 44 bits
     * (c : low 12 of m) as the top of a 52-bit fraction (mc_rng.hpp: words_to_normals; MC_STREAM_VERSION 2) */
    uint32_t W[12];
    philox4x32_10((uint32_t)(unit >> 32), (uint32_t)unit, 3 * block, domain, (uint32_t)seed, (uint32_t)(seed >> 32), W);
    philox4x32_10((uint32_t)(unit >> 32), (uint32_t)unit, 3 * block + 1, domain, (uint32_t)seed, (uint32_t)(seed >> 32), W + 4);
    philox4x32_10((uint32_t)(unit >> 32), (uint32_t)unit, 3 * block + 2, domain, (uint32_t)seed, (uint32_t)(seed >> 32), W + 8);
    for (int p = 0; p < 4; ++p) {
        const uint32_t a = W[3 * p], m = W[3 * p + 1], c = W[3 * p + 2];
        const double ua = ((double)(((uint64_t)a << 20) | (m >> 12)) + 0.5) * 0x1p-52;
        const double ub = ((double)((((uint64_t)c << 12) | (m & 0xfffu)) << 8) + 0.5) * 0x1p-52;
        const double radius = sqrt(-2.0 * log(ua)), ang = 2.0 * M_PI * ub;
        z[2 * p] = radius * cos(ang);
        z[2 * p + 1] = radius * sin(ang);
    }
#endif
}

/* the seed of the next call: the *_ex entry points set it for their own call, everything else reads MC_SEED */
static int g_seed_given;
static uint64_t g_seed;
static uint64_t seed_from_env(void)
{
    if (g_seed_given)
        return g_seed;
    const char *s = getenv("MC_SEED");
    return s ? strtoull(s, NULL, 0) : MC_DEFAULT_SEED;
}

/* MC_ANTITHETIC=1: the antithetic-variates estimator of the GPU engine (mc_context_set_antithetic):
 * the sample of a path is the mean of its value at z and at -z.  Read once per call. */
static int g_antithetic;
/* MC_CONTROL_VARIATE=1 (host_basketOpt only): the geometric-basket control variate of the GPU engine
 * (mc_context_set_control_variate); the closed-form mean is added back in host_basketOpt. */
static int g_control;

/* ---- how many threads ---------------------------------------------------------------------------
 * MC_HOST_THREADS if set; otherwise OpenMP's default capped by the CPU time the container is actually granted
 * (cgroup v2 cpu.max, or v1 cfs quota / period): a pod that sees 256 hardware threads but may use 16 CPUs' worth of
 * time runs HALF as fast on 256 threads as on 16 (throttling: profiles/r02_host_twin_threads.log). */
static int cgroup_cpu_limit(void)
{
    long long quota = -1, period = 100000;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32] = "";
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0)
            quota = atoll(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
        if (fscanf(f, "%lld", &quota) != 1)
            quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) {
            if (fscanf(f, "%lld", &period) != 1)
                period = 100000;
            fclose(f);
        }
    }
    if (quota <= 0 || period <= 0)
        return 0;   /* no limit */
    return (int)((quota + period - 1) / period);
}

/* threads the next host_* call will use (exported for the benchmark's record; not in the reference) */
int mc_host_threads(void)
{
    const char *cap = getenv("MC_HOST_THREADS");
    if (cap && atoi(cap) > 0)
        return atoi(cap);
#ifdef _OPENMP
    int n = omp_get_num_procs();
    const int limit = cgroup_cpu_limit();
    if (limit > 0 && limit < n)
        n = limit;
    return n > 0 ? n : 1;
#else
    return 1;
#endif
}

/* ---- chunked, thread-count-independent accumulation ------------------------------------------ */
#define CHUNK 65536ll
typedef void (*chunk_fn)(const void *ctx, uint64_t seed, long long first, long long count, double out[2]);

static OptionValue simulate(chunk_fn fn, const void *ctx, long long paths, double discount)
{
    OptionValue v = {0, 0};
    if (paths < 2)
        return v;
    const long long n_chunks = (paths + CHUNK - 1) / CHUNK;
    double *part = (double *)malloc(sizeof(double) * 2 * (size_t)n_chunks);
    const uint64_t seed = seed_from_env();
    g_antithetic = getenv("MC_ANTITHETIC") && atoi(getenv("MC_ANTITHETIC"));
    g_control = getenv("MC_CONTROL_VARIATE") && atoi(getenv("MC_CONTROL_VARIATE"));
#ifndef MC_SINGLE_PRECISION
    g_normals_f32 = getenv("MC_F64_NORMALS") && !strcmp(getenv("MC_F64_NORMALS"), "f32");
    g_npb = g_normals_f32 ? 4 : NPB;
#endif
    {   /* MC_VERBOSE >= 2: what this call resolved its environment to, once per process (INTEGRATION.md section 1 holds the table) */
        static int told;
        const char *vb = getenv("MC_VERBOSE");
        if (!told && vb && atoi(vb) >= 2) {
            told = 1;
            const char *isa = getenv("MC_HOST_ISA");
            fprintf(stderr, "CPU twin config (libmchost): MC_HOST_THREADS -> %d threads (cgroup CPU limit %d, 0 = none) MC_HOST_ISA=%s MC_HOST_SCALAR=%d "
                    "MC_ANTITHETIC=%d MC_CONTROL_VARIATE=%d MC_F64_NORMALS=%s seed=0x%llx\n",
                    mc_host_threads(), cgroup_cpu_limit(), isa ? isa : "(widest the CPU has)", getenv("MC_HOST_SCALAR") != NULL, g_antithetic, g_control,
                    g_normals_f32 ? "f32" : "native", (unsigned long long)seed);
        }
    }
#ifdef _OPENMP
    omp_set_num_threads(mc_host_threads());
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (long long c = 0; c < n_chunks; ++c) {
        const long long first = c * CHUNK, count = (first + CHUNK <= paths) ? CHUNK : paths - first;
        fn(ctx, seed, first, count, part + 2 * c);
    }
    double sum = 0, sum2 = 0;
    for (long long c = 0; c < n_chunks; ++c)
        sum += part[2 * c], sum2 += part[2 * c + 1];
    free(part);
    double e, ci;
    mc_closing(sum, sum2, (uint64_t)paths, discount, &e, &ci);
    v.Expected = (mc_real)e;
    v.Confidence = (mc_real)ci;
    return v;
}

/* host_simd.c: the same loops over whole batches, structure-of-arrays, compiled once per vector ISA */
typedef void (*vanilla_units_fn)(uint64_t seed, uint64_t unit0, long long n_units, mc_real spot, mc_real strike, mc_real drift,
                                 mc_real vol, int antithetic, double out[2]);
typedef void (*basket_paths_fn)(uint64_t seed, uint64_t first, long long n_paths, const mc_real *p, const mc_real *d, const mc_real *v,
                                const mc_real *s0, const mc_real *w, mc_real strike, mc_real t, mc_real r, int antithetic, int control,
                                double out[2]);
typedef void (*cva_paths_fn)(uint64_t seed, uint64_t first, long long n_paths, mc_real s0, mc_real strike, mc_real r, mc_real v, mc_real t,
                             int n_dates, mc_real defint, mc_real lgd, int antithetic, double out[2]);
#define MC_SIMD_DECL(sfx)                                                                                                        \
    void mc_host_vanilla_units_##sfx(uint64_t, uint64_t, long long, mc_real, mc_real, mc_real, mc_real, int, double[2]);            \
    void mc_host_basket_paths_##sfx(uint64_t, uint64_t, long long, const mc_real *, const mc_real *, const mc_real *, const mc_real *, \
                                    const mc_real *, mc_real, mc_real, mc_real, int, int, double[2]);                               \
    void mc_host_cva_paths_##sfx(uint64_t, uint64_t, long long, mc_real, mc_real, mc_real, mc_real, mc_real, int, mc_real, mc_real,  \
                                 int, double[2]);
MC_SIMD_DECL(base)
MC_SIMD_DECL(avx2)
MC_SIMD_DECL(avx512)
#define SIMD_BATCH 256   /* host_simd.c BATCH */

typedef struct {
    vanilla_units_fn vanilla;
    basket_paths_fn basket;
    cva_paths_fn cva;
} simd_set;

/* NULL members never; MC_HOST_SCALAR=1 makes the callers skip the set altogether */
static const simd_set *simd(void)
{
    static const simd_set base = {mc_host_vanilla_units_base, mc_host_basket_paths_base, mc_host_cva_paths_base};
#if defined(__x86_64__) && defined(__GNUC__)
    static const simd_set avx2 = {mc_host_vanilla_units_avx2, mc_host_basket_paths_avx2, mc_host_cva_paths_avx2};
    static const simd_set avx512 = {mc_host_vanilla_units_avx512, mc_host_basket_paths_avx512, mc_host_cva_paths_avx512};
    const char *isa = getenv("MC_HOST_ISA");   /* "base", "avx2", "avx512": tests and A/B runs */
    if (isa && !strcmp(isa, "base"))
        return &base;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && !(isa && !strcmp(isa, "avx2")))
        return &avx512;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"))
        return &avx2;
#endif
    return &base;
}

/* vanilla: MonteCarloKernel.cu:67-71 */
static void vanilla_chunk(const void *ctx, uint64_t seed, long long first, long long count, double out[2])
{
    const OptionData *o = (const OptionData *)ctx;
    const mc_real drift = (mc_real)(((double)o->r - 0.5 * (double)o->v * (double)o->v) * (double)o->t);
    const mc_real vol = (mc_real)((double)o->v * sqrt((double)o->t));
    double s = 0, s2 = 0;
    mc_real z[MAX_NPB];
    uint64_t have = (uint64_t)-1;
    long long i0 = 0;
    const uint64_t npb = (uint64_t)g_npb;
    if (first % NPB == 0 && !getenv("MC_HOST_SCALAR") && !g_normals_f32) {   /* whole batches of units: the vectorised form (host_simd.c) */
        const long long units = (count / NPB) / SIMD_BATCH * SIMD_BATCH;
        if (units > 0) {
            double part[2];
            simd()->vanilla(seed, (uint64_t)first / NPB, units, o->s, o->k, drift, vol, g_antithetic, part);
            s = part[0], s2 = part[1];
            i0 = units * NPB;
        }
    }
    for (long long i = i0; i < count; ++i) {
        const uint64_t p = (uint64_t)(first + i), unit = p / npb;
        if (unit != have)
            block_normals(seed, MC_DOMAIN_VANILLA, unit, 0, z), have = unit;
        const mc_real v = o->s * R_EXP(drift + vol * z[p % npb]) - o->k;
        mc_real payoff = v > 0 ? v : 0;
        if (g_antithetic) {
            const mc_real vm = o->s * R_EXP(drift - vol * z[p % npb]) - o->k;
            payoff = (mc_real)0.5 * (payoff + (vm > 0 ? vm : 0));
        }
        const double pay = (double)payoff;
        s += pay, s2 += pay * pay;
    }
    out[0] = s, out[1] = s2;
}

/* basket: MonteCarloKernel.cu:74-101 (lower triangle of the factor only) */
static void basket_chunk(const void *ctx, uint64_t seed, long long first, long long count, double out[2])
{
    const MultiOptionData *o = (const MultiOptionData *)ctx;
    const mc_real sqrt_t = (mc_real)sqrt((double)o->t);
    const int npb = g_npb, nblk = (N + npb - 1) / npb;
    mc_real g[N + MAX_NPB];
    double s = 0, s2 = 0, wsum = 0;
    for (int a = 0; a < N; ++a)
        wsum += (double)o->w[a];
    long long i0 = 0;
    if (!getenv("MC_HOST_SCALAR") && !g_normals_f32 && count >= SIMD_BATCH) {   /* whole batches of paths: the vectorised form (host_simd.c) */
        double part[2];
        i0 = count / SIMD_BATCH * SIMD_BATCH;
        simd()->basket(seed, (uint64_t)first, i0, &o->p[0][0], o->d, o->v, o->s, o->w, o->k, o->t, o->r, g_antithetic, g_control, part);
        s = part[0], s2 = part[1];
    }
    for (long long i = i0; i < count; ++i) {
        for (int b = 0; b < nblk; ++b)
            block_normals(seed, MC_DOMAIN_BASKET, (uint64_t)(first + i), (uint32_t)b, g + b * npb);
        mc_real payoff = 0;
        for (int sign = 1; sign >= (g_antithetic ? -1 : 1); sign -= 2) {
            mc_real basket = 0, lg = (mc_real)log(wsum);
            for (int a = 0; a < N; ++a) {
                mc_real bt = 0;
                for (int b = 0; b <= a; ++b)
                    bt += o->p[a][b] * ((mc_real)sign * g[b]);
                bt += o->d[a];
                const mc_real mu = (mc_real)(((double)o->r - 0.5 * (double)o->v[a] * (double)o->v[a]) * (double)o->t);
                const mc_real x = mu + o->v[a] * bt * sqrt_t;
                basket += o->s[a] * R_EXP(x) * o->w[a];
                lg += (mc_real)((double)o->w[a] / wsum) * (R_LOG(o->s[a]) + x);
            }
            const mc_real v = basket - o->k;
            payoff += v > 0 ? v : 0;
            if (g_control) {
                const mc_real gv = R_EXP(lg) - o->k;
                payoff -= gv > 0 ? gv : 0;
            }
        }
        if (g_antithetic)
            payoff *= (mc_real)0.5;
        const double pay = (double)payoff;
        s += pay, s2 += pay * pay;
    }
    out[0] = s, out[1] = s2;
}

/* CVA, device ordering: MonteCarloKernel.cu:241-262; product semantics of DESIGN.md 4.4 */
static void cva_chunk(const void *ctx, uint64_t seed, long long first, long long count, double out[2])
{
    const CVA *c = (const CVA *)ctx;
    const OptionData *o = &c->option;
    const mc_real dt = o->t / c->n;
    const mc_real step_drift = (mc_real)(((double)o->r - 0.5 * (double)o->v * (double)o->v) * (double)dt);
    const mc_real step_vol = (mc_real)((double)o->v * sqrt((double)dt));
    double s = 0, s2 = 0;
    mc_real z[MAX_NPB];
    long long i0 = 0;
    const int npb = g_npb;
    if (!getenv("MC_HOST_SCALAR") && !g_normals_f32 && count >= SIMD_BATCH) {   /* whole batches of paths: the vectorised form (host_simd.c) */
        double part[2];
        i0 = count / SIMD_BATCH * SIMD_BATCH;
        simd()->cva(seed, (uint64_t)first, i0, o->s, o->k, o->r, o->v, o->t, c->n, c->defInt, c->lgd, g_antithetic, part);
        s = part[0], s2 = part[1];
    }
    for (long long i = i0; i < count; ++i) {
        mc_real spot = o->s, mirror = o->s, ttm = o->t, acc = 0;
        for (int j = 1; j <= c->n; ++j) {
            const double t_prev = (double)dt * (j - 1), t_now = (double)dt * j;
            const mc_real dpd = (mc_real)(-exp(-(double)c->defInt * t_prev) * expm1(-(double)c->defInt * (t_now - t_prev)));
            mc_real ee = 0;
            ttm -= dt;
            if (ttm >= 0) {
                const int idx = j - 1;
                if (idx % npb == 0)
                    block_normals(seed, MC_DOMAIN_CVA, (uint64_t)(first + i), (uint32_t)(idx / npb), z);
                spot = spot * R_EXP(step_drift + step_vol * z[idx % npb]);
                mirror = mirror * R_EXP(step_drift - step_vol * z[idx % npb]);
                for (int leg = 0; leg < (g_antithetic ? 2 : 1); ++leg) {
                    const mc_real sx = leg ? mirror : spot;
                    if (ttm == 0)
                        ee += sx > o->k ? sx - o->k : 0;
                    else
                        ee += bs_call(sx, o->k, o->r, o->v, ttm);
                }
                if (g_antithetic)
                    ee *= (mc_real)0.5;
            }
            acc += dpd * ee;
        }
        acc *= c->lgd;
        s += (double)acc, s2 += (double)acc * (double)acc;
    }
    out[0] = s, out[1] = s2;
}

OptionValue host_vanillaOpt(OptionData option, int path)
{
    return simulate(vanilla_chunk, &option, path, exp(-(double)option.r * (double)option.t));
}

OptionValue host_basketOpt(MultiOptionData *option, int path)
{
    const double disc = exp(-(double)option->r * (double)option->t);
    OptionValue v = simulate(basket_chunk, option, path, disc);
    if (g_control) { /* simulated: payoff - control; add the control's closed-form mean back */
        double mean = 0;
#ifdef MC_SINGLE_PRECISION
        const mc_basket_f32 b = {N, option->s, option->v, &option->p[0][0], option->d, option->w, option->k, option->t, option->r};
        if (mc_basket_control_mean_f32(&b, &mean) != MC_OK) {
#else
        const mc_basket_f64 b = {N, option->s, option->v, &option->p[0][0], option->d, option->w, option->k, option->t, option->r};
        if (mc_basket_control_mean_f64(&b, &mean) != MC_OK) {
#endif
            fprintf(stderr, "Error in host_basketOpt: %s\n", mc_last_error());
            exit(1);
        }
        v.Expected = (mc_real)((double)v.Expected + disc * mean);
    }
    return v;
}

OptionValue host_cvaEquityOption(CVA *cva, int path)
{
    return simulate(cva_chunk, cva, path, 1.0);
}

/* ---- explicit-seed variants (not in the reference, whose API has no seed parameter: SURVEY 8b "RNG contract").
 * Not re-entrant, like the rest of this file's entry points. ---- */
OptionValue host_vanillaOpt_ex(OptionData option, int path, uint64_t seed)
{
    g_seed_given = 1, g_seed = seed;
    const OptionValue v = host_vanillaOpt(option, path);
    g_seed_given = 0;
    return v;
}

OptionValue host_basketOpt_ex(MultiOptionData *option, int path, uint64_t seed)
{
    g_seed_given = 1, g_seed = seed;
    const OptionValue v = host_basketOpt(option, path);
    g_seed_given = 0;
    return v;
}

OptionValue host_cvaEquityOption_ex(CVA *cva, int path, uint64_t seed)
{
    g_seed_given = 1, g_seed = seed;
    const OptionValue v = host_cvaEquityOption(cva, path);
    g_seed_given = 0;
    return v;
}
