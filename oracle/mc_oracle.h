/*
 * mc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Monte Carlo path (marcomatteo/MonteCarloCUDA,
 * {double,single}_precision/MonteCarloHost.c and the device formulas of MonteCarloKernel.cu).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (montecarlocuda_amd/csrc) never links, imports or calls it.
 *
 * Parity status: PINNED.  The "host" family below is bit-checked against the unmodified
 * reference MonteCarloHost.c compiled in the dev container (oracle/_ref, recipe in
 * oracle/Makefile) and against the golden vectors committed under tests/golden/ that were
 * generated from it (tests/golden/gen_golden.py).
 *
 * Two families, both for f32 and f64 (suffix _f32 / _f64):
 *
 *  orc_host_*  : the reference CPU algorithm on the reference CPU random stream
 *                (glibc rand() + cosine-branch Box-Muller, srand(seed)).  Bit-for-bit equal
 *                to the compiled reference under a pinned time() seed -- "hop A".
 *
 *  orc_dev_*   : the reference DEVICE formulas (MonteCarloKernel.cu:67-129,222-262) evaluated
 *                on the product's counter-based Philox4x32-10 stream, path by path, in the
 *                stated precision with libm.  This is what the HIP kernels are compared
 *                against on identical counters -- "hop B".  `antithetic` != 0 selects the
 *                antithetic-variates estimator (not in the reference): the
 *                sample of path p is the mean of the values at its normals z and at -z.
 */
#ifndef MC_ORACLE_H_
#define MC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* All fields are doubles so one struct serves both precisions (an f32 value is exact in f64). */
typedef struct {
    double expected;    /* price (discounted mean) or CVA          */
    double confidence;  /* 95 % half width, 1.96 s / sqrt(n)        */
    double sum;         /* sum of per-path values                   */
    double sum2;        /* sum of squared per-path values           */
    long long n;        /* paths simulated                          */
} orc_result;

/* Stream domains: counter word 3 of the Philox block (keeps product streams disjoint). */
#define ORC_DOMAIN_VANILLA 1u
#define ORC_DOMAIN_BASKET  2u
#define ORC_DOMAIN_CVA     3u

/* Philox4x32-10 (Salmon et al., SC'11; same generator rocRAND/cuRAND ship). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* XORWOW (the reference's cuRAND generator, dp/MonteCarloKernel.cu:285-290), rocRAND's seeding and subsequence layout:
 * state[0..4] = xorshift words, state[5] = Weyl word.  orc_xorwow_begin/_end switch the orc_dev_* family to
 * one-XORWOW-sequence-per-lane normals (see mc_oracle.c). */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint32_t state[6]);
uint32_t orc_xorwow_next(uint32_t state[6]);
void orc_xorwow_jump_column(int i, int c, uint32_t out[5]);
void orc_grid_normals(uint32_t num_blocks, uint32_t block, uint32_t thread, uint32_t count, float *out);
void orc_xorwow_begin(uint64_t seed, uint64_t subsequence_base, uint32_t lanes, uint64_t unit0);
void orc_xorwow_end(void);

/* fp64 orc_dev_* family: 1 = draw FOUR fp32 normals per block and widen them (`double z = curand_normal(...)`,
 * dp/MonteCarloKernel.cu:68,78,250), 0 (default) = native fp64 normals.  Twin of mc_context_set_normals. */
void orc_set_normals_f32(int on);

/* flags of orc_dev_cva_on_normals_* (bridge tests): see mc_oracle_impl.h */
#define ORC_CVA_HOST_ORDER 1
#define ORC_CVA_REF_DP 2
#define ORC_CVA_REF_T0 4

/* Closing formulas shared by every estimator: reference MonteCarloHost.c:220-228 /
 * MonteCarloKernel.cu:420-423 (discount = exp(-rT) for prices, 1 for CVA :466). fp64. */
void orc_closing(double sum, double sum2, long long n, double discount,
                 double *expected, double *confidence);

#define ORC_DECL(X, REAL)                                                                        \
    /* deterministic pieces */                                                                   \
    REAL orc_cnd_##X(REAL d);                                                                    \
    REAL orc_bs_call_##X(REAL s, REAL k, REAL r, REAL v, REAL t);                                \
    void orc_chol_##X(int n, const REAL *c, REAL *a);                                            \
    int orc_factor_from_cov_##X(int n, const REAL *cov, REAL *v, REAL *corr, REAL *a);           \
    /* reference CPU stream */                                                                   \
    void orc_host_uniforms_##X(unsigned seed, int count, REAL *out);                             \
    void orc_host_gaussians_##X(unsigned seed, int count, REAL *out);                            \
    void orc_host_vanilla_##X(REAL s, REAL k, REAL r, REAL v, REAL t, int paths, unsigned seed,  \
                              orc_result *out);                                                  \
    void orc_host_basket_##X(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,  \
                             const REAL *w, REAL k, REAL t, REAL r, int paths, unsigned seed,    \
                             int vol_in_diffusion, orc_result *out);                             \
    void orc_host_cva_##X(REAL s, REAL k, REAL r, REAL v, REAL t, REAL defint, REAL lgd,         \
                          int n_grid, int paths, unsigned seed, orc_result *out);                \
    /* the same with every path's value handed back (test tap; arithmetic untouched) */          \
    void orc_host_vanilla_paths_##X(REAL s, REAL k, REAL r, REAL v, REAL t, int paths,           \
                                    unsigned seed, REAL *payoffs, orc_result *out);              \
    void orc_host_basket_paths_##X(int n, const REAL *s, const REAL *v, const REAL *p,           \
                                   const REAL *d, const REAL *w, REAL k, REAL t, REAL r,         \
                                   int paths, unsigned seed, int vol_in_diffusion,               \
                                   REAL *payoffs, orc_result *out);                              \
    void orc_host_cva_paths_##X(REAL s, REAL k, REAL r, REAL v, REAL t, REAL defint, REAL lgd,   \
                                int n_grid, int paths, unsigned seed, REAL *values,              \
                                orc_result *out);                                                \
    /* the reference's sequential REAL accumulation + closing on given per-path values */        \
    void orc_ref_close_##X(const REAL *values, int paths, int discounted, REAL r, REAL t,        \
                           orc_result *out);                                                     \
    /* the DEVICE formulas on a caller-supplied normal stream (bridge to the compiled reference) */ \
    int orc_dev_npb_##X(void);                                                                   \
    void orc_dev_vanilla_on_normals_##X(REAL s, REAL k, REAL r, REAL v, REAL t, const REAL *z,   \
                                        uint64_t n_paths, int antithetic, REAL *payoffs,         \
                                        orc_result *out);                                        \
    void orc_dev_basket_on_normals_##X(int n, const REAL *s, const REAL *v, const REAL *p,       \
                                       const REAL *d, const REAL *w, REAL k, REAL t, REAL r,     \
                                       const REAL *g, uint64_t n_paths, int mode,                \
                                       REAL *payoffs, orc_result *out);                          \
    void orc_dev_cva_on_normals_##X(REAL s, REAL k, REAL r, REAL v, REAL t, REAL defint,         \
                                    REAL lgd, int n_grid, const REAL *z, uint64_t n_paths,       \
                                    int antithetic, int flags, REAL *values, orc_result *out);   \
    /* product stream (Philox), device formulas */                                               \
    void orc_dev_normals_##X(uint64_t seed, uint32_t domain, uint64_t unit, uint32_t block,      \
                             REAL *z);                                                           \
    void orc_dev_vanilla_##X(REAL s, REAL k, REAL r, REAL v, REAL t, uint64_t seed,              \
                             uint64_t first_path, uint64_t n_paths, int antithetic,              \
                             REAL *payoffs, orc_result *out);                                    \
    void orc_dev_vanilla_greeks_##X(REAL s, REAL k, REAL r, REAL v, REAL t, uint64_t seed,       \
                                    uint64_t first_path, uint64_t n_paths, orc_result *out3);    \
    void orc_dev_vanilla_greeks_lr_##X(REAL s, REAL k, REAL r, REAL v, REAL t, uint64_t seed,    \
                                       uint64_t first_path, uint64_t n_paths, orc_result *out3); \
    void orc_dev_basket_greeks_##X(int n, const REAL *s, const REAL *v, const REAL *p,           \
                                   const REAL *d, const REAL *w, REAL k, REAL t, REAL r,         \
                                   uint64_t seed, uint64_t first_path, uint64_t n_paths,         \
                                   orc_result *out);                                             \
    void orc_dev_cva_greeks_##X(REAL s, REAL k, REAL r, REAL v, REAL t, REAL defint, REAL lgd,   \
                                int n_grid, uint64_t seed, uint64_t first_path,                  \
                                uint64_t n_paths, orc_result *out3);                             \
    int orc_dev_basket_greeks_lr_##X(int n, const REAL *s, const REAL *v, const REAL *p,         \
                                     const REAL *d, const REAL *w, REAL k, REAL t, REAL r,       \
                                     uint64_t seed, uint64_t first_path, uint64_t n_paths,       \
                                     orc_result *out);                                           \
    void orc_dev_cva_greeks_lr_##X(REAL s, REAL k, REAL r, REAL v, REAL t, REAL defint,          \
                                   REAL lgd, int n_grid, uint64_t seed, uint64_t first_path,     \
                                   uint64_t n_paths, orc_result *out3);                          \
    void orc_dev_basket_##X(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,   \
                            const REAL *w, REAL k, REAL t, REAL r, uint64_t seed,                \
                            uint64_t first_path, uint64_t n_paths, int mode,                     \
                            REAL *payoffs, orc_result *out);                                     \
    double orc_basket_control_mean_##X(int n, const REAL *s, const REAL *v, const REAL *p,       \
                                       const REAL *d, const REAL *w, REAL k, REAL t, REAL r);    \
    void orc_dev_cva_##X(REAL s, REAL k, REAL r, REAL v, REAL t, REAL defint, REAL lgd,          \
                         int n_grid, uint64_t seed, uint64_t first_path, uint64_t n_paths,       \
                         int antithetic, REAL *values, orc_result *out);

ORC_DECL(f32, float)
ORC_DECL(f64, double)

/* Normals per block of the stream: f32 4 (one Philox block, one u32 per uniform); f64 8 (THREE Philox blocks, 96 bits per
 * Box-Muller pair: stream version 2, include/mc_mi355x.h MC_STREAM_VERSION). */
#define ORC_NORMALS_PER_BLOCK_F32 4
#define ORC_NORMALS_PER_BLOCK_F64 8

#ifdef __cplusplus
}
#endif
#endif /* MC_ORACLE_H_ */
