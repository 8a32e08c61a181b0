/*
 * mc_mi355x_test.h -- TEST HOOKS of libmc_mi355x.so.  NOT part of the drop-in surface (include/mc_mi355x.h is).
 *
 * The same shared library exports them, so the tests exercise the shipping binary; nothing here is needed to price
 * anything, speed is irrelevant, and two of the switches reproduce behaviour of the REFERENCE's CPU path that the product
 * deliberately does not have -- which is why they are kept out of the public header.
 *   - the simulation kernels on a caller-supplied normal stream (mc_*_from_normals_*): how the reference's own glibc
 *     normal stream is pushed through the HIP hot kernels and compared with numbers the compiled reference printed;
 *   - generator dumps: Philox/XORWOW normals and words (mc_normals_*, mc_xorwow_words, mc_grid_normals);
 *   - the launch-geometry mode's two forms side by side: mc_context_set_grid_form, per-path dumps mc_*_paths_grid_*.
 */
#ifndef MC_MI355X_TEST_H_
#define MC_MI355X_TEST_H_

#include "mc_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- generator dumps ---------------------------------------------------------------------------------------------- */
/* The generator alone (tests): words_each consecutive 32-bit outputs of each of the XORWOW subsequences
 * first_subsequence .. first_subsequence + n_subsequences - 1 for `seed`, subsequence-major, to the HOST array h_out. */
int mc_xorwow_words(mc_context *ctx, uint64_t seed, uint64_t first_subsequence, uint32_t n_subsequences,
                    uint32_t words_each, uint32_t *h_out);

/* The normals of Philox blocks (unit = first_unit .. first_unit+n_units-1, block, domain):
 * 4 per unit in f32, 8 per unit in f64 (4 under MC_NORMALS_F32), written unit-major to the HOST array h_out.  In f64 block b
 * of the stream is Philox blocks 3b .. 3b + 2 (MC_STREAM_VERSION 2). */
int mc_normals_f32(mc_context *ctx, uint64_t seed, uint32_t domain, uint64_t first_unit,
                   uint64_t n_units, uint32_t block, float *h_out);
int mc_normals_f64(mc_context *ctx, uint64_t seed, uint32_t domain, uint64_t first_unit,
                   uint64_t n_units, uint32_t block, double *h_out);

/* The first `count` normals of every thread's stream: h_out[(b * num_threads + t) * count + k] (tests). */
int mc_grid_normals(mc_context *ctx, int num_blocks, int num_threads, uint32_t count, float *h_out);


/* ---- test hooks: the simulation kernels on a caller-supplied normal stream -------------------------------------
 * NOT part of the drop-in surface; speed irrelevant.  They run the pricing call's own simulation kernel for that size --
 * payoff, per-lane sums and fp32 flushes, DPP/LDS reduction, last-arriver final reduction -- instantiated with a
 * generator policy that READS the normals from HBM instead of drawing them, so that the reference's normal stream
 * (glibc rand() + Box-Muller, MonteCarloHost.c:111-121) can be pushed through the HIP path and the result compared with
 * numbers the compiled reference printed (tests/test_gpu_from_normals.py, tests/golden/ref_mc.json).
 *   h_normals  HOST array.  vanilla: n_paths values, path i uses h_normals[i].  basket: n_paths * opt->n, path i's
 *              normals in drawing order (MonteCarloHost.c:150-161).  CVA: n_paths * n_grid, date j of path i at
 *              [i * n_grid + j - 1].
 *   h_values   HOST array of n_paths per-path values (undiscounted), or NULL.
 *   flags      MC_FROM_NORMALS_NO_VOL (basket): the diffusion without the volatility -- the model the reference's dp
 *              CPU path computes (dp/MonteCarloHost.c:180); only its goldens need it.
 *              MC_FROM_NORMALS_HOST_ORDER (CVA): the reference CPU loop's ordering, exposure of date j at the spot of
 *              date j - 1 (dp/MonteCarloHost.c:254-261).
 * Kernels: vanilla -- the hot kernels (whole units) + the masked kernel (a partial last unit, per-path values);
 * basket -- 3 and 4 assets: the kernel-argument kernels, 16: the tiled kernels, otherwise the generic kernel;
 * CVA -- cva_kernel.  Plain estimator, at most 2^26 paths. */
#define MC_FROM_NORMALS_NO_VOL 1
#define MC_FROM_NORMALS_HOST_ORDER 2
int mc_vanilla_from_normals_f32(mc_context *ctx, const mc_option_f32 *opt, const float *h_normals, uint64_t n_paths,
                                float *h_values, mc_result *out);
int mc_vanilla_from_normals_f64(mc_context *ctx, const mc_option_f64 *opt, const double *h_normals, uint64_t n_paths,
                                double *h_values, mc_result *out);
int mc_basket_from_normals_f32(mc_context *ctx, const mc_basket_f32 *opt, const float *h_normals, uint64_t n_paths,
                               int flags, float *h_values, mc_result *out);
int mc_basket_from_normals_f64(mc_context *ctx, const mc_basket_f64 *opt, const double *h_normals, uint64_t n_paths,
                               int flags, double *h_values, mc_result *out);
int mc_cva_from_normals_f32(mc_context *ctx, const mc_cva_f32 *cva, const float *h_normals, uint64_t n_paths,
                            int flags, float *h_values, mc_result *out);
int mc_cva_from_normals_f64(mc_context *ctx, const mc_cva_f64 *cva, const double *h_normals, uint64_t n_paths,
                            int flags, double *h_values, mc_result *out);

/* ---- launch-geometry mode (mc_*_run_grid_*, mc_mi355x.h): the two forms side by side ------------------------------------
 * MC_GRID_FORM_AUTO (default): the fused kernels where they exist, the staged form otherwise.  _STAGED: always round 3's
 * staged form (normals through HBM, priced by the engine's kernels) -- the checker.  _FUSED: the fused kernels or
 * MC_ERR_UNSUPPORTED.  mc_*_paths_grid_*: the per-path values (undiscounted) of such a call, in the call's path order
 * (path p = block * paths_per_block + i; thread i mod num_threads of the block prices it), n <= 2^26, to the HOST array
 * h_out -- the two forms must give the same bits (tests/test_gpu_grid.py). */
enum { MC_GRID_FORM_AUTO = 0, MC_GRID_FORM_STAGED = 1, MC_GRID_FORM_FUSED = 2 };
int mc_context_set_grid_form(mc_context *ctx, int form);
int mc_vanilla_paths_grid_f32(mc_context *ctx, const mc_option_f32 *opt, int num_blocks, int num_threads, uint64_t paths_per_block, float *h_out);
int mc_vanilla_paths_grid_f64(mc_context *ctx, const mc_option_f64 *opt, int num_blocks, int num_threads, uint64_t paths_per_block, double *h_out);
int mc_basket_paths_grid_f32(mc_context *ctx, const mc_basket_f32 *opt, int num_blocks, int num_threads, uint64_t paths_per_block, float *h_out);
int mc_basket_paths_grid_f64(mc_context *ctx, const mc_basket_f64 *opt, int num_blocks, int num_threads, uint64_t paths_per_block, double *h_out);
int mc_cva_paths_grid_f32(mc_context *ctx, const mc_cva_f32 *cva, int num_blocks, int num_threads, uint64_t paths_per_block, float *h_out);
int mc_cva_paths_grid_f64(mc_context *ctx, const mc_cva_f64 *cva, int num_blocks, int num_threads, uint64_t paths_per_block, double *h_out);

#ifdef __cplusplus
}
#endif
#endif /* MC_MI355X_TEST_H_ */
