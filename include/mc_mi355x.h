/*
 * mc_mi355x.h -- C ABI of libmc_mi355x.so, the MI355X (gfx950) Monte Carlo engine.
 *
 * Plain C: pointers, sizes and PODs only.  This is the boundary a maintainer of the reference
 * (marcomatteo/MonteCarloCUDA) binds to in place of MonteCarloKernel.cu; INTEGRATION.md shows
 * the binding.  Each entry point cites the reference interface it replaces ("dp/" =
 * double_precision/, the single_precision/ twin is the same code with float).
 *
 * Shape of the API
 *   - an opaque context per GPU owns every device buffer (the reference re-allocates and
 *     re-seeds on every call: dp/MonteCarloKernel.cu:296-342,344-363 -- that fixed cost is
 *     what a persistent context removes);
 *   - *_launch_* enqueue the simulation of the path range [first_path, first_path+n_paths) on
 *     a caller-supplied HIP stream and leave {sum, sum2, n} (three doubles) in device memory:
 *     the 24-byte payload of the multi-GPU all-reduce.  No host synchronisation;
 *   - *_run_* are the synchronous forms: launch, wait, close the estimator on the host;
 *   - *_paths_* return per-path values for a (small) path range: used by the parity tests.
 * Sizes: first_path and n_paths are 64-bit; ONE call covers at most 8 segments of 2^31 units (a unit = 4 / 8 vanilla paths in
 * f32 / f64, one basket or CVA path), none crossing a multiple of 2^32 units: up to 6.9e10 / 1.4e11 vanilla paths and
 * 1.7e10 basket or CVA paths per call (MC_ERR_INVALID beyond: "split it").  Larger jobs are several calls on consecutive
 * ranges; their triples add (mc_closing takes the sums).
 * One launch per call: the last workgroup to arrive adds the per-workgroup (sum, sum2) pairs and writes the triple.
 * Contract of a context: it owns one pair buffer, one ticket block and one constant table, so its calls execute one
 * after the other; calls on one stream are ordered by the stream, a call on a DIFFERENT stream than the context's
 * previous call is ordered behind it by the library (event + wait).  For overlap use one context per stream.  A context
 * is not thread-safe: one host thread per context.  Several GPUs from one process: include/mc_multi.h.
 *
 * Random numbers: Philox4x32-10, key = 64-bit seed, counter = {unit_hi, unit_lo, block,
 * domain} (unit = 64-bit index of a path or of a block of vanilla paths: 4 in f32, 8 in f64).  A path's normals depend only on
 * (seed, global path index), never on the launch geometry or on how a range is split over GPUs.  Layout per product:
 * DESIGN.md "RNG".  XORWOW, the reference's generator, is selectable: mc_context_set_generator.
 *
 * Precision suffix: _f32 simulates in float (sums are still accumulated in double),
 * _f64 simulates in double.  Struct layouts equal the reference's of that precision.
 */
#ifndef MC_MI355X_H_
#define MC_MI355X_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status ------------------------------------------------------------------------- */
enum {
    MC_OK = 0,
    MC_ERR_INVALID = 1,   /* bad argument (message in mc_last_error)                    */
    MC_ERR_HIP = 2,       /* a HIP runtime call failed                                   */
    MC_ERR_NO_DEVICE = 3, /* no usable gfx950 device                                     */
    MC_ERR_UNSUPPORTED = 4
};
/* Text of the last failure on this thread ("" if none). */
const char *mc_last_error(void);

/* Default seed of the legacy entry points (the reference's device seed is fixed too:
 * dp/MonteCarloKernel.cu:289). */
#define MC_DEFAULT_SEED 0x4D435F4D49333535ull

/* Layout of the random streams (which counter yields which normal).  Results for a given seed are comparable only between
 * builds of the same version.  1: rounds 1-2 (fp64: one Philox block = 2 normals).  2: the fp64 streams draw EIGHT normals
 * from three consecutive Philox blocks -- 96 bits per Box-Muller pair: 52-bit radius, 44-bit angle;
 * a vanilla fp64 unit is 8 paths.  The fp32 streams are unchanged. */
#define MC_STREAM_VERSION 2

/* Stream domains (Philox counter word 3). */
#define MC_DOMAIN_VANILLA 1u
#define MC_DOMAIN_BASKET 2u
#define MC_DOMAIN_CVA 3u

#define MC_MAX_ASSETS 16         /* basket sizes whose constants can travel as kernel arguments: 1..16 */
#define MC_MAX_ASSETS_GENERIC 64 /* largest basket: sizes up to 32 have register-resident kernels, 33..64 a generic one */

/* ---- inputs (layouts == reference MonteCarlo.h of that precision) ------------------- */
typedef struct { float s, k, r, v, t; } mc_option_f32;   /* OptionData, sp/MonteCarlo.h:33-39 */
typedef struct { double s, k, r, v, t; } mc_option_f64;  /* OptionData, dp/MonteCarlo.h:32-38 */

/* Basket with a RUNTIME asset count (the reference fixes N at compile time,
 * dp/MonteCarlo.h:16,41-50).  p = row-major n x n lower-triangular Cholesky factor. */
typedef struct {
    int n;
    const float *s, *v, *p, *d, *w;
    float k, t, r;
} mc_basket_f32;
typedef struct {
    int n;
    const double *s, *v, *p, *d, *w;
    double k, t, r;
} mc_basket_f64;

/* CVA of one call: dp/MonteCarlo.h:57-65 without the unused `ns`. */
typedef struct { float defint, lgd; mc_option_f32 option; int n_grid; } mc_cva_f32;
typedef struct { double defint, lgd; mc_option_f64 option; int n_grid; } mc_cva_f64;

/* ---- outputs ------------------------------------------------------------------------ */
typedef struct {
    double expected;    /* discounted mean payoff (price) or CVA                           */
    double confidence;  /* 1.96 s / sqrt(n)        dp/MonteCarloKernel.cu:421-422          */
    double sum, sum2;   /* sum and sum of squares of the per-path values                   */
    uint64_t n;         /* paths simulated                                                 */
    float kernel_ms;    /* device time of the call's kernels (HIP events around them)       */
    float wall_ms;      /* host wall-clock of the whole call: entry to return, launch, wait and
                         * 24-byte read-back included (what the reference's drivers time around
                         * dev_*: dp/vanillaOpt.cu:77-83); context creation is NOT in it          */
} mc_result;

/* ---- context ------------------------------------------------------------------------ */
typedef struct mc_context mc_context;

int mc_device_count(void);
/* PCI bus id of a visible device ("0000:75:00.0"; buf of >= 13 bytes): which physical GPU an index is -- what
 * bench.py's roster and multiBench print for every rank / device of a multi-GPU run. */
int mc_device_pci_bus_id(int device, char *buf, int len);
/* Create the per-GPU context (replaces dp/MonteCarloKernel.cu:296 MonteCarlo_init).
 * blocks = simulation grid size; 0 picks the default (8 workgroups of 256 per CU).
 * Current device: every call that takes a context makes that context's device the calling thread's current HIP device
 * (hipSetDevice) and leaves it so -- a thread that also drives other devices through HIP re-selects its own afterwards.
 * (libmc_multi, which visits several devices per call, puts the caller's device back itself: mc_multi.h.) */
int mc_context_create(int device, int blocks, mc_context **out);
/* Replaces dp/MonteCarloKernel.cu:344 MonteCarlo_closing. */
void mc_context_destroy(mc_context *ctx);
int mc_context_device(const mc_context *ctx);
/* The non-blocking HIP stream the context owns (the synchronous *_run_* calls use it). */
void *mc_context_stream(const mc_context *ctx);
int mc_context_blocks(const mc_context *ctx);
/* Shape of the context's most recent simulation launch: workgroups and lanes per workgroup (0, 0 before the first).  The
 * committed PMC profiles describe launches of one shape; bench.py compares (profiles/pmc_traffic.json "grid_workgroups"). */
int mc_context_last_launch(const mc_context *ctx, int *workgroups, int *group_size);
/* Name / CU count / clock of the context's device, for logs. */
int mc_context_info(const mc_context *ctx, char *name, int name_len, int *compute_units, int *clock_mhz);

/* Estimator switch (not in the reference, whose estimator is plain Monte Carlo).
 * on != 0: antithetic variates -- the sample of path p is the mean of the payoff at its normals z
 * and at -z; n_paths then counts such pairs, the triple and the confidence interval refer to the
 * pair means.  Same random stream, same path indexing; about 1.2-1.9x the time per sample for a
 * 2.7-4x smaller variance on the reference's three products. */
int mc_context_set_antithetic(mc_context *ctx, int on);

/* Basket products only (not in the reference): geometric-basket control variate.
 * on != 0: the per-path value becomes payoff(arithmetic basket) - payoff(geometric basket
 * G = W prod_a S_a(T)^(w_a/W), W = sum w) on the same normals; G is lognormal, so E[max(G-K,0)] has
 * the closed form mc_basket_control_mean_* returns (fp64, undiscounted).  mc_basket_run_* adds it
 * back (expected = discount * (sum/n + mean)); users of mc_basket_launch_* do the same after their
 * all-reduce.  Needs w[a] > 0, s[a] > 0, k > 0.  Combines with antithetic variates.  Typical
 * variance reduction on the BASELINE baskets: ~150x (x2.5 more with antithetic). */
int mc_context_set_control_variate(mc_context *ctx, int on);

/* Device timing of the synchronous *_run_* calls.  on != 0 (default): the call's kernels are bracketed by two HIP
 * events and mc_result.kernel_ms reports them (replaces the reference's cudaEvent pairs, dp/MonteCarloKernel.cu:380-386),
 * the result comes back through a 24-byte copy and a stream synchronize.  on == 0: no events; the last workgroup of the
 * call writes the result straight into pinned host memory and the calling thread polls for it from user space --
 * about 10 us less host time per call (profiles/r02_call_latency.log); kernel_ms is then 0.  The legacy symbols run
 * with timing off unless MC_VERBOSE is set. */
int mc_context_set_timing(mc_context *ctx, int on);

/* Generator switch.  MC_RNG_PHILOX (default): Philox4x32-10, counter-based -- a path's normals depend
 * only on (seed, global path index).  MC_RNG_XORWOW: the reference's generator (cuRAND XORWOW, dp/MonteCarloKernel.cu:
 * 285-290 curand_init, :68,78,250 curand_normal), hand-written for gfx950, with rocRAND's seeding and subsequence
 * layout: lane l of a launch starts at rocrand_init(seed, subsequence_base + l, 0) -- one sequence per lane, as the
 * reference keeps one curandState per thread -- and draws four words per block of normals.  Word for word rocRAND's
 * sequence (tests); cuRAND 7.5's own seeding is not in the reference tree, so the reference's exact stream stays
 * "parity unpinned".  Consequences of a per-lane stream: the sample depends on (seed, subsequence_base, grid size,
 * position in the range) instead of the global path index, first_path only places the range; a call must fit one
 * launch (<= 2^31 units); ranks of a multi-GPU job need disjoint subsequence bases (mc_multi_* sets them); vanilla
 * calls run at Philox speed, baskets on the generic kernel; Greeks are Philox-only.  Subsequence numbers stay below 2^48. */
enum { MC_RNG_PHILOX = 0, MC_RNG_XORWOW = 1 };
int mc_context_set_generator(mc_context *ctx, int generator, uint64_t subsequence_base);
/* Normals of the fp64 kernels.  MC_NORMALS_NATIVE (default): true fp64 normals, two per Philox block (52-bit uniforms,
 * mc_math_f64.hpp).  MC_NORMALS_F32: the reference's own dp arithmetic -- `double z = curand_normal(...)`, a FLOAT normal
 * widened to double (dp/MonteCarloKernel.cu:68,78,250): four normals per Philox block through the hardware
 * fp32 transcendentals, everything downstream of the normal in fp64.  A different (coarser: 24-bit normals, |z| < 6.77)
 * stream than the default, same path indexing; about 1.4-1.7x the paths/s on the 16-asset basket and the 256-date CVA
 * (DESIGN.md).  Philox only; no effect on the _f32 entry points (their Greeks included); the fp64 Greeks entry points refuse it.  Also selected at
 * context creation by the environment variable MC_F64_NORMALS=f32 (how the legacy symbols get it). */
enum { MC_NORMALS_NATIVE = 0, MC_NORMALS_F32 = 1 };
int mc_context_set_normals(mc_context *ctx, int mode);

/* Where a call's per-workgroup (sum, sum2) pairs are added up (replaces the reference's D2H copy and host loop
 * over blocks, dp/MonteCarloKernel.cu:405,416-419).  fused != 0 (default): inside the simulation kernel, by the
 * last workgroup to arrive -- one launch per call.  fused == 0: by a second, one-workgroup launch (the A/B
 * baseline; also selected by the environment variable MC_FINISH=kernel at context creation).  Both add the pairs
 * in the same fixed order: the results are identical bits. */
int mc_context_set_finish(mc_context *ctx, int fused);

/* CVA: how many adjacent lanes share one path's dates (reference: one thread walks all N_GRID dates of its path,
 * dp/MonteCarloKernel.cu:241-262).  lanes == 0 (default): chosen per call from its size -- a large call keeps one lane per path
 * and hands only its last partial wave-trip (n mod 64 x 4 x CUs paths, when that is at most 60 % of a trip) to date-parallel
 * workgroups of the same launch; a small call (up to 7/4 wave-trips on a grid of 64 dates or more, 3/4 of one on a shorter
 * grid) runs date-parallel as a whole -- the reference driver's own 131 072 paths (dp/cvaOpt.cu:12-15) are two
 * trips and keep one lane per path.  lanes == 1: never (the one-lane-per-path kernel only).  lanes == 2, 4, ... 64:
 * the whole call date-parallel with that many lanes per path (capped by the number of 8-date chunks of the grid).  The
 * estimate differs between settings only by the association of two sums per path (1e-16-level in fp64, 1e-7 in fp32);
 * XORWOW calls (one sequence per lane) and grids whose table exceeds 48 KB of LDS always use one lane per path.
 * Also set at context creation by the environment variable MC_CVA_DATE_LANES. */
int mc_context_set_cva_date_lanes(mc_context *ctx, int lanes);
int mc_basket_control_mean_f32(const mc_basket_f32 *opt, double *mean);
int mc_basket_control_mean_f64(const mc_basket_f64 *opt, double *mean);

/* Where the time of the context's last synchronous call (mc_*_run_*, mc_*_run_grid_*) went -- the reference prints the same
 * stages from inside every call (RNG set-up dp/MonteCarloKernel.cu:317-323, allocations :326-341, kernel :380-386, copy :404-409,
 * closing :415-427); here they are returned, and the legacy symbols print them under MC_VERBOSE=1.  Consecutive host-clock
 * intervals, so setup + table_upload + launch + kernel + readback + closing = wall_ms (to the clamping of readback_ms at 0):
 *   setup_ms         work a repeated identical call does not do again: XORWOW jump matrices and start states (the reference's
 *                    randomSetup, paid there on EVERY call), launch-geometry states of a new (numBlocks, numThreads), buffer growth;
 *                    waited for on the device, so that it is not inside kernel_ms
 *   table_upload_ms  constant tables built on the host and uploaded when the inputs changed (CVA per-date rows, tiled basket matrix)
 *   launch_ms        the rest of the host time before the wait: folding the inputs, the launch calls themselves (the library's code
 *                    object is loaded by mc_context_create, not by the first launch: HIP would otherwise spend 7-10 ms inside it)
 *   kernel_ms        device time of the call's kernels (HIP events; 0 with timing off -- the kernel is then inside readback_ms), capped at
 *                    the host's wait (the opening event is stamped at once on an idle device, so whatever the host does inside
 *                    the launch call -- before round 5's preload: the code-object load -- would otherwise be counted twice)
 *   readback_ms      from the last launch call to the triple on the host, minus kernel_ms: launch latency, copy / poll
 *   closing_ms       price and confidence interval on the host
 *   context_create_ms  what mc_context_create took for this context (once; NOT part of wall_ms): HIP runtime start-up and the load
 *                    of the code object on the first context of a process (~240 ms on a fresh box, ~5 ms for a second context),
 *                    stream, buffers.  first_call = 1 on the context's first synchronous call. */
typedef struct {
    float setup_ms, table_upload_ms, launch_ms, kernel_ms, readback_ms, closing_ms, wall_ms, context_create_ms;
    int first_call;
} mc_call_stats;
int mc_context_last_call_stats(const mc_context *ctx, mc_call_stats *out);
/* The context's resolved configuration (device, grid, estimator, generator, kernel-family limits ...) as one line of text;
 * printed to stderr by mc_context_create when MC_VERBOSE >= 2. */
int mc_context_describe(const mc_context *ctx, char *buf, int len);

/* Sampled device timing of the simulation kernel (not the finishing kernel): every `every`-th
 * launch is bracketed by two HIP events on its launch stream (0 = off; at most 512 samples are
 * kept between reads).  Replaces the reference's cudaEvent pair around each launch
 * (dp/MonteCarloKernel.cu:380-386,393-399,447-453), returned instead of printed. */
int mc_context_profile(mc_context *ctx, int every);
/* Waits for the sampled launches; returns their count and summed duration, then resets. */
int mc_context_profile_read(mc_context *ctx, int *samples, double *total_ms);

/* ---- asynchronous launches ------------------------------------------------------------
 * d_triple: DEVICE pointer to 3 doubles, overwritten with {sum, sum2, n}.
 * stream  : hipStream_t passed as void*; NULL = the HIP null stream, as in any HIP API
 *           (mc_context_stream(ctx) is the context's own non-blocking stream).
 * Replace the kernel launch + D2H + host block-sum of dp/MonteCarloKernel.cu:365-419
 * (vanilla :381, basket :394) and :433-465 (CVA :448). */
int mc_vanilla_launch_f32(mc_context *ctx, const mc_option_f32 *opt, uint64_t seed,
                          uint64_t first_path, uint64_t n_paths, double *d_triple, void *stream);
int mc_vanilla_launch_f64(mc_context *ctx, const mc_option_f64 *opt, uint64_t seed,
                          uint64_t first_path, uint64_t n_paths, double *d_triple, void *stream);
int mc_basket_launch_f32(mc_context *ctx, const mc_basket_f32 *opt, uint64_t seed,
                         uint64_t first_path, uint64_t n_paths, double *d_triple, void *stream);
int mc_basket_launch_f64(mc_context *ctx, const mc_basket_f64 *opt, uint64_t seed,
                         uint64_t first_path, uint64_t n_paths, double *d_triple, void *stream);
int mc_cva_launch_f32(mc_context *ctx, const mc_cva_f32 *cva, uint64_t seed,
                      uint64_t first_path, uint64_t n_paths, double *d_triple, void *stream);
int mc_cva_launch_f64(mc_context *ctx, const mc_cva_f64 *cva, uint64_t seed,
                      uint64_t first_path, uint64_t n_paths, double *d_triple, void *stream);

/* Order `stream` behind everything `ctx` has enqueued so far (event on the stream of its last call, wait on `stream`):
 * how a consumer of d_triple on ANOTHER stream -- a copy, an RCCL all-reduce -- is sequenced after the launch without
 * a host synchronisation.  A no-op when `stream` is the stream of the last call. */
int mc_context_order(mc_context *ctx, void *stream);

/* Results of asynchronous launches straight into pinned host memory (what include/mc_multi.h reads its devices back with).
 * mc_context_arm_direct: the NEXT mc_*_launch_* call on `ctx` also has its last workgroup store {sum, sum2, n} into a
 * pinned, host-coherent slot the context owns -- n last, with system-scope release semantics.  *slot = the slot's host
 * address; its word 2 is preset to -1 (n is never negative): poll `(*slot)[2] != -1` from user space, then read the
 * three doubles.  No D2H copy command, no event, no sleeping synchronize; d_triple is still written.  One armed launch in
 * flight per context; a synchronous mc_*_run_* call before the launch cancels the arming; needs the fused final
 * reduction (the default).
 * mc_context_publish: enqueues on `stream` a one-lane kernel that copies the three doubles at d_src (device memory: e.g.
 * the output of an all-reduce enqueued on that stream before it) into a second slot of the context, same protocol.
 * ONE publish in flight per context as well: a second publish before the first slot has been read re-arms the same slot. */
int mc_context_arm_direct(mc_context *ctx, const volatile double **slot);
int mc_context_publish(mc_context *ctx, const double *d_src, void *stream, const volatile double **slot);

/* 1 when everything `ctx` has enqueued has completed, 0 while some of it is pending, -1 on error (hipStreamQuery of the
 * stream of its last call): a host may poll this from user space instead of sleeping in a synchronize. */
int mc_context_idle(mc_context *ctx);

/* ---- synchronous runs (launch + wait + closing formulas) --------------------------------
 * Replace dev_vanillaOpt / dev_basketOpt / dev_cvaEquityOption (dp/MonteCarloKernel.cu:
 * 500,483,517) with an explicit seed, a 64-bit path range and a status code. */
int mc_vanilla_run_f32(mc_context *ctx, const mc_option_f32 *opt, uint64_t seed,
                       uint64_t first_path, uint64_t n_paths, mc_result *out);
int mc_vanilla_run_f64(mc_context *ctx, const mc_option_f64 *opt, uint64_t seed,
                       uint64_t first_path, uint64_t n_paths, mc_result *out);
int mc_basket_run_f32(mc_context *ctx, const mc_basket_f32 *opt, uint64_t seed,
                      uint64_t first_path, uint64_t n_paths, mc_result *out);
int mc_basket_run_f64(mc_context *ctx, const mc_basket_f64 *opt, uint64_t seed,
                      uint64_t first_path, uint64_t n_paths, mc_result *out);
int mc_cva_run_f32(mc_context *ctx, const mc_cva_f32 *cva, uint64_t seed,
                   uint64_t first_path, uint64_t n_paths, mc_result *out);
int mc_cva_run_f64(mc_context *ctx, const mc_cva_f64 *cva, uint64_t seed,
                   uint64_t first_path, uint64_t n_paths, mc_result *out);

/* ---- pathwise Greeks of the vanilla call (the reference prices only) -------------
 * One pass, the pricing kernels' stream and path indexing: price, delta = dV/dS and vega = dV/dsigma
 * as discounted means of  I (S_T - K),  I S_T / S,  I S_T (sqrt(T) z - sigma T),  I = [S_T > K],
 * each with its own 95 % half-width.  Plain estimator only. */
typedef struct { mc_result price, delta, vega; } mc_vanilla_greeks;
int mc_vanilla_greeks_run_f32(mc_context *ctx, const mc_option_f32 *opt, uint64_t seed,
                              uint64_t first_path, uint64_t n_paths, mc_vanilla_greeks *out);
int mc_vanilla_greeks_run_f64(mc_context *ctx, const mc_option_f64 *opt, uint64_t seed,
                              uint64_t first_path, uint64_t n_paths, mc_vanilla_greeks *out);

/* Likelihood-ratio forms of the same three numbers: delta = payoff z / (S sigma sqrt T), vega = payoff ((z^2 - 1) / sigma
 * - z sqrt T), z the path's normal.  They differentiate the density instead of the payoff (no indicator), at the price
 * of a larger variance on a smooth payoff like this one; needs v > 0 and t > 0. */
int mc_vanilla_greeks_lr_run_f32(mc_context *ctx, const mc_option_f32 *opt, uint64_t seed,
                                 uint64_t first_path, uint64_t n_paths, mc_vanilla_greeks *out);
int mc_vanilla_greeks_lr_run_f64(mc_context *ctx, const mc_option_f64 *opt, uint64_t seed,
                                 uint64_t first_path, uint64_t n_paths, mc_vanilla_greeks *out);

/* ---- pathwise Greeks of the basket call ------------------------------------------------
 * On the pricing kernels' stream and path indexing (reference formulas dp/MonteCarloKernel.cu:74-101), with
 * B = sum_a w_a S_a(T), I = [B > K]:  price,  delta[a] = dV/dS_a = I w_a S_a(T) / S_a,
 * vega[a] = dV/dv_a = I w_a S_a(T) (bt_a sqrt T - v_a T); discounted means with their own 95 % half-widths.
 * delta and vega are caller arrays of opt->n results.  Plain estimator; one pass over the paths per 8 assets (a lane keeps the
 * (sum, sum2) pairs of the price and of eight assets' two derivatives in registers). */
int mc_basket_greeks_run_f32(mc_context *ctx, const mc_basket_f32 *opt, uint64_t seed, uint64_t first_path,
                             uint64_t n_paths, mc_result *price, mc_result *delta, mc_result *vega);
int mc_basket_greeks_run_f64(mc_context *ctx, const mc_basket_f64 *opt, uint64_t seed, uint64_t first_path,
                             uint64_t n_paths, mc_result *price, mc_result *delta, mc_result *vega);
/* Likelihood-ratio forms of the same 1 + 2n numbers: the payoff times the score of the terminal prices' joint lognormal
 * density.  With y = L^-T g (the path's normals whitened by the inverse factor, formed on the host):
 *   delta[a] = payoff y_a / (S_a v_a sqrt T),
 *   vega[a]  = payoff [ (y_a (L g)_a - 1) / v_a + (sqrt T d_a - v_a T) y_a / (v_a sqrt T) ]
 * (one asset: the vanilla forms above).  Needs t > 0, every v[a] > 0 and a non-singular factor (every p[a][a] > 0 -- the
 * reference driver's own 3 x 3 correlation is singular, SURVEY 2.3 #10, and is refused). */
int mc_basket_greeks_lr_run_f32(mc_context *ctx, const mc_basket_f32 *opt, uint64_t seed, uint64_t first_path,
                                uint64_t n_paths, mc_result *price, mc_result *delta, mc_result *vega);
int mc_basket_greeks_lr_run_f64(mc_context *ctx, const mc_basket_f64 *opt, uint64_t seed, uint64_t first_path,
                                uint64_t n_paths, mc_result *price, mc_result *delta, mc_result *vega);

/* ---- CVA with its pathwise delta and vega -------------------------------------------------
 * On the CVA kernel's stream (reference loop dp/MonteCarloKernel.cu:241-262), not discounted, like the CVA itself (:466):
 *   d CVA / d S_0   = LGD sum_j dp_j cnd(d1_j) S_j / S_0
 *   d CVA / d sigma = LGD sum_j dp_j [ S_j phi(d1_j) sqrt(tau_j) + S_j cnd(d1_j) (W_j - sigma t_j) ],  W_j the Brownian
 *                     motion at date j (closed-form vega of the exposure + the path's own sensitivity).  Plain estimator. */
typedef struct { mc_result cva, delta, vega; } mc_cva_greeks;
int mc_cva_greeks_run_f32(mc_context *ctx, const mc_cva_f32 *cva, uint64_t seed, uint64_t first_path,
                          uint64_t n_paths, mc_cva_greeks *out);
int mc_cva_greeks_run_f64(mc_context *ctx, const mc_cva_f64 *cva, uint64_t seed, uint64_t first_path,
                          uint64_t n_paths, mc_cva_greeks *out);
/* Likelihood-ratio forms: only the first transition's density depends on S_0, every transition's on sigma, and the closed-form
 * exposure depends on sigma explicitly (that part stays pathwise):
 *   d CVA / d S_0   = E[ CVA_path z_1 ] / (S_0 sigma sqrt(dt))
 *   d CVA / d sigma = E[ LGD sum_j dp_j S_j phi(d1_j) sqrt(tau_j) + CVA_path sum_j ((z_j^2 - 1) / sigma - z_j sqrt(dt)) ]
 * Same stream, same CVA plane; the variance grows with the number of dates (the scores add up). */
int mc_cva_greeks_lr_run_f32(mc_context *ctx, const mc_cva_f32 *cva, uint64_t seed, uint64_t first_path,
                             uint64_t n_paths, mc_cva_greeks *out);
int mc_cva_greeks_lr_run_f64(mc_context *ctx, const mc_cva_f64 *cva, uint64_t seed, uint64_t first_path,
                             uint64_t n_paths, mc_cva_greeks *out);

/* ---- per-path values, for parity tests ---------------------------------------------------
 * h_out: HOST pointer to n_paths values (undiscounted payoffs / per-path CVA).  Same kernels
 * as above with a store of every value added; n_paths <= 2^26. */
int mc_vanilla_paths_f32(mc_context *ctx, const mc_option_f32 *opt, uint64_t seed,
                         uint64_t first_path, uint64_t n_paths, float *h_out);
int mc_vanilla_paths_f64(mc_context *ctx, const mc_option_f64 *opt, uint64_t seed,
                         uint64_t first_path, uint64_t n_paths, double *h_out);
int mc_basket_paths_f32(mc_context *ctx, const mc_basket_f32 *opt, uint64_t seed,
                        uint64_t first_path, uint64_t n_paths, float *h_out);
int mc_basket_paths_f64(mc_context *ctx, const mc_basket_f64 *opt, uint64_t seed,
                        uint64_t first_path, uint64_t n_paths, double *h_out);
int mc_cva_paths_f32(mc_context *ctx, const mc_cva_f32 *cva, uint64_t seed,
                     uint64_t first_path, uint64_t n_paths, float *h_out);
int mc_cva_paths_f64(mc_context *ctx, const mc_cva_f64 *cva, uint64_t seed,
                     uint64_t first_path, uint64_t n_paths, double *h_out);
/* ---- compatibility mode: the reference's launch geometry and per-thread XORWOW streams ---------------------------
 * The reference's result depends on (numBlocks, numThreads): dp/MonteCarloKernel.cu:285-290 gives every thread of the
 * launch its own cuRAND XORWOW state, curand_init(seed = blockIdx.x + gridDim.x, subsequence = threadIdx.x, offset 0);
 * thread t of a block prices paths t, t + numThreads, ... < N_PATH of that block (:146,191,240) and draws its normals
 * with curand_normal() one after the other (two words per Box-Muller pair, the second member kept for the next call;
 * :68,78,250 -- in the dp build the float normal is widened to double).  These calls reproduce that sample: same seeding
 * geometry, same thread-to-path assignment, same per-thread order of the draws -- a CVA path draws for the dates whose
 * `t -= dt` is still >= 0 (:249) -- with the generator and Box-Muller of rocRAND/hipRAND (rocrand_xorwow.h,
 * rocrand_normal.h), i.e. what a HIP build of the reference draws on this GPU, bit for bit (tests/test_gpu_grid.py).
 * cuRAND itself seeds XORWOW with other constants and is not in this image: equality with an NVIDIA run of the reference
 * is NOT claimed ("parity unpinned").
 *   n = num_blocks * paths_per_block paths are priced (the reference's numBlocks * (sims / numBlocks)).
 * How it runs (round 4): the call IS the reference's launch -- num_blocks workgroups of num_threads threads, every thread
 * with its XORWOW state and Box-Muller pair in registers, pricing its own paths through the per-path code of the hot kernels;
 * no normal touches HBM (1e8 vanilla paths at 512 x 128: 0.40 ms in round 3's staged form, see DESIGN.md for this one).
 * Shapes without a fused kernel (baskets beyond 16 assets, more blocks than the context's pair buffer holds) take round
 * 3's staged form: the normals of the call are written to HBM (one Real per draw) and priced by the simulation kernels
 * of mc_*_run_* through the external-normals policy -- same sample, same per-path values.  MC_GRID_FORM=staged / fused in
 * the environment forces a form.  The start states of a geometry are set up on first use and kept (the last 4 geometries).
 * Plain estimator only; the context's generator / normals settings are ignored.  num_threads <= 1024, at most 2^24 threads
 * and 2^31 paths.  mc_result.kernel_ms covers the pricing kernel only; wall_ms the whole call. */
int mc_vanilla_run_grid_f32(mc_context *ctx, const mc_option_f32 *opt, int num_blocks, int num_threads,
                            uint64_t paths_per_block, mc_result *out);
int mc_vanilla_run_grid_f64(mc_context *ctx, const mc_option_f64 *opt, int num_blocks, int num_threads,
                            uint64_t paths_per_block, mc_result *out);
int mc_basket_run_grid_f32(mc_context *ctx, const mc_basket_f32 *opt, int num_blocks, int num_threads,
                           uint64_t paths_per_block, mc_result *out);
int mc_basket_run_grid_f64(mc_context *ctx, const mc_basket_f64 *opt, int num_blocks, int num_threads,
                           uint64_t paths_per_block, mc_result *out);
int mc_cva_run_grid_f32(mc_context *ctx, const mc_cva_f32 *cva, int num_blocks, int num_threads,
                        uint64_t paths_per_block, mc_result *out);
int mc_cva_run_grid_f64(mc_context *ctx, const mc_cva_f64 *cva, int num_blocks, int num_threads,
                        uint64_t paths_per_block, mc_result *out);
/* ---- host-side helpers -------------------------------------------------------------- */
/* Closing formulas of dp/MonteCarloKernel.cu:420-423 (discount = exp(-rT)) and :466-468
 * (discount = 1), in fp64, from an (all-reduced) triple. */
void mc_closing(double sum, double sum2, uint64_t n, double discount, double *expected,
                double *confidence);
/* Contiguous shard of `total` paths owned by `rank` of `world`: [*first, *first + *count).
 * rank g owns [floor(gP/G), floor((g+1)P/G)). */
void mc_shard_range(uint64_t total, int rank, int world, uint64_t *first, uint64_t *count);
/* Cholesky with the reference's semantics (dp/MonteCarloHost.c:90-105: column-oriented,
 * a non-positive pivot leaves its column zero).  Row-major n x n; returns the number of
 * non-positive pivots met (0 = the input was positive definite). */
int mc_chol_f32(int n, const float *c, float *a);
int mc_chol_f64(int n, const double *c, double *a);
/* Covariance-matrix input (the reference's drivers take volatilities and a correlation matrix and
 * call Chol themselves, dp/basketOpt.cu:34-61,96-99).  cov = row-major n x n covariance of the assets' annualised
 * log-returns; like Chol (dp/MonteCarloHost.c:96-99) only its lower triangle is read.  Out: v[n] = sqrt(cov[a][a])
 * and p[n*n] = Cholesky factor (reference semantics) of the correlation matrix cov[a][b] / (v[a] v[b]) -- the `v`
 * and `p` of mc_basket_* / MultiOptionData.  Returns the number of non-positive pivots (0 = positive definite), or
 * -1 for a bad argument (n < 1, NULL, a non-finite entry, a diagonal entry <= 0). */
int mc_factor_from_cov_f32(int n, const float *cov, float *v, float *p);
int mc_factor_from_cov_f64(int n, const double *cov, double *v, double *p);

#ifdef __cplusplus
}
#endif
#endif /* MC_MI355X_H_ */
