/*
 * mc_multi.h -- C ABI of libmc_multi.so: one pricing call sharded over several MI355X of one node, driven by
 * ONE host process, closed by ONE RCCL all-reduce of the 24-byte {sum, sum2, n} triple over xGMI.
 *
 * New relative to the reference, which is strictly single-device (no cudaSetDevice, no NCCL/MPI anywhere:
 * SURVEY 2.2); it fans out the reference's three GPU entry points
 *     dev_basketOpt / dev_vanillaOpt / dev_cvaEquityOption      dp/MonteCarloKernel.cu:483,500,517
 * the way BASELINE.json's north_star asks: host code stays in C, "paths shard embarrassingly across the 8 GPUs
 * of one node with a final RCCL all-reduce of partial (sum, sum^2) over xGMI".
 *
 * How a call runs (mc_multi.cpp):
 *   device g of G owns the contiguous global paths [first + floor(g n / G), first + floor((g+1) n / G))
 *   (mc_shard_range), same seed: a path's normals depend only on (seed, global path index), so the union of the
 *   shards IS the single-GPU sample; every device's launch (mc_*_launch_*, include/mc_mi355x.h) is enqueued on that
 *   device's own stream -- with more than one device by that device's own LAUNCHER THREAD, which the handle creates and
 *   the calling thread starts through one flag word, so that all devices start together (serially from the calling
 *   thread an asynchronous launch costs ~4 us: device 7 of 8 would start ~30 us late) -- and leaves its triple in that
 *   device's HBM; then ONE grouped
 *   ncclAllReduce(count = 3, ncclDouble, ncclSum) on the same streams (communicators from ncclCommInitAll, created
 *   once with the handle, never per call), 24 bytes read back from the first device, closing formulas on the host.
 *   The pre-reduction triples are read back as well and added on the host in device order: the cross-check of the
 *   collective (they must agree to 1e-12 relative, else the call fails) and, with MC_MULTI_REDUCE=host or
 *   mc_multi_set_reduce(m, MC_REDUCE_HOST), the result itself -- no RCCL needed then (also the only way to list
 *   one device twice, which RCCL refuses: used by the tests to exercise G > 1 on a one-GPU box).
 *
 * The legacy symbols (libmcgpu_f32/_f64.so) take this path when the environment variable MC_DEVICES is set
 * ("0,1,2,3" or "all"); without it they keep using one device (MC_DEVICE) and never load RCCL.
 *
 * Results: sums differ from the single-device call only in the order of fp64 additions (per-workgroup pairs are
 * added per device first): <= 1e-12 relative in fp64, <= 2e-9 in fp32 (the fp32 kernels add up to 16 values in
 * float before each flush to double, and which 16 depends on the shard boundaries).  Errors: status codes as in
 * mc_mi355x.h, text from mc_multi_last_error().
 */
#ifndef MC_MULTI_H_
#define MC_MULTI_H_

#include "mc_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_multi mc_multi;

enum { MC_REDUCE_RCCL = 0, MC_REDUCE_HOST = 1 };

/* devices: HIP device ordinals, n_devices of them; devices == NULL means 0 .. n_devices-1, and n_devices <= 0
 * with devices == NULL means every visible device.  blocks as in mc_context_create (0 = default grid).
 * One mc_context per listed device (replaces dp/MonteCarloKernel.cu:296 MonteCarlo_init, once instead of per call). */
int mc_multi_create(const int *devices, int n_devices, int blocks, mc_multi **out);
void mc_multi_destroy(mc_multi *m);
int mc_multi_size(const mc_multi *m);
/* the i-th device's context (e.g. for mc_context_info); owned by the handle */
mc_context *mc_multi_context(mc_multi *m, int i);
/* estimator switches, applied to every device (mc_context_set_antithetic / _control_variate) */
int mc_multi_set_antithetic(mc_multi *m, int on);
int mc_multi_set_control_variate(mc_multi *m, int on);
/* normals of the fp64 kernels on every device (mc_context_set_normals: MC_NORMALS_NATIVE or MC_NORMALS_F32) */
int mc_multi_set_normals(mc_multi *m, int mode);
/* generator of every device (mc_context_set_generator); under MC_RNG_XORWOW device g's lanes run the subsequences
 * subsequence_base + (lanes of devices 0..g-1) + lane, so that no two lanes of the job share a sequence */
int mc_multi_set_generator(mc_multi *m, int generator, uint64_t subsequence_base);
/* on != 0 (default): two HIP events per device and call, kernel_ms reported; 0: none, kernel_ms = 0 (the launching thread
 * reaches the next device sooner; what the legacy symbols use unless MC_VERBOSE is set) */
int mc_multi_set_timing(mc_multi *m, int on);
/* MC_REDUCE_RCCL (default; MC_MULTI_REDUCE=host in the environment selects the other) or MC_REDUCE_HOST */
int mc_multi_set_reduce(mc_multi *m, int mode);
/* Launcher threads: one per device, created with the handle when it has more than one device AND the process can keep
 * devices + 1 threads running (affinity mask capped by the cgroup CPU quota: cpu.max / cfs quota); MC_MULTI_THREADS=0 in the
 * environment keeps the serial fan-out from the calling thread, =1 forces the threads.  A launcher thread spins on the crew's
 * call-number word for MC_MULTI_LINGER_US (default 5000) after its last job and then sleeps: calls shorter than that which
 * follow each other pay no wake-up, an idle handle uses no core, and a handle called more rarely than every 5 ms is served by
 * the calling thread -- a sleeping thread's job is run by the caller AT ONCE (no wake-up on the critical path; the sleepers are
 * woken after the fan-out, and only when calls shorter than the linger time come within the linger time of each other), the job of a spinning thread that
 * has not taken it 15 us after the hand-off (its core was taken away) likewise.  What no take-over can bound: a thread that loses
 * its core INSIDE the job it has claimed (or a runtime call that blocks) -- counted as `slow_claimed` (5 of 2.4 million jobs took
 * more than 1 ms that way, the worst 7.66 ms, p99.9 of the whole fan-out 58 us: profiles/r05_multi_soak_300k_calls.log).  The
 * WAIT for such a job is bounded all the same: when a launcher thread has not returned from a launch it claimed 20 s after the
 * hand-off, mc_multi_*_run_* fails with MC_ERR_HIP naming the device instead of spinning for ever.  The handle is unusable
 * from then on (every later call fails at once) and mc_multi_destroy leaks it on purpose -- the stuck thread may still come back
 * and touch it; the option struct passed to the call that timed out must stay valid for as long as the process lives.
 * What the threads buy, measured with timing OFF (mc_multi_set_timing(m, 0): pinned-slot read-back, what the legacy symbols
 * use): eight launches enqueued in 8.8 us (median) instead of 22.2 serial; the tail is in profiles/r05_multi_soak_*.log
 * (p50 / p99 / p99.9 / max of every call's fan-out, and who was late in the slowest twenty).  With timing on, every call ends
 * in a copy and a synchronize per device, and launches that follow a synchronize serialise inside the HIP runtime whatever
 * thread issues them (profiles/r04_multi_fixed_cost_and_fanout_2.log).
 * mc_multi_launcher_threads: how many the handle runs (0 = serial).  mc_multi_last_fanout_us: host time from the entry of
 * the last mc_multi_*_run_* call until the LAST device's launch had been enqueued.  mc_multi_last_fanout_trace: the same call
 * per device (returns the device count; fills at most `cap` entries): seen_us[g] = when device g's thread saw the call (-1:
 * the caller ran the job because the thread was late, -2: because it was asleep, -3: serial fan-out), enqueued_us[g] = when its
 * launch had been enqueued.  mc_multi_fanout_stats: what became of every job so far.  mc_multi_describe: the handle's resolved
 * configuration as one line (printed to stderr at creation under MC_VERBOSE=2).  One calling thread per handle. */
int mc_multi_launcher_threads(const mc_multi *m);
double mc_multi_last_fanout_us(const mc_multi *m);
/* The collective as the calling thread saw it, last call, pinned-slot read-back (timing off) with MC_REDUCE_RCCL: microseconds from
 * the moment the LAST device's own triple was visible on the host to the moment the all-reduced triple was -- the grouped
 * ncclAllReduce plus the one-lane publish kernel behind it (-1: not measured: host reduction, copy read-back).
 * mc_multi_last_device_us: per device, microseconds from call entry until its own triple was visible (returns the device count). */
double mc_multi_last_collective_us(const mc_multi *m);
int mc_multi_last_device_us(const mc_multi *m, int cap, double *delivered_us);
int mc_multi_last_fanout_trace(const mc_multi *m, int cap, double *seen_us, double *enqueued_us);
typedef struct {
    uint64_t calls, by_worker, served_parked, stolen, slow_claimed, wakeups;
} mc_multi_fanout_counts;
int mc_multi_fanout_stats(const mc_multi *m, mc_multi_fanout_counts *out);
int mc_multi_describe(const mc_multi *m, char *buf, int len);
/* The calling thread's current HIP device is unchanged by mc_multi_create / _destroy / _*_run_* (restored on the way out). */
/* Text of the last failure of an mc_multi_* call on this thread ("" if none). */
const char *mc_multi_last_error(void);
/* |RCCL sum - host sum| / |host sum| of the last call's `sum` (0 when the host did the reduction) */
double mc_multi_last_reduce_error(const mc_multi *m);

/* Synchronous sharded runs.  out->kernel_ms = the slowest device's simulation time (HIP events on its stream),
 * out->wall_ms = host wall-clock from the first launch to the closed estimate ("wall-clock from first
 * launch to the all-reduced result").  Same arguments as mc_*_run_* (include/mc_mi355x.h). */
int mc_multi_vanilla_run_f32(mc_multi *m, const mc_option_f32 *opt, uint64_t seed, uint64_t first_path,
                             uint64_t n_paths, mc_result *out);
int mc_multi_vanilla_run_f64(mc_multi *m, const mc_option_f64 *opt, uint64_t seed, uint64_t first_path,
                             uint64_t n_paths, mc_result *out);
int mc_multi_basket_run_f32(mc_multi *m, const mc_basket_f32 *opt, uint64_t seed, uint64_t first_path,
                            uint64_t n_paths, mc_result *out);
int mc_multi_basket_run_f64(mc_multi *m, const mc_basket_f64 *opt, uint64_t seed, uint64_t first_path,
                            uint64_t n_paths, mc_result *out);
int mc_multi_cva_run_f32(mc_multi *m, const mc_cva_f32 *cva, uint64_t seed, uint64_t first_path,
                         uint64_t n_paths, mc_result *out);
int mc_multi_cva_run_f64(mc_multi *m, const mc_cva_f64 *cva, uint64_t seed, uint64_t first_path,
                         uint64_t n_paths, mc_result *out);

#ifdef __cplusplus
}
#endif
#endif /* MC_MULTI_H_ */
