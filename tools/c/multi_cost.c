/* multi_cost.c -- where the fixed cost of one libmc_multi call goes (C5 shard of 8: CVA 256 dates, 1.25e6 paths, fp64).
 * On ONE device, medians of REPS calls each:
 *   single      mc_cva_run_f64, timing off: the single-device floor (last workgroup writes pinned memory, host polls)
 *   single+ev   the same with timing on: HIP events + 24-byte copy + synchronize; kernel_ms = the events' figure
 *   multi host  libmc_multi, host reduction, pinned slots (no RCCL call, no publish kernel)
 *   multi rccl  libmc_multi, RCCL all-reduce over a communicator of ONE + publish kernel, pinned slots
 *   ... copy    the two multi forms with round 2's read-back (MC_MULTI_READBACK=copy is read at handle creation)
 *   fan-out     EIGHT contexts on device 0 (the one-GPU stand-in for 8 devices; host reduction, RCCL refuses a repeated device),
 *               1.25e6 paths sharded over them: host time from call entry until the LAST device's launch had been enqueued
 *               (mc_multi_last_fanout_us), serial from the calling thread (MC_MULTI_THREADS=0: rounds 2-3) against one
 *               launcher thread per device (round 4) -- VERDICT r03 "next" #1c asks for <= 6 us
 * Build on the GPU box:
 *   gcc -O2 -std=gnu11 -Iinclude tools/c/multi_cost.c -Lmontecarlocuda_amd/csrc -lmc_multi -lmc_mi355x -lm \
 *       -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -Wl,-rpath-link,montecarlocuda_amd/csrc:/opt/rocm/lib -o /tmp/multi_cost */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mc_multi.h"

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }
enum { REPS = 41 };
static double median(double *t) { qsort(t, REPS, sizeof *t, cmp); return t[REPS / 2]; }

/* one timed call: wall-clock of `call`, and the simulation kernel's own duration from the HIP events bound to its
 * dispatch (mc_context_profile: independent of the read-back form) */
#define TIMED(ctx, call, wall_us, kernel_us)                                                       \
    do {                                                                                           \
        const double t0_ = now_us();                                                               \
        if ((call) != MC_OK) { fprintf(stderr, "%s | %s\n", mc_multi_last_error(), mc_last_error()); return 1; } \
        (wall_us) = now_us() - t0_;                                                                \
        int ns_ = 0; double ms_ = 0;                                                               \
        mc_context_profile_read(ctx, &ns_, &ms_);                                                  \
        (kernel_us) = ns_ ? ms_ * 1e3 / ns_ : 0;                                                   \
    } while (0)

static void report(const char *name, double *wall, double *kern)
{
    double d[REPS];
    for (int i = 0; i < REPS; ++i) d[i] = wall[i] - kern[i];
    const double w = median(wall), k = median(kern), dd = median(d);
    printf("%-36s wall %8.1f us   kernel %8.1f us   wall - kernel (per call, median) %6.1f us  (min %5.1f, max %6.1f)\n", name, w, k, dd, d[0], d[REPS - 1]);
}

int main(int argc, char **argv)
{
    const uint64_t paths = argc > 1 ? strtoull(argv[1], NULL, 0) : 1250000ull;
    const mc_cva_f64 cva = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};
    mc_result r;
    double wall[REPS], kern[REPS], w_, k_;
    mc_context *c;
    if (mc_context_create(0, 0, &c) != MC_OK) { fprintf(stderr, "%s\n", mc_last_error()); return 1; }
    mc_context_profile(c, 1);
    for (int timing = 1; timing >= 0; --timing) {
        mc_context_set_timing(c, timing);
        for (int i = -3; i < REPS; ++i) {
            TIMED(c, mc_cva_run_f64(c, &cva, MC_DEFAULT_SEED, 0, paths, &r), w_, k_);
            if (i >= 0) wall[i] = w_, kern[i] = k_;
        }
        report(timing ? "single, events + copy + sync" : "single, pinned slot (the floor)", wall, kern);
    }
    mc_context_destroy(c);
    for (int copy = 0; copy < 2; ++copy) {
        if (copy) setenv("MC_MULTI_READBACK", "copy", 1); else unsetenv("MC_MULTI_READBACK");
        for (int mode = 1; mode >= 0; --mode) {   /* 1 = host reduction, 0 = RCCL */
            mc_multi *m;
            if (mc_multi_create(NULL, 1, 0, &m) != MC_OK) { fprintf(stderr, "%s\n", mc_multi_last_error()); return 1; }
            mc_multi_set_reduce(m, mode ? MC_REDUCE_HOST : MC_REDUCE_RCCL);
            mc_multi_set_timing(m, 0);
            mc_context *mc0 = mc_multi_context(m, 0);
            mc_context_profile(mc0, 1);
            for (int i = -3; i < REPS; ++i) {
                TIMED(mc0, mc_multi_cva_run_f64(m, &cva, MC_DEFAULT_SEED, 0, paths, &r), w_, k_);
                if (i >= 0) wall[i] = w_, kern[i] = k_;
            }
            char name[64];
            snprintf(name, sizeof name, "multi %s, %s", mode ? "host sum" : "rccl(1 rank)", copy ? "copies + sync" : "pinned slots");
            report(name, wall, kern);
            mc_multi_destroy(m);
        }
    }
    /* fan-out over 8 "devices", pinned-slot read-back (the product's default; rounds up to r04's first log measured this
     * section with MC_MULTI_READBACK=copy still set by the loop above: every call then ends in 8 copies + 8 synchronizes,
     * and the launches that follow a synchronize cost 2-4 x as much host time each) and, for comparison, the copy form */
    for (int pass = 0; pass < 4; ++pass) {
        const int threads = pass & 1, copy = pass >> 1;
        if (copy) setenv("MC_MULTI_READBACK", "copy", 1); else unsetenv("MC_MULTI_READBACK");
        if (threads) unsetenv("MC_MULTI_THREADS"); else setenv("MC_MULTI_THREADS", "0", 1);
        const int dev8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        mc_multi *m;
        if (mc_multi_create(dev8, 8, 0, &m) != MC_OK) { fprintf(stderr, "%s\n", mc_multi_last_error()); return 1; }
        mc_multi_set_reduce(m, MC_REDUCE_HOST);
        mc_multi_set_timing(m, 0);
        double fan[REPS];
        for (int i = -3; i < REPS; ++i) {
            const double t0 = now_us();
            if (mc_multi_cva_run_f64(m, &cva, MC_DEFAULT_SEED, 0, paths, &r) != MC_OK) { fprintf(stderr, "%s | %s\n", mc_multi_last_error(), mc_last_error()); return 1; }
            if (i >= 0) wall[i] = now_us() - t0, fan[i] = mc_multi_last_fanout_us(m);
        }
        qsort(fan, REPS, sizeof *fan, cmp);
        printf("fan-out over 8 contexts, %-13s %-28s call entry -> last launch enqueued: median %6.2f us  (min %5.2f, max %6.2f)   call wall median %8.1f us   [%d launcher threads]\n",
               copy ? "copies + sync," : "pinned slots,", threads ? "one launcher thread each:" : "serial (MC_MULTI_THREADS=0):", fan[REPS / 2], fan[0], fan[REPS - 1], median(wall), mc_multi_launcher_threads(m));
        mc_multi_destroy(m);
    }
    return 0;
}
