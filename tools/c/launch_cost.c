/* launch_cost.c -- host time of one asynchronous mc_*_launch_* call (what a single thread pays per device when it
 * fans one pricing call out over G GPUs: libmc_multi.so).  G contexts on device 0, each with its own stream; per
 * round the G launches are timed together, then all streams are drained.
 *   gcc -O2 -Iinclude tools/c/launch_cost.c -Lmontecarlocuda_amd/csrc -lmc_mi355x -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -o /tmp/launch_cost */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mc_mi355x.h"

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }

int main(void)
{
    enum { G = 8, ROUNDS = 400 };
    mc_context *ctx[G];
    double *triple[G];
    extern int hipMalloc(void **, size_t);
    for (int g = 0; g < G; ++g) {
        if (mc_context_create(0, 0, &ctx[g]) != MC_OK) { fprintf(stderr, "%s\n", mc_last_error()); return 1; }
        if (hipMalloc((void **)&triple[g], 24) != 0) return 1;
    }
    const mc_option_f32 van = {100.f, 100.f, 0.048790f, 0.2f, 1.f};
    const mc_cva_f64 cva = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};
    double t_van[ROUNDS], t_cva[ROUNDS];
    for (int r = 0; r < ROUNDS; ++r) {
        double t0 = now_us();
        for (int g = 0; g < G; ++g)
            mc_vanilla_launch_f32(ctx[g], &van, MC_DEFAULT_SEED, (uint64_t)g << 20, 1 << 20, triple[g], mc_context_stream(ctx[g]));
        t_van[r] = (now_us() - t0) / G;
        for (int g = 0; g < G; ++g)
            while (mc_context_idle(ctx[g]) == 0) {}
        t0 = now_us();
        for (int g = 0; g < G; ++g)
            mc_cva_launch_f64(ctx[g], &cva, MC_DEFAULT_SEED, (uint64_t)g << 16, 1 << 16, triple[g], mc_context_stream(ctx[g]));
        t_cva[r] = (now_us() - t0) / G;
        for (int g = 0; g < G; ++g)
            while (mc_context_idle(ctx[g]) == 0) {}
    }
    qsort(t_van, ROUNDS, sizeof(double), cmp);
    qsort(t_cva, ROUNDS, sizeof(double), cmp);
    printf("host time per asynchronous launch, %d contexts on one device, median of %d rounds (min):\n", G, ROUNDS);
    printf("  mc_vanilla_launch_f32: %.2f us (%.2f)\n  mc_cva_launch_f64 (256 dates, table cached): %.2f us (%.2f)\n", t_van[ROUNDS / 2], t_van[0],
           t_cva[ROUNDS / 2], t_cva[0]);
    return 0;
}
