/* multi_soak.c -- soak of libmc_multi's launcher threads (csrc/mc_multi_host.hpp: one call-number word for the crew, a job
 * belongs to whoever claims it, workers park after the linger time): EIGHT contexts on device 0, host reduction, a long run
 * of short calls of cycling sizes and products with pauses of 0 ... 4 ms between them (shorter and longer than the linger
 * time given in MC_MULTI_LINGER_US, so that workers are met spinning, parking and asleep), every result compared BIT FOR
 * BIT with the same call through a handle with the serial fan-out (MC_MULTI_THREADS=0).  A lost or doubled hand-off shows as
 * a mismatch, a wrong n, or a hang (run under a timeout).
 *   gcc -O2 -std=gnu11 -Iinclude tools/c/multi_soak.c -Lmontecarlocuda_amd/csrc -lmc_multi -lmc_mi355x -lm \
 *       -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -Wl,-rpath-link,montecarlocuda_amd/csrc:/opt/rocm/lib -o /tmp/multi_soak
 *   MC_MULTI_LINGER_US=1000 /tmp/multi_soak [calls] */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mc_multi.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv)
{
    const long calls = argc > 1 ? atol(argv[1]) : 200000;
    const int dev8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mc_multi *threaded, *serial;
    if (mc_multi_create(dev8, 8, 0, &threaded) != MC_OK) { fprintf(stderr, "%s\n", mc_multi_last_error()); return 2; }
    setenv("MC_MULTI_THREADS", "0", 1);
    if (mc_multi_create(dev8, 8, 0, &serial) != MC_OK) { fprintf(stderr, "%s\n", mc_multi_last_error()); return 2; }
    unsetenv("MC_MULTI_THREADS");
    mc_multi *h[2] = {threaded, serial};
    for (int i = 0; i < 2; ++i) {
        mc_multi_set_reduce(h[i], MC_REDUCE_HOST);
        mc_multi_set_timing(h[i], 0);
    }
    printf("launcher threads: %d and %d; %ld calls\n", mc_multi_launcher_threads(threaded), mc_multi_launcher_threads(serial), calls);
    const mc_option_f32 v32 = {100.f, 100.f, 0.048790f, 0.2f, 1.f};
    const mc_option_f64 v64 = {100., 100., 0.048790, 0.2, 1.};
    const mc_cva_f64 c64 = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 16};
    long bad = 0;
    unsigned lcg = 12345u;
    const double t0 = now_s();
    double fan_sum = 0, fan_max = 0;
    for (long i = 0; i < calls; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        const uint64_t n = 5 + (lcg >> 8) % 40000, first = (uint64_t)i * 1000003ull;   /* also fewer paths than devices */
        mc_result a, b;
        int ra, rb;
        switch (i % 3) {
        case 0: ra = mc_multi_vanilla_run_f32(threaded, &v32, MC_DEFAULT_SEED, first, n, &a), rb = mc_multi_vanilla_run_f32(serial, &v32, MC_DEFAULT_SEED, first, n, &b); break;
        case 1: ra = mc_multi_vanilla_run_f64(threaded, &v64, MC_DEFAULT_SEED, first, n, &a), rb = mc_multi_vanilla_run_f64(serial, &v64, MC_DEFAULT_SEED, first, n, &b); break;
        default: ra = mc_multi_cva_run_f64(threaded, &c64, MC_DEFAULT_SEED, first, 1 + n / 16, &a), rb = mc_multi_cva_run_f64(serial, &c64, MC_DEFAULT_SEED, first, 1 + n / 16, &b); break;
        }
        if (ra != MC_OK || rb != MC_OK) { fprintf(stderr, "call %ld failed: %s\n", i, mc_multi_last_error()); return 2; }
        const double f = mc_multi_last_fanout_us(threaded);
        fan_sum += f, fan_max = f > fan_max ? f : fan_max;
        if (!(a.sum == b.sum && a.sum2 == b.sum2 && a.n == b.n)) {
            if (++bad <= 5) printf("MISMATCH at call %ld: n %llu / %llu sum %.17g / %.17g\n", i, (unsigned long long)a.n, (unsigned long long)b.n, a.sum, b.sum);
        }
        if ((lcg >> 4) % 97 == 0) {   /* a pause of 0.05 ... 4 ms now and then */
            struct timespec ts = {0, 50000 + (long)((lcg >> 12) % 80) * 50000};
            nanosleep(&ts, NULL);
        }
        if (i % 50000 == 49999) printf("%ld calls, %ld mismatches, %.1f s\n", i + 1, bad, now_s() - t0), fflush(stdout);
    }
    printf("%ld calls through 8 launcher threads against the serial fan-out: %ld mismatches, %.1f s; fan-out mean %.2f us, worst %.1f us\n", calls, bad,
           now_s() - t0, fan_sum / calls, fan_max);
    mc_multi_destroy(threaded);
    mc_multi_destroy(serial);
    return bad ? 1 : 0;
}
