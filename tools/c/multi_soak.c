/* multi_soak.c -- soak of libmc_multi's launcher threads (csrc/mc_multi_host.hpp: one call-number word for the crew, a job
 * belongs to whoever claims it, workers park after the linger time): EIGHT contexts on device 0, host reduction, a long run
 * of short calls of cycling sizes and products with pauses of 0 ... 4 ms between them (shorter and longer than the linger
 * time given in MC_MULTI_LINGER_US, so that workers are met spinning, parking and asleep), every result compared BIT FOR
 * BIT with the same call through a handle with the serial fan-out (MC_MULTI_THREADS=0).  A lost or doubled hand-off shows as
 * a mismatch, a wrong n, or a hang (run under a timeout).
 *   gcc -O2 -std=gnu11 -Iinclude tools/c/multi_soak.c -Lmontecarlocuda_amd/csrc -lmc_multi -lmc_mi355x -lm \
 *       -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -Wl,-rpath-link,montecarlocuda_amd/csrc:/opt/rocm/lib -o /tmp/multi_soak
 * Every call's fan-out (call entry -> last device's launch enqueued) is kept: the run ends with p50 / p99 / p99.9 / max for the
 * threaded AND the serial handle, what became of every job (mc_multi_fanout_stats), and for the twenty slowest threaded calls each
 * device's "saw the call -> launch enqueued" pair (mc_multi_last_fanout_trace) -- which says whether the late party was a sleeping
 * thread, a thread that lost its core before taking the job (the caller takes over), or one that lost it INSIDE the job.
 *   MC_MULTI_LINGER_US=1000 /tmp/multi_soak [calls] */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mc_multi.h"

static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : x > y;
}

typedef struct {
    double fan;
    long call;
    double gap_ms;      /* host time since the previous call returned */
    double seen[8], at[8];
} Slow;
enum { N_SLOW = 20 };

static void percentiles(const char *name, float *v, long n)
{
    qsort(v, (size_t)n, sizeof v[0], cmp_float);
    double sum = 0;
    for (long i = 0; i < n; ++i) sum += v[i];
    printf("fan-out of %ld calls, %s: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  p99.9 %.2f  p99.99 %.2f  max %.1f us\n", n, name, sum / n, v[n / 2],
           v[(long)(n * 0.9)], v[(long)(n * 0.99)], v[(long)(n * 0.999)], v[(long)(n * 0.9999)], v[n - 1]);
}

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
    /* the first launch of each kernel on a handle loads its code object (10 ms for the first, ~1 ms for the others: round 4's
     * "worst 10 851 us" was exactly this call): made here, reported apart, not part of the percentiles */
    {
        mc_result r;
        for (int k = 0; k < 3; ++k) {
            for (int i = 0; i < 2; ++i) {
                const int rc = k == 0 ? mc_multi_vanilla_run_f32(h[i], &v32, MC_DEFAULT_SEED, 0, 1000, &r)
                             : k == 1 ? mc_multi_vanilla_run_f64(h[i], &v64, MC_DEFAULT_SEED, 0, 1000, &r)
                                      : mc_multi_cva_run_f64(h[i], &c64, MC_DEFAULT_SEED, 0, 1000, &r);
                if (rc != MC_OK) { fprintf(stderr, "warm-up failed: %s\n", mc_multi_last_error()); return 2; }
            }
            printf("first call of kernel %d on the handles (code-object load, not in the percentiles): fan-out %.1f us threaded, %.1f us serial\n", k,
                   mc_multi_last_fanout_us(threaded), mc_multi_last_fanout_us(serial));
        }
    }
    const double t0 = now_s();
    double fan_sum = 0, fan_max = 0;
    float *fan_t = malloc(sizeof(float) * (size_t)calls), *fan_s = malloc(sizeof(float) * (size_t)calls);
    Slow slow[N_SLOW];
    int n_slow = 0;
    double last_return = now_s();
    for (long i = 0; i < calls; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        const uint64_t n = 5 + (lcg >> 8) % 40000, first = (uint64_t)i * 1000003ull;   /* also fewer paths than devices */
        mc_result a, b;
        int ra, rb;
        const double gap_ms = (now_s() - last_return) * 1e3;
        switch (i % 3) {
        case 0: ra = mc_multi_vanilla_run_f32(threaded, &v32, MC_DEFAULT_SEED, first, n, &a), rb = mc_multi_vanilla_run_f32(serial, &v32, MC_DEFAULT_SEED, first, n, &b); break;
        case 1: ra = mc_multi_vanilla_run_f64(threaded, &v64, MC_DEFAULT_SEED, first, n, &a), rb = mc_multi_vanilla_run_f64(serial, &v64, MC_DEFAULT_SEED, first, n, &b); break;
        default: ra = mc_multi_cva_run_f64(threaded, &c64, MC_DEFAULT_SEED, first, 1 + n / 16, &a), rb = mc_multi_cva_run_f64(serial, &c64, MC_DEFAULT_SEED, first, 1 + n / 16, &b); break;
        }
        if (ra != MC_OK || rb != MC_OK) { fprintf(stderr, "call %ld failed: %s\n", i, mc_multi_last_error()); return 2; }
        last_return = now_s();
        const double f = mc_multi_last_fanout_us(threaded);
        fan_sum += f, fan_max = f > fan_max ? f : fan_max;
        fan_t[i] = (float)f, fan_s[i] = (float)mc_multi_last_fanout_us(serial);
        if (n_slow < N_SLOW || f > slow[N_SLOW - 1].fan) {      /* keep the twenty slowest, sorted, with their per-device trace */
            int k = n_slow < N_SLOW ? n_slow++ : N_SLOW - 1;
            while (k > 0 && slow[k - 1].fan < f) {
                slow[k] = slow[k - 1];
                --k;
            }
            slow[k].fan = f, slow[k].call = i, slow[k].gap_ms = gap_ms;
            mc_multi_last_fanout_trace(threaded, 8, slow[k].seen, slow[k].at);
        }
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
    percentiles("8 launcher threads", fan_t, calls);
    percentiles("serial fan-out (MC_MULTI_THREADS=0)", fan_s, calls);
    mc_multi_fanout_counts c;
    mc_multi_fanout_stats(threaded, &c);
    printf("jobs of the threaded handle: %llu calls; %llu run by their launcher thread, %llu by the caller because the thread was asleep, %llu because it was late "
           "(> 15 us without taking the job); %llu took > 1 ms INSIDE their thread's claimed job; %llu wake-ups\n",
           (unsigned long long)c.calls, (unsigned long long)c.by_worker, (unsigned long long)c.served_parked, (unsigned long long)c.stolen,
           (unsigned long long)c.slow_claimed, (unsigned long long)c.wakeups);
    char cfg[1024];
    mc_multi_describe(threaded, cfg, (int)sizeof cfg);
    printf("%s\n", cfg);
    printf("the %d slowest threaded calls -- per device: saw the call -> launch enqueued, us since call entry (-1: caller took over a late thread's job, "
           "-2: caller served a sleeping thread):\n", n_slow);
    for (int k = 0; k < n_slow; ++k) {
        printf("  call %7ld  fan-out %9.1f us  (%.3f ms after the previous call returned):", slow[k].call, slow[k].fan, slow[k].gap_ms);
        for (int g = 0; g < 8; ++g)
            printf("  [%d] %.1f -> %.1f", g, slow[k].seen[g], slow[k].at[g]);
        printf("\n");
    }
    mc_multi_destroy(threaded);
    mc_multi_destroy(serial);
    free(fan_t), free(fan_s);
    return bad ? 1 : 0;
}
