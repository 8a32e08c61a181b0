/* engine_fanout.cpp -- the ENGINE's launch (mc_cva_launch_f64: arm + argument set-up + one kernel launch) from T plain
 * pthreads at once, one context per thread, all on device 0.  tools/c/launch_contention.hip shows the HIP runtime issuing
 * empty launches of different streams side by side (8 threads: fan-out 6.4 us against 8 x 2.8 serial); libmc_multi's
 * launcher threads measured no gain for the same fan-out (24 us, profiles/r04_multi_fixed_cost_and_fanout.log).  This
 * program sits between the two: no LaunchCrew, no polling caller -- only the engine's own launch path under concurrency,
 * split into its phases (arm_direct / launch), so that what serialises can be named.
 *   hipcc -O2 -Iinclude tools/c/engine_fanout.cpp -Lmontecarlocuda_amd/csrc -lmc_mi355x -lpthread \
 *       -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -o tools/c/engine_fanout      (host code only; hipcc for hipMalloc) */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mc_mi355x.h"
#include "../../montecarlocuda_amd/csrc/mc_multi_host.hpp"

enum { TMAX = 8, ROUNDS = 300 };

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }
static double median(double *t, int n) { qsort(t, (size_t)n, sizeof *t, cmp); return t[n / 2]; }

static mc_context *ctx[TMAX];
static double *d_triple[TMAX];
static const volatile double *slot[TMAX];
static std::atomic<int> go, done_count, quit_flag;
static double t_arm[TMAX][ROUNDS], t_launch[TMAX][ROUNDS], t_end[TMAX][ROUNDS];
static uint64_t paths_each;
static int use_vanilla, set_device, use_crew;
static const mc_cva_f64 cva = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};
static const mc_option_f32 van = {100.f, 100.f, 0.04879f, 0.2f, 1.f};

static int one_launch(int t, int r)
{
    const double a = now_us();
    if (set_device && hipSetDevice(0) != hipSuccess) return 1;   /* what libmc_multi's device_part does first */
    if (mc_context_arm_direct(ctx[t], &slot[t]) != MC_OK) return 1;
    const double b = now_us();
    int rc;
    if (use_vanilla)
        rc = mc_vanilla_launch_f32(ctx[t], &van, MC_DEFAULT_SEED, (uint64_t)t * paths_each, paths_each, d_triple[t], mc_context_stream(ctx[t]));
    else
        rc = mc_cva_launch_f64(ctx[t], &cva, MC_DEFAULT_SEED, (uint64_t)t * paths_each, paths_each, d_triple[t], mc_context_stream(ctx[t]));
    const double c = now_us();
    if (rc != MC_OK) { fprintf(stderr, "launch: %s\n", mc_last_error()); return 1; }
    if (r >= 0) t_arm[t][r] = b - a, t_launch[t][r] = c - b, t_end[t][r] = c;
    return 0;
}

static void wait_slot(int t)
{
    while (slot[t][2] == -1.0) { }
}

static void *worker(void *arg)
{
    const int t = (int)(long)arg;
    int seen = 0;
    for (;;) {
        int cur;
        while ((cur = go.load(std::memory_order_acquire)) == seen)
            if (quit_flag.load(std::memory_order_relaxed)) return NULL;
        seen = cur;
        if (one_launch(t, cur - 1 - 20)) exit(1);
        wait_slot(t);
        done_count.fetch_add(1, std::memory_order_release);
    }
}

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i)
        use_vanilla |= !strcmp(argv[i], "vanilla"), set_device |= !strcmp(argv[i], "setdevice"), use_crew |= !strcmp(argv[i], "crew");
    if (set_device) printf("# with hipSetDevice(0) before every arm (counted in the arm column)\n");
    const uint64_t total = use_vanilla ? 8000000ull : 1250000ull;
    printf("# engine_fanout: %s launches, arm_direct + launch per thread, one context per thread on device 0; %d rounds after 20 warm-ups\n",
           use_vanilla ? "mc_vanilla_launch_f32 (1e6 paths each)" : "mc_cva_launch_f64 (C5 shard of 8 split again over the threads)", ROUNDS);
    printf("%8s %16s %16s %22s\n", "threads", "arm us (med)", "launch us (med)", "fan-out us (med)");
    for (int T = 1; T <= TMAX; T *= 2) {
        paths_each = total / (uint64_t)T;
        for (int t = 0; t < T; ++t) {
            if (mc_context_create(0, 0, &ctx[t]) != MC_OK) { fprintf(stderr, "%s\n", mc_last_error()); return 1; }
            if (hipMalloc((void **)&d_triple[t], 3 * sizeof(double)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
        }
        go.store(0), done_count.store(0), quit_flag.store(0);
        if (use_crew) {   /* the same launches through libmc_multi's LaunchCrew: T launcher threads, the caller only hands off and waits */
            mc_host::LaunchCrew crew(T, std::chrono::microseconds(2000));
            static double fanc[ROUNDS];
            static int round_idx;
            for (int r = -20; r < ROUNDS; ++r) {
                int rc[TMAX];
                round_idx = r;
                const double t0 = now_us();
                crew.run_all([](void *, int g) -> int { return one_launch(g, round_idx); }, nullptr, rc);
                for (int t = 0; t < T; ++t) {
                    if (rc[t]) return 1;
                    wait_slot(t);
                }
                if (r >= 0) {
                    double last = 0;
                    for (int t = 0; t < T; ++t)
                        last = t_end[t][r] > last ? t_end[t][r] : last;
                    fanc[r] = last - t0;
                }
            }
            static double a2[TMAX * ROUNDS], l2[TMAX * ROUNDS];
            int n2 = 0;
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < ROUNDS; ++r)
                    a2[n2] = t_arm[t][r], l2[n2] = t_launch[t][r], ++n2;
            printf("%8d %16.2f %16.2f %22.2f   (LaunchCrew%s)\n", T, median(a2, n2), median(l2, n2), median(fanc, ROUNDS), crew.yields() ? ", yielding" : "");
            for (int t = 0; t < T; ++t)
                mc_context_destroy(ctx[t]), (void)hipFree(d_triple[t]);
            continue;
        }
        pthread_t th[TMAX];
        for (int t = 1; t < T; ++t)
            pthread_create(&th[t], NULL, worker, (void *)(long)t);
        static double fan[ROUNDS];
        for (int r = 1; r <= ROUNDS + 20; ++r) {
            done_count.store(0);
            const double t0 = now_us();
            go.store(r, std::memory_order_release);
            if (one_launch(0, r - 1 - 20)) return 1;
            wait_slot(0);
            while (done_count.load(std::memory_order_acquire) != T - 1) { }
            if (r > 20) {
                double last = 0;
                for (int t = 0; t < T; ++t)
                    last = t_end[t][r - 21] > last ? t_end[t][r - 21] : last;
                fan[r - 21] = last - t0;
            }
        }
        quit_flag.store(1);
        for (int t = 1; t < T; ++t)
            pthread_join(th[t], NULL);
        static double a[TMAX * ROUNDS], l[TMAX * ROUNDS];
        int n = 0;
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < ROUNDS; ++r)
                a[n] = t_arm[t][r], l[n] = t_launch[t][r], ++n;
        printf("%8d %16.2f %16.2f %22.2f\n", T, median(a, n), median(l, n), median(fan, ROUNDS));
        for (int t = 0; t < T; ++t)
            mc_context_destroy(ctx[t]), (void)hipFree(d_triple[t]);
    }
    return 0;
}
