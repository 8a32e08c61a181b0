// shard_clock.hip -- is C5's shard of 8 slower than T(1)/8 because of wave quantisation or because of the clock?
// (VERDICT r03 "weak" #2 / "next" #1a).  One context on device 0, the CVA kernel of BASELINE configs[4]
// (256 dates, fp64), launches timed on the device by HIP events.
//
//   A  hot sweep     after a 300 ms pre-heat, path counts from 1.00e6 to 1.45e6 back to back (asynchronous launches,
//                    30 per size): a staircase in time-per-launch at multiples of 65 536 paths (one wave-trip per SIMD)
//                    would be quantisation; a straight line through the origin is throughput
//   B  gaps          the 1.25e6-path launch (and the 1e7-path one) as synchronous calls separated by host sleeps of
//                    0 ... 200 ms: what a cooling / re-ramping clock does to the same launch
//   C  harness       the sequence drivers/multiBench runs for a row: 2 warm-ups + 10 calls of T(1), then of the shard,
//                    without and with a 300 ms pre-heat before each
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/c/shard_clock.hip -Lmontecarlocuda_amd/csrc -lmc_mi355x
//         -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -o tools/c/shard_clock
// Under rocprofv3 (--kernel-trace --stats, or --pmc GRBM_GUI_ACTIVE for the clock) run `tools/c/shard_clock A` etc.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mc_mi355x.h"

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define MC(call)                                                                  \
    do {                                                                          \
        if ((call) != MC_OK) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #call, mc_last_error());          \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static const mc_cva_f64 CVA = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};

static int preheat(mc_context *c, double *d_triple, double ms)
{
    const double t0 = now_ms();
    uint64_t j = 0;
    while (now_ms() - t0 < ms) {
        for (int i = 0; i < 4; ++i, ++j)
            MC(mc_cva_launch_f64(c, &CVA, MC_DEFAULT_SEED, (1ull << 40) + j * 10000000ull, 10000000ull, d_triple, mc_context_stream(c)));
        if (hipStreamSynchronize((hipStream_t)mc_context_stream(c)) != hipSuccess) return 1;
    }
    return 0;
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "ABCD";
    mc_context *c;
    MC(mc_context_create(0, 0, &c));
    double *d_triple = nullptr;
    if (hipMalloc(&d_triple, 3 * sizeof(double)) != hipSuccess) return 1;
    hipStream_t st = (hipStream_t)mc_context_stream(c);
    char name[128];
    int cus = 0, mhz = 0;
    mc_context_info(c, name, sizeof name, &cus, &mhz);
    printf("device: %s, %d CUs, %d MHz nominal; CVA 256 dates fp64; grid rule: 12 workgroups per CU\n", name, cus, mhz);

    // per-launch durations of a back-to-back burst: every launch between its own pair of stream events; all launches are
    // queued before the first is read, so the device never idles
    auto timed_burst = [&](uint64_t n, int reps, std::vector<double> &us) -> int {
        us.clear();
        std::vector<hipEvent_t> e0(reps), e1(reps);
        for (int i = 0; i < reps; ++i) {
            if (hipEventCreate(&e0[i]) != hipSuccess || hipEventCreate(&e1[i]) != hipSuccess) return 1;
        }
        for (int i = 0; i < reps; ++i) {
            if (hipEventRecord(e0[i], st) != hipSuccess) return 1;
            MC(mc_cva_launch_f64(c, &CVA, MC_DEFAULT_SEED, (uint64_t)i * n, n, d_triple, st));
            if (hipEventRecord(e1[i], st) != hipSuccess) return 1;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        for (int i = 0; i < reps; ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0[i], e1[i]) != hipSuccess) return 1;
            us.push_back(ms * 1e3);
            hipEventDestroy(e0[i]);
            hipEventDestroy(e1[i]);
        }
        return 0;
    };
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mn = [](const std::vector<double> &v) { return *std::min_element(v.begin(), v.end()); };
    auto mx = [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };

    if (strchr(which, 'A')) {
        printf("\nA. hot sweep (300 ms pre-heat, then 30 back-to-back launches per size; stream-event brackets, so each figure\n"
               "   includes the ~2 us launch boundary).  65 536 paths = one wave-trip on each of the 1024 SIMDs.\n"
               "   lanes = 1: cva_kernel alone (one lane per path); auto: the last partial trip on cva_dates_kernel beside it\n");
        printf("%10s %12s | %10s %10s %10s | %10s %10s %10s | %8s\n", "paths", "trips/SIMD", "1: med us", "min", "max", "auto: med", "min", "max", "auto/1");
        if (preheat(c, d_triple, 300)) return 1;
        const uint64_t sizes[] = {1000000, 1048576, 1100000, 1114112, 1150000, 1179648, 1200000, 1245184, 1250000, 1280000, 1310720,
                                  1340000, 1376256, 1400000, 1441792, 1450000, 2500000, 5000000, 10000000};
        std::vector<double> us, us2;
        for (uint64_t n : sizes) {
            MC(mc_context_set_cva_date_lanes(c, 1));
            if (timed_burst(n, n > 2000000 ? 12 : 30, us)) return 1;
            MC(mc_context_set_cva_date_lanes(c, 0));
            if (timed_burst(n, n > 2000000 ? 12 : 30, us2)) return 1;
            printf("%10llu %12.3f | %10.1f %10.1f %10.1f | %10.1f %10.1f %10.1f | %8.4f\n", (unsigned long long)n, n / 65536.0, med(us), mn(us), mx(us),
                   med(us2), mn(us2), mx(us2), med(us2) / med(us));
        }
    }
    if (strchr(which, 'D')) {
        printf("\nD. small calls, hot (pre-heat, 40 back-to-back launches each): the whole call on cva_dates_kernel with L lanes per path\n"
               "   against cva_kernel (L = 1); 131 072 paths is the reference driver's own call (dp/cvaOpt.cu:12-15)\n");
        printf("%10s %8s", "paths", "auto us");
        for (int l = 1; l <= 64; l *= 2) printf(" %8s%-2d", "L=", l);
        printf("\n");
        if (preheat(c, d_triple, 300)) return 1;
        std::vector<double> us;
        for (uint64_t n : {4096ull, 16384ull, 32768ull, 65536ull, 98304ull, 131072ull, 196608ull, 262144ull, 393216ull, 524288ull, 786432ull, 1048576ull}) {
            MC(mc_context_set_cva_date_lanes(c, 0));
            if (timed_burst(n, 40, us)) return 1;
            printf("%10llu %8.1f", (unsigned long long)n, med(us));
            for (int l = 1; l <= 64; l *= 2) {
                MC(mc_context_set_cva_date_lanes(c, l));
                if (timed_burst(n, 40, us)) return 1;
                printf(" %10.1f", med(us));
            }
            printf("\n");
        }
        MC(mc_context_set_cva_date_lanes(c, 0));
    }
    if (strchr(which, 'B')) {
        printf("\nB. the same launch as synchronous calls (mc_cva_run_f64, timing on: kernel_ms from events) separated by host sleeps;\n"
               "   12 calls per gap after a 300 ms pre-heat, in this order\n");
        printf("%10s %10s %12s %10s %10s %12s\n", "paths", "gap ms", "kernel med", "min", "max", "wall med us");
        mc_context_set_timing(c, 1);
        for (uint64_t n : {1250000ull, 10000000ull}) {
            if (preheat(c, d_triple, 300)) return 1;
            for (double gap : {0.0, 0.05, 0.2, 1.0, 5.0, 20.0, 100.0, 0.0}) {
                std::vector<double> k, w;
                for (int i = 0; i < 12; ++i) {
                    mc_result r;
                    MC(mc_cva_run_f64(c, &CVA, MC_DEFAULT_SEED, (uint64_t)i * n, n, &r));
                    k.push_back(r.kernel_ms * 1e3), w.push_back(r.wall_ms * 1e3);
                    if (gap > 0) usleep((useconds_t)(gap * 1e3));
                }
                printf("%10llu %10.2f %12.1f %10.1f %10.1f %12.1f\n", (unsigned long long)n, gap, med(k), mn(k), mx(k), med(w));
            }
        }
    }
    if (strchr(which, 'C')) {
        printf("\nC. the strong-row harness: T(1) = 1e7 paths, shard of 8 = 1.25e6 paths; 2 warm-ups + 10 synchronous calls each\n"
               "   (timing off: pinned-slot read-back, wall-clock per call), cold = straight after 1 s of idle, hot = after a 300 ms pre-heat\n");
        printf("%6s %10s %12s %10s %10s %14s\n", "state", "paths", "wall med us", "min", "max", "T(1)/(8 T(s))");
        mc_context_set_timing(c, 0);
        for (int hot = 0; hot < 2; ++hot)
            for (int pass = 0; pass < 2; ++pass) {
                double t1 = 0;
                for (uint64_t n : {10000000ull, 1250000ull}) {
                    if (hot) { if (preheat(c, d_triple, 300)) return 1; } else sleep(1);
                    std::vector<double> w;
                    for (int i = -2; i < 10; ++i) {
                        mc_result r;
                        const double t0 = now_ms();
                        MC(mc_cva_run_f64(c, &CVA, MC_DEFAULT_SEED, 0, n, &r));
                        if (i >= 0) w.push_back((now_ms() - t0) * 1e3);
                    }
                    const double m = med(w);
                    if (n == 10000000ull) t1 = m;
                    printf("%6s %10llu %12.1f %10.1f %10.1f %14.4f\n", hot ? "hot" : "cold", (unsigned long long)n, m, mn(w), mx(w),
                           n == 10000000ull ? 1.0 : t1 / (8 * m));
                }
            }
    }
    hipFree(d_triple);
    mc_context_destroy(c);
    return 0;
}
