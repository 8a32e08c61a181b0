/*
 * multiBench.c -- strong scaling of ONE pricing call over the GPUs of a node, from one C process.
 *
 * BASELINE.json configs[3] (basket call, 16 correlated assets, 1e9 paths, fp64) and configs[4] (CVA, 256 dates x
 * 1e7 paths, fp64) -- and 10x those sizes, as SURVEY 8e asks -- sharded over G = 1, 2, 4, 8 ... devices with
 * mc_multi_* (include/mc_multi.h: shards + one RCCL all-reduce of the 24-byte triple).  Timed as SURVEY 8d/8e
 * prescribe: wall-clock from the first launch to the all-reduced, closed estimate on the host; 2 warm-ups, then
 * `reps` calls (without the per-device HIP events unless --events; one more call with them gives kernel_ms), median and
 * minimum reported; handle (contexts + RCCL communicators) creation reported once,
 * separately.  One JSON object per line on stdout: bench.py embeds them ("c_multi"), people read them.
 *
 * Then, on the first device alone, shard 0 of G = 2, 4, 8 of every workload: the device side of the scaling curve,
 * measurable without the other devices ("shard_of" lines).
 *
 *   multiBench [--reps R] [--max-devices G] [--small] [--events]    (--small: 1/100 of the sizes, for tests)
 */
#include "driver_util.h"
#include "mc_multi.h"

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}

#define CHECK(call)                                                                                                   \
    do {                                                                                                              \
        if ((call) != MC_OK) {                                                                                        \
            printf("{\"error\": \"%s failed: %s | %s\"}\n", #call, mc_multi_last_error(), mc_last_error());          \
            return 1;                                                                                                 \
        }                                                                                                             \
    } while (0)

int main(int argc, char **argv)
{
    int reps = 5, max_devices = 64, small = 0, events = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-devices") && i + 1 < argc) max_devices = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--small")) small = 1;
        else if (!strcmp(argv[i], "--events")) events = 1;      /* keep the per-device HIP events during the timed calls too */
        else if (!strcmp(argv[i], "--no-events")) events = 0;   /* the default, accepted for old command lines */
        else {
            fprintf(stderr, "usage: %s [--reps R] [--max-devices G] [--small] [--events]\n", argv[0]);
            return 1;
        }
    }
    if (reps < 1 || reps > 100) reps = 5;
    const int visible = mc_device_count();
    if (visible < 1) {
        printf("{\"error\": \"no HIP device visible\"}\n");
        return 1;
    }
    /* C4: n = 16, S = 100, w = 1/16, vols alternating 0.3 / 0.2, equicorrelation 0.5, K = 100, r = 0.048790164, T = 1 */
    enum { NA = 16 };
    double corr[NA * NA], L[NA * NA], s[NA], v[NA], d[NA], w[NA];
    for (int i = 0; i < NA; ++i) {
        s[i] = 100, v[i] = i % 2 ? 0.2 : 0.3, d[i] = 0, w[i] = 1.0 / NA;
        for (int j = 0; j < NA; ++j)
            corr[NA * i + j] = i == j ? 1.0 : 0.5;
    }
    if (mc_chol_f64(NA, corr, L) != 0)
        return 1;
    const mc_basket_f64 c4 = {NA, s, v, L, d, w, 100., 1., 0.048790164};
    const mc_cva_f64 c5 = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};   /* cvaOpt.cu:22-34 with 256 dates */
    const uint64_t scale = small ? 100 : 1;
    struct { const char *name; int is_cva; uint64_t paths; } work[4] = {
        {"C4 basket n=16 f64, 1e9 paths", 0, 1000000000ull / scale}, {"C4 x10: basket n=16 f64, 1e10 paths", 0, 10000000000ull / scale},
        {"C5 CVA 256 dates f64, 1e7 paths", 1, 10000000ull / scale}, {"C5 x10: CVA 256 dates f64, 1e8 paths", 1, 100000000ull / scale}};
    double t1[4] = {0, 0, 0, 0};   /* median at G = 1, for the efficiency column */
    for (int G = 1; G <= visible && G <= max_devices; G *= 2) {
        mc_multi *m;
        const double t_create0 = now_s();
        CHECK(mc_multi_create(NULL, G, 0, &m));
        mc_result r;
        CHECK(mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, 0, 1000, &r));   /* creates the RCCL communicators */
        const double create_s = now_s() - t_create0;
        printf("{\"devices\": %d, \"create_s\": %.3f, \"what\": \"contexts + ncclCommInitAll + first call, once per handle\"}\n", G, create_s);
        for (int k = 0; k < 4; ++k) {
            double t[100];
            /* the timed calls run without the per-device HIP events (two runtime calls less per device on the launching
             * thread); call `reps` (not timed) runs with them, for kernel_ms */
            for (int i = -2; i <= reps; ++i) {
                CHECK(mc_multi_set_timing(m, events || i == reps));
                const double t0 = now_s();
                if (work[k].is_cva)
                    CHECK(mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, 0, work[k].paths, &r));
                else
                    CHECK(mc_multi_basket_run_f64(m, &c4, MC_DEFAULT_SEED, 0, work[k].paths, &r));
                if (i >= 0 && i < reps)
                    t[i] = now_s() - t0;
            }
            qsort(t, (size_t)reps, sizeof t[0], cmp_double);
            const double med = t[reps / 2];
            if (G == 1)
                t1[k] = med;
            printf("{\"devices\": %d, \"workload\": \"%s\", \"paths\": %llu, \"reps\": %d, \"wall_ms_median\": %.4f, \"wall_ms_min\": %.4f, "
                   "\"paths_per_s\": %.6g, \"strong_efficiency_vs_1\": %.4f, \"kernel_ms_slowest_device\": %.4f, \"value\": %.9g, "
                   "\"confidence_95\": %.3g, \"rccl_vs_host_rel\": %.3g}\n",
                   G, work[k].name, (unsigned long long)work[k].paths, reps, med * 1e3, t[0] * 1e3, (double)work[k].paths / med,
                   t1[k] > 0 ? t1[k] / (G * med) : 0.0, r.kernel_ms, r.expected, r.confidence, mc_multi_last_reduce_error(m));
            fflush(stdout);
        }
        mc_multi_destroy(m);
    }
    /* What ONE device does at G = 2, 4, 8, measured on the first device alone: shard 0 of G of every workload through
     * the same call (launch, all-reduce over a communicator of one, read-back, closing).  T(1) / (G T(shard)) is the
     * strong-scaling efficiency the device side allows -- everything except the G-rank all-reduce's extra latency --
     * and it can be measured on a one-GPU box. */
    {
        mc_multi *m;
        mc_result r;
        CHECK(mc_multi_create(NULL, 1, 0, &m));
        CHECK(mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, 0, 1000, &r));
        for (int G = 2; G <= 8; G *= 2)
            for (int k = 0; k < 4; ++k) {
                uint64_t lo = 0, cnt = 0;
                mc_shard_range(work[k].paths, 0, G, &lo, &cnt);
                double t[100];
                for (int i = -2; i <= reps; ++i) {
                    CHECK(mc_multi_set_timing(m, events || i == reps));
                    const double t0 = now_s();
                    if (work[k].is_cva)
                        CHECK(mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, lo, cnt, &r));
                    else
                        CHECK(mc_multi_basket_run_f64(m, &c4, MC_DEFAULT_SEED, lo, cnt, &r));
                    if (i >= 0 && i < reps)
                        t[i] = now_s() - t0;
                }
                qsort(t, (size_t)reps, sizeof t[0], cmp_double);
                const double med = t[reps / 2];
                printf("{\"shard_of\": %d, \"devices\": 1, \"workload\": \"%s\", \"paths\": %llu, \"reps\": %d, \"wall_ms_median\": %.4f, "
                       "\"wall_ms_min\": %.4f, \"kernel_ms\": %.4f, \"device_side_efficiency\": %.4f, "
                       "\"what\": \"shard 0 of %d on one device: T(1) / (%d T(shard)), the all-reduce between devices not included\"}\n",
                       G, work[k].name, (unsigned long long)cnt, reps, med * 1e3, t[0] * 1e3, r.kernel_ms,
                       t1[k] > 0 ? t1[k] / (G * med) : 0.0, G, G);
                fflush(stdout);
            }
        mc_multi_destroy(m);
    }
    return 0;
}
