/*
 * multiBench.c -- strong scaling of ONE pricing call over the GPUs of a node, from one C process.
 *
 * BASELINE.json configs[3] (basket call, 16 correlated assets, 1e9 paths, fp64) and configs[4] (CVA, 256 dates x
 * 1e7 paths, fp64) -- and 10x those sizes, as SURVEY 8e asks -- sharded over G = 1, 2, 4, 8 ... devices with
 * mc_multi_* (include/mc_multi.h: shards + one RCCL all-reduce of the 24-byte triple); the same four once more on fp32
 * normals (mc_multi_set_normals(MC_NORMALS_F32): the reference's own dp arithmetic, dp/MonteCarloKernel.cu:68,78,250 --
 * shorter kernels, so fixed costs weigh more).  Timed as SURVEY 8d/8e prescribe: wall-clock from the first launch to
 * the all-reduced, closed estimate on the host, median (and minimum) of `reps` >= 10 calls.
 *
 * Clock conditioning (profiles/r04_shard_clock_quantisation_vs_dvfs.log): an MI355X that has idled runs its first ~30 ms
 * of work at 1.9-2.2 GHz, a long fp64 launch settles at 2.30 GHz and 1 ms launches back to back at 2.38 GHz; C5's 1 ms
 * shard measured after two warm-up calls took 1.15-1.19 ms, after 300 ms of load 0.97-0.98 ms.  So every row -- T(1) and
 * T(shard) alike -- is measured HOT: the row's own call repeated for >= 300 ms (at least twice) on every device, then
 * `reps` calls.  The base-size rows are measured COLD as well (0.5 s of idle, two warm-up calls, 5 calls: what a
 * one-off call sees) and reported beside it; efficiencies are hot / hot and cold / cold.
 *
 * Handle (contexts + RCCL communicators) creation is reported once, separately.  One JSON object per line on stdout:
 * bench.py embeds them ("c_multi"), people read them.
 *
 * Then, on the first device alone, shard 0 of G = 2, 4, 8 of every workload: the device side of the scaling curve,
 * measurable without the other devices ("shard_of" lines).
 *
 *   multiBench [--reps R] [--max-devices G] [--small] [--events] [--no-cold] [--no-n32] [--brief] [--preheat-ms MS]
 *              (--small: 1/100 of the sizes and no pre-heat, for tests;
 *               --brief: what bench.py's default run embeds -- the base sizes only (C4, C5, and both on fp32 normals), shard 0
 *               of 8 only, and the G = 1 handle re-used for the shard rows: one RCCL communicator set-up instead of two)
 */
#include <unistd.h>

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

enum { NA = 16, MAX_REPS = 100 };
typedef struct {
    const char *config;      /* the key bench.py's rows use: C4, C4x10, C5, C5x10, and the same with _n32 */
    const char *name;
    int is_cva, n32, base;   /* base: a BASELINE size (measured cold as well) */
    uint64_t paths;
} Work;

static mc_basket_f64 c4;
static const mc_cva_f64 c5 = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};   /* cvaOpt.cu:22-34 with 256 dates */

static int one_call(mc_multi *m, const Work *w, uint64_t first, uint64_t n, mc_result *r)
{
    return w->is_cva ? mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, first, n, r) : mc_multi_basket_run_f64(m, &c4, MC_DEFAULT_SEED, first, n, r);
}

typedef struct {
    double med, min, fanout_us, fanout_us_max, preheat_ms;
    double collective_us;          /* median over the timed calls: last device's own triple on the host -> all-reduced triple on the host (-1: not measured) */
    double device_us_min, device_us_max;   /* last timed call: earliest / latest device delivery since call entry */
    int calls_preheat;
} Timing;

/* hot: the row's own call for >= preheat_ms (at least twice), then `reps` timed calls.  cold: 0.5 s idle, two warm-ups. */
static int time_row(mc_multi *m, const Work *w, uint64_t first, uint64_t n, int reps, int events, int hot, double preheat_ms, Timing *out,
                    mc_result *r)
{
    double t[MAX_REPS];
    CHECK(mc_multi_set_normals(m, w->n32 ? MC_NORMALS_F32 : MC_NORMALS_NATIVE));
    CHECK(mc_multi_set_timing(m, events));
    out->calls_preheat = 0;
    const double p0 = now_s();
    if (!hot)
        usleep(500000);
    while (out->calls_preheat < 2 || (hot && (now_s() - p0) * 1e3 < preheat_ms)) {
        CHECK(one_call(m, w, first, n, r));
        out->calls_preheat++;
    }
    out->preheat_ms = hot ? (now_s() - p0) * 1e3 : 0.0;
    double fan = 0, fan_max = 0, coll[MAX_REPS];
    int n_coll = 0;
    for (int i = 0; i < reps; ++i) {
        const double t0 = now_s();
        CHECK(one_call(m, w, first, n, r));
        t[i] = now_s() - t0;
        const double f = mc_multi_last_fanout_us(m), c = mc_multi_last_collective_us(m);
        fan += f, fan_max = f > fan_max ? f : fan_max;
        if (c >= 0)
            coll[n_coll++] = c;
    }
    out->fanout_us_max = fan_max;
    qsort(t, (size_t)reps, sizeof t[0], cmp_double);
    out->med = t[reps / 2], out->min = t[0], out->fanout_us = fan / reps;
    qsort(coll, (size_t)n_coll, sizeof coll[0], cmp_double);
    out->collective_us = n_coll ? coll[n_coll / 2] : -1.0;
    double dev[64];
    const int G = mc_multi_last_device_us(m, 64, dev);
    out->device_us_min = out->device_us_max = -1.0;
    for (int g = 0; g < G && g < 64; ++g)
        if (dev[g] >= 0) {
            out->device_us_min = (out->device_us_min < 0 || dev[g] < out->device_us_min) ? dev[g] : out->device_us_min;
            out->device_us_max = dev[g] > out->device_us_max ? dev[g] : out->device_us_max;
        }
    return 0;
}

int main(int argc, char **argv)
{
    int reps = 10, max_devices = 64, small = 0, events = 0, cold = 1, n32 = 1, brief = 0;
    double preheat_ms = 300.0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-devices") && i + 1 < argc) max_devices = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--preheat-ms") && i + 1 < argc) preheat_ms = atof(argv[++i]);
        else if (!strcmp(argv[i], "--small")) small = 1;
        else if (!strcmp(argv[i], "--events")) events = 1;      /* keep the per-device HIP events during the timed calls too */
        else if (!strcmp(argv[i], "--no-events")) events = 0;   /* the default, accepted for old command lines */
        else if (!strcmp(argv[i], "--no-cold")) cold = 0;
        else if (!strcmp(argv[i], "--no-n32")) n32 = 0;
        else if (!strcmp(argv[i], "--brief")) brief = 1;
        else {
            fprintf(stderr, "usage: %s [--reps R] [--max-devices G] [--small] [--events] [--no-cold] [--no-n32] [--brief] [--preheat-ms MS]\n", argv[0]);
            return 1;
        }
    }
    if (reps < 1 || reps > MAX_REPS) reps = 10;
    if (small) preheat_ms = 0, cold = 0;
    const int visible = mc_device_count();
    if (visible < 1) {
        printf("{\"error\": \"no HIP device visible\"}\n");
        return 1;
    }
    /* C4: n = 16, S = 100, w = 1/16, vols alternating 0.3 / 0.2, equicorrelation 0.5, K = 100, r = 0.048790164, T = 1 */
    static double corr[NA * NA], L[NA * NA], s[NA], v[NA], d[NA], w[NA];
    for (int i = 0; i < NA; ++i) {
        s[i] = 100, v[i] = i % 2 ? 0.2 : 0.3, d[i] = 0, w[i] = 1.0 / NA;
        for (int j = 0; j < NA; ++j)
            corr[NA * i + j] = i == j ? 1.0 : 0.5;
    }
    if (mc_chol_f64(NA, corr, L) != 0)
        return 1;
    c4 = (mc_basket_f64){NA, s, v, L, d, w, 100., 1., 0.048790164};
    const uint64_t scale = small ? 100 : 1;
    const Work all_work[8] = {
        {"C4", "C4 basket n=16 f64, 1e9 paths", 0, 0, 1, 1000000000ull / scale},
        {"C4x10", "C4 x10: basket n=16 f64, 1e10 paths", 0, 0, 0, 10000000000ull / scale},
        {"C5", "C5 CVA 256 dates f64, 1e7 paths", 1, 0, 1, 10000000ull / scale},
        {"C5x10", "C5 x10: CVA 256 dates f64, 1e8 paths", 1, 0, 0, 100000000ull / scale},
        {"C4_n32", "C4 on fp32 normals: basket n=16 f64, 1e9 paths", 0, 1, 1, 1000000000ull / scale},
        {"C4x10_n32", "C4 x10 on fp32 normals: basket n=16 f64, 1e10 paths", 0, 1, 0, 10000000000ull / scale},
        {"C5_n32", "C5 on fp32 normals: CVA 256 dates f64, 1e7 paths", 1, 1, 1, 10000000ull / scale},
        {"C5x10_n32", "C5 x10 on fp32 normals: CVA 256 dates f64, 1e8 paths", 1, 1, 0, 100000000ull / scale}};
    Work work[8];
    int n_work = 0;
    for (int k = 0; k < (n32 ? 8 : 4); ++k)
        if (!brief || all_work[k].base)
            work[n_work++] = all_work[k];
    double t1[8] = {0}, t1_cold[8] = {0};   /* medians at G = 1, for the efficiency columns */
    mc_multi *first_handle = NULL;   /* --brief: the G = 1 handle serves the shard rows too */
    for (int G = 1; G <= visible && G <= max_devices; G *= 2) {
        mc_multi *m;
        const double t_create0 = now_s();
        CHECK(mc_multi_create(NULL, G, 0, &m));
        mc_result r;
        CHECK(mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, 0, 1000, &r));   /* creates the RCCL communicators */
        const double create_s = now_s() - t_create0;
        printf("{\"devices\": %d, \"create_s\": %.3f, \"launcher_threads\": %d, \"what\": \"contexts + ncclCommInitAll + first call, once per handle\"}\n", G,
               create_s, mc_multi_launcher_threads(m));
        for (int k = 0; k < n_work; ++k) {
            Timing hot, cd = {0};
            if (time_row(m, &work[k], 0, work[k].paths, reps, events, 1, preheat_ms, &hot, &r)) return 1;
            const mc_result res = r;
            const double rel = mc_multi_last_reduce_error(m);
            CHECK(mc_multi_set_timing(m, 1));   /* one more call with the per-device events, for kernel_ms (still hot) */
            CHECK(one_call(m, &work[k], 0, work[k].paths, &r));
            const float kernel_ms = r.kernel_ms;
            const int do_cold = cold && work[k].base;
            if (do_cold && time_row(m, &work[k], 0, work[k].paths, reps < 5 ? reps : 5, events, 0, 0, &cd, &r)) return 1;
            if (G == 1)
                t1[k] = hot.med, t1_cold[k] = cd.med;
            printf("{\"devices\": %d, \"config\": \"%s\", \"workload\": \"%s\", \"normals\": \"%s\", \"paths\": %llu, \"reps\": %d, \"preheat_ms\": %.0f, \"wall_ms_median\": %.4f, "
                   "\"wall_ms_min\": %.4f, \"paths_per_s\": %.6g, \"strong_efficiency_vs_1\": %.4f, \"kernel_ms_slowest_device\": %.4f, "
                   "\"fanout_us\": %.2f, \"fanout_us_max\": %.2f, \"collective_us\": %.2f, \"device_delivery_us\": [%.1f, %.1f], \"value\": %.9g, "
                   "\"confidence_95\": %.3g, \"rccl_vs_host_rel\": %.3g",
                   G, work[k].config, work[k].name, work[k].n32 ? "f32" : "f64", (unsigned long long)work[k].paths, reps, hot.preheat_ms, hot.med * 1e3,
                   hot.min * 1e3, (double)work[k].paths / hot.med, t1[k] > 0 ? t1[k] / (G * hot.med) : 0.0, kernel_ms, hot.fanout_us, hot.fanout_us_max, hot.collective_us, hot.device_us_min, hot.device_us_max,
                   res.expected, res.confidence, rel);
            if (do_cold)
                printf(", \"cold\": {\"wall_ms_median\": %.4f, \"wall_ms_min\": %.4f, \"strong_efficiency_vs_1\": %.4f, \"what\": \"0.5 s idle, 2 warm-up calls, 5 calls\"}",
                       cd.med * 1e3, cd.min * 1e3, t1_cold[k] > 0 ? t1_cold[k] / (G * cd.med) : 0.0);
            printf("}\n");
            fflush(stdout);
        }
        if (brief && G == 1)
            first_handle = m;
        else
            mc_multi_destroy(m);
    }
    /* What ONE device does at G = 2, 4, 8, measured on the first device alone: shard 0 of G of every workload through
     * the same call (launch, all-reduce over a communicator of one, read-back, closing).  T(1) / (G T(shard)) is the
     * strong-scaling efficiency the device side allows -- everything except the G-rank all-reduce's extra latency --
     * and it can be measured on a one-GPU box. */
    {
        mc_multi *m = first_handle;
        mc_result r;
        if (!m) {
            CHECK(mc_multi_create(NULL, 1, 0, &m));
            CHECK(mc_multi_cva_run_f64(m, &c5, MC_DEFAULT_SEED, 0, 1000, &r));
        }
        for (int G = brief ? 8 : 2; G <= 8; G *= 2)
            for (int k = 0; k < n_work; ++k) {
                uint64_t lo = 0, cnt = 0;
                mc_shard_range(work[k].paths, 0, G, &lo, &cnt);
                Timing hot, cd = {0};
                if (time_row(m, &work[k], lo, cnt, reps, events, 1, preheat_ms, &hot, &r)) return 1;
                CHECK(mc_multi_set_timing(m, 1));
                CHECK(one_call(m, &work[k], lo, cnt, &r));
                const float kernel_ms = r.kernel_ms;
                const int do_cold = cold && work[k].base;
                if (do_cold && time_row(m, &work[k], lo, cnt, reps < 5 ? reps : 5, events, 0, 0, &cd, &r)) return 1;
                printf("{\"shard_of\": %d, \"devices\": 1, \"config\": \"%s\", \"workload\": \"%s\", \"normals\": \"%s\", \"paths\": %llu, \"reps\": %d, \"preheat_ms\": %.0f, "
                       "\"wall_ms_median\": %.4f, \"wall_ms_min\": %.4f, \"kernel_ms\": %.4f, \"device_side_efficiency\": %.4f",
                       G, work[k].config, work[k].name, work[k].n32 ? "f32" : "f64", (unsigned long long)cnt, reps, hot.preheat_ms, hot.med * 1e3, hot.min * 1e3,
                       kernel_ms,
                       t1[k] > 0 ? t1[k] / (G * hot.med) : 0.0);
                if (do_cold)
                    printf(", \"cold\": {\"wall_ms_median\": %.4f, \"device_side_efficiency\": %.4f}", cd.med * 1e3,
                           t1_cold[k] > 0 ? t1_cold[k] / (G * cd.med) : 0.0);
                printf(", \"what\": \"shard 0 of %d on one device: T(1) / (%d T(shard)), hot / hot; the all-reduce between devices not included\"}\n", G, G);
                fflush(stdout);
            }
        mc_multi_destroy(m);
    }
    return 0;
}
