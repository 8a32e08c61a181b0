/*
 * multi_check.c -- C-level check of libmc_multi.so (include/mc_multi.h) on whatever GPUs the box has.
 *
 * For every product and precision: the sharded call over {0} (RCCL communicator of one), over {0,0,0}
 * (three shards on one device, host reduction -- RCCL refuses a repeated device), and -- when more than
 * one device is visible -- over ALL devices with the RCCL all-reduce, must reproduce the single-device
 * call mc_*_run_* (n, and sum / sum2 up to the order of fp64 additions: 1e-12 relative in fp64, 2e-9 in fp32
 * where the shard boundaries move which 16 fp32 values are added before a flush to double), and the RCCL
 * result must equal the host sum of the device triples.  Also: fewer paths than devices, error paths.
 * Prints one line per comparison; exit status 0 iff everything held.  Run by tests/test_gpu_multi.py.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mc_multi.h"

static int failures;

static void compare(const char *what, const mc_result *a, const mc_result *b, double rel)
{
    const double e1 = fabs(a->sum - b->sum) / fabs(b->sum), e2 = fabs(a->sum2 - b->sum2) / fabs(b->sum2);
    const double ee = fabs(a->expected - b->expected) / fabs(b->expected), ec = fabs(a->confidence - b->confidence) / fabs(b->confidence);
    const int ok = a->n == b->n && e1 <= rel && e2 <= rel && ee <= rel && ec <= 10 * rel;
    printf("%-58s n=%llu  rel.err sum %.2e sum2 %.2e price %.2e ci %.2e  %s\n", what, (unsigned long long)a->n, e1, e2, ee, ec,
           ok ? "ok" : "MISMATCH");
    failures += !ok;
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        if ((call) != MC_OK) {                                                                        \
            fprintf(stderr, "%s failed: %s | %s\n", #call, mc_multi_last_error(), mc_last_error());   \
            return 2;                                                                                 \
        }                                                                                             \
    } while (0)

int main(void)
{
    const int visible = mc_device_count();
    printf("visible devices: %d\n", visible);
    if (visible < 1) {
        fprintf(stderr, "no GPU\n");
        return 2;
    }
    mc_context *one;
    CHECK(mc_context_create(0, 0, &one));
    mc_multi *m_one, *m_three, *m_all = NULL;
    /* the repeated-device handle: three contexts on device 0 by default; MULTI_CHECK_RANKS=8 makes it eight -- the node's own
     * size, for the call sequence, the launcher threads and the slot arrays at G = 8 (tests/test_gpu_multi.py) */
    const int d0 = 0, d000[16] = {0};
    int R = getenv("MULTI_CHECK_RANKS") ? atoi(getenv("MULTI_CHECK_RANKS")) : 3;
    if (R < 2 || R > 16)
        R = 3;
    char rep_name[64] = "{0";
    for (int i = 1; i < R; ++i)
        strcat(rep_name, ",0");
    strcat(rep_name, "}");
    CHECK(mc_multi_create(&d0, 1, 0, &m_one));
    CHECK(mc_multi_create(d000, R, 0, &m_three));
    /* MC_MULTI_ALLOW_REPEATED_DEVICES=1 (with tests/cpp/rccl_mock.hip preloaded: a test double of the collective whose ranks
     * may share a device): the three shards on device 0 go through the GROUPED all-reduce call sequence -- the G > 1 control
     * flow that real RCCL cannot be given on a one-GPU box */
    const char *rep = getenv("MC_MULTI_ALLOW_REPEATED_DEVICES");
    const int mock_collective = rep && atoi(rep) != 0;
    if (!mock_collective)
        CHECK(mc_multi_set_reduce(m_three, MC_REDUCE_HOST));
    if (visible > 1)
        CHECK(mc_multi_create(NULL, 0, 0, &m_all));
    printf("handles: {0} RCCL, %s %s%s\n", rep_name, mock_collective ? "grouped all-reduce through the preloaded test double" : "host reduce",
           m_all ? ", all devices RCCL" : "");

    const uint64_t seed = MC_DEFAULT_SEED;
    mc_multi *handles[3] = {m_one, m_three, m_all};
    char rep_label[96];
    snprintf(rep_label, sizeof rep_label, "%s %s", rep_name, mock_collective ? "mock collective" : "host");
    const char *names[3] = {"{0} rccl", rep_label, "all rccl"};
    char label[192];

    /* market data: the reference drivers' (vanillaOpt.cu:22-26, cvaOpt.cu:22-34), BASELINE's 4-asset basket */
    const mc_option_f32 v32 = {100.f, 100.f, 0.048790f, 0.2f, 1.f};
    const mc_option_f64 v64 = {100., 100., 0.048790, 0.2, 1.};
    const mc_cva_f32 c32 = {0.03f, 0.6f, {100.f, 100.f, 0.05f, 0.2f, 1.f}, 64};
    const mc_cva_f64 c64 = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 64};
    double corr[16], L64[16], s64[4], vol64[4], d64[4], w64[4];
    float L32[16], s32[4], vol32[4], d32[4], w32[4], corr32[16];
    for (int i = 0; i < 4; ++i) {
        s64[i] = 100, vol64[i] = i % 2 ? 0.2 : 0.3, d64[i] = 0, w64[i] = 0.25;
        s32[i] = 100, vol32[i] = i % 2 ? 0.2f : 0.3f, d32[i] = 0, w32[i] = 0.25f;
        for (int j = 0; j < 4; ++j)
            corr[4 * i + j] = i == j ? 1.0 : 0.5, corr32[4 * i + j] = i == j ? 1.f : 0.5f;
    }
    if (mc_chol_f64(4, corr, L64) != 0 || mc_chol_f32(4, corr32, L32) != 0)
        return 2;
    const mc_basket_f64 b64 = {4, s64, vol64, L64, d64, w64, 100., 1., 0.048790164};
    const mc_basket_f32 b32 = {4, s32, vol32, L32, d32, w32, 100.f, 1.f, 0.048790164f};

    const uint64_t first = 12345, n = 30000007;   /* unaligned, not a multiple of any device count */
    mc_result ref, got;
    for (int h = 0; h < 3; ++h) {
        mc_multi *m = handles[h];
        if (!m)
            continue;
        for (int anti = 0; anti < 2; ++anti) {
            mc_context_set_antithetic(one, anti);
            CHECK(mc_multi_set_antithetic(m, anti));
            CHECK(mc_vanilla_run_f32(one, &v32, seed, first, n, &ref));
            CHECK(mc_multi_vanilla_run_f32(m, &v32, seed, first, n, &got));
            snprintf(label, sizeof label, "vanilla f32 %s%s", names[h], anti ? " antithetic" : "");
            compare(label, &got, &ref, 2e-9);
            CHECK(mc_vanilla_run_f64(one, &v64, seed, first, n, &ref));
            CHECK(mc_multi_vanilla_run_f64(m, &v64, seed, first, n, &got));
            snprintf(label, sizeof label, "vanilla f64 %s%s", names[h], anti ? " antithetic" : "");
            compare(label, &got, &ref, 1e-12);
            CHECK(mc_basket_run_f32(one, &b32, seed, first, n / 4, &ref));
            CHECK(mc_multi_basket_run_f32(m, &b32, seed, first, n / 4, &got));
            snprintf(label, sizeof label, "basket4 f32 %s%s", names[h], anti ? " antithetic" : "");
            compare(label, &got, &ref, 2e-9);
            CHECK(mc_basket_run_f64(one, &b64, seed, first, n / 4, &ref));
            CHECK(mc_multi_basket_run_f64(m, &b64, seed, first, n / 4, &got));
            snprintf(label, sizeof label, "basket4 f64 %s%s", names[h], anti ? " antithetic" : "");
            compare(label, &got, &ref, 1e-12);
            CHECK(mc_cva_run_f32(one, &c32, seed, first, n / 64, &ref));
            CHECK(mc_multi_cva_run_f32(m, &c32, seed, first, n / 64, &got));
            snprintf(label, sizeof label, "cva64 f32 %s%s", names[h], anti ? " antithetic" : "");
            compare(label, &got, &ref, 2e-9);
            CHECK(mc_cva_run_f64(one, &c64, seed, first, n / 64, &ref));
            CHECK(mc_multi_cva_run_f64(m, &c64, seed, first, n / 64, &got));
            snprintf(label, sizeof label, "cva64 f64 %s%s", names[h], anti ? " antithetic" : "");
            compare(label, &got, &ref, 1e-12);
        }
        mc_context_set_antithetic(one, 0);
        CHECK(mc_multi_set_antithetic(m, 0));
        /* the two read-back forms of run_sharded: events + copies + synchronize (timing on, used so far) and the pinned
         * slots the devices write themselves, polled from user space (timing off).  Same kernels, same sums: same bits.
         * Many short calls back to back: a slot's sentinel and the tickets are reset between calls. */
        {
            mc_result on, off;
            CHECK(mc_multi_set_timing(m, 1));
            CHECK(mc_multi_cva_run_f64(m, &c64, seed, first, n / 64, &on));
            CHECK(mc_multi_set_timing(m, 0));
            for (int rep = 0; rep < 200; ++rep) {
                CHECK(mc_multi_cva_run_f64(m, &c64, seed, first, n / 64, &off));
                if (off.sum != on.sum || off.sum2 != on.sum2 || off.n != on.n) {
                    printf("MISMATCH direct read-back %s rep %d: %.17g vs %.17g\n", names[h], rep, off.sum, on.sum);
                    ++failures;
                    break;
                }
            }
            CHECK(mc_multi_vanilla_run_f32(m, &v32, seed, first, 4099, &off));   /* a call of a few microseconds */
            CHECK(mc_multi_set_timing(m, 1));
            CHECK(mc_multi_vanilla_run_f32(m, &v32, seed, first, 4099, &on));
            if (off.sum != on.sum || off.sum2 != on.sum2 || off.kernel_ms != 0.0f) {
                printf("MISMATCH direct read-back %s, small call\n", names[h]);
                ++failures;
            }
            printf("ok    read-back forms agree bit for bit on %s (200 direct calls)\n", names[h]);
        }
        /* control variate: the closed-form mean is added back once, after the reduction */
        mc_context_set_control_variate(one, 1);
        CHECK(mc_multi_set_control_variate(m, 1));
        CHECK(mc_basket_run_f64(one, &b64, seed, first, n / 4, &ref));
        CHECK(mc_multi_basket_run_f64(m, &b64, seed, first, n / 4, &got));
        snprintf(label, sizeof label, "basket4 f64 %s control variate", names[h]);
        compare(label, &got, &ref, 1e-10);   /* sums of small differences */
        mc_context_set_control_variate(one, 0);
        CHECK(mc_multi_set_control_variate(m, 0));
        /* fewer paths than devices: the empty shards contribute zeros */
        CHECK(mc_vanilla_run_f64(one, &v64, seed, 7, 2, &ref));
        CHECK(mc_multi_vanilla_run_f64(m, &v64, seed, 7, 2, &got));
        snprintf(label, sizeof label, "vanilla f64 %s, 2 paths", names[h]);
        compare(label, &got, &ref, 1e-12);
        printf("  last RCCL-vs-host relative difference on this handle: %.3g\n", mc_multi_last_reduce_error(m));
    }
    /* launcher threads (round 4): {0,0,0} runs one thread per device; the serial fan-out from the calling thread
     * (MC_MULTI_THREADS=0, rounds 2-3) must give the same bits -- same shards, same kernels, same host sum in device order.
     * 500 short calls back to back and calls after pauses longer than the linger time (parked workers). */
    {
        setenv("MC_MULTI_THREADS", "0", 1);
        mc_multi *serial;
        CHECK(mc_multi_create(d000, R, 0, &serial));
        unsetenv("MC_MULTI_THREADS");
        CHECK(mc_multi_set_reduce(serial, MC_REDUCE_HOST));
        printf("launcher threads: %s default %d, MC_MULTI_THREADS=0 %d, {0} %d\n", rep_name, mc_multi_launcher_threads(m_three),
               mc_multi_launcher_threads(serial), mc_multi_launcher_threads(m_one));
        char cfg[1024];
        CHECK(mc_multi_describe(m_three, cfg, (int)sizeof cfg));
        printf("  %s\n", cfg);
        /* threads exist when the process may keep R + 1 threads running (affinity mask capped by the cgroup quota); on a smaller
         * grant the handle is served serially by design */
        const int th = mc_multi_launcher_threads(m_three);
        failures += !((th == R || (th == 0 && strstr(cfg, "fewer CPUs"))) && mc_multi_launcher_threads(serial) == 0 && mc_multi_launcher_threads(m_one) == 0);
        mc_result a, b;
        int same = 1;
        for (int timing = 0; timing < 2; ++timing) {
            CHECK(mc_multi_set_timing(m_three, timing));
            CHECK(mc_multi_set_timing(serial, timing));
            for (int rep = 0; rep < (timing ? 20 : 500); ++rep) {
                const uint64_t cnt = 1000 + 977 * (uint64_t)rep;
                CHECK(mc_multi_cva_run_f64(m_three, &c64, seed, first + rep, cnt, &a));
                CHECK(mc_multi_cva_run_f64(serial, &c64, seed, first + rep, cnt, &b));
                same = same && a.sum == b.sum && a.sum2 == b.sum2 && a.n == b.n;
                if (rep % 100 == 99) {   /* let the workers park (MC_MULTI_LINGER_US=2000 from the test wrapper), then call again */
                    struct timespec ts = {0, 5000000};
                    nanosleep(&ts, NULL);
                }
            }
        }
        CHECK(mc_multi_basket_run_f32(m_three, &b32, seed, first, n / 4, &a));
        CHECK(mc_multi_basket_run_f32(serial, &b32, seed, first, n / 4, &b));
        same = same && a.sum == b.sum && a.sum2 == b.sum2 && a.n == b.n;
        printf("%s threaded fan-out == serial fan-out bit for bit (520 CVA calls of growing size + a basket call); last fan-out %.1f us threaded, %.1f us serial\n",
               same ? "ok   " : "MISMATCH", mc_multi_last_fanout_us(m_three), mc_multi_last_fanout_us(serial));
        failures += !same;
        /* a launch that fails on a worker thread: the status and the worker's own error text reach the caller */
        mc_option_f64 badopt = v64;
        badopt.s = -1;
        const int rc = mc_multi_vanilla_run_f64(m_three, &badopt, seed, 0, 1000, &a);
        printf("error text from a launcher thread: %s\n", mc_multi_last_error());
        failures += !(rc == MC_ERR_INVALID && strstr(mc_multi_last_error(), "need s>0") != NULL);
        CHECK(mc_multi_vanilla_run_f64(m_three, &v64, seed, 7, 1000, &a));   /* and the handle stays usable */
        CHECK(mc_multi_vanilla_run_f64(serial, &v64, seed, 7, 1000, &b));
        failures += !(a.sum == b.sum);
        mc_multi_destroy(serial);
    }
    /* a 1e9-path fp64 basket call through the multi path (BASELINE configs[3] is n = 16; shape check only here) */
    CHECK(mc_multi_vanilla_run_f32(m_all ? m_all : m_one, &v32, seed, 0, 1000000000ull, &got));
    printf("vanilla f32 1e9 paths over %d device(s): %.6f +- %.6f, kernel %.3f ms, call %.3f ms\n", mc_multi_size(m_all ? m_all : m_one),
           got.expected, got.confidence, got.kernel_ms, got.wall_ms);
    failures += !(fabs(got.expected - 10.386270784322328) < 3.5 / 1.96 * got.confidence);

    /* errors are statuses, not crashes */
    if (mc_multi_vanilla_run_f64(m_one, &v64, seed, 0, 0, &got) == MC_OK) ++failures;
    if (mc_multi_vanilla_run_f64(NULL, &v64, seed, 0, 10, &got) == MC_OK) ++failures;
    const int bad = 99;
    mc_multi *nope = NULL;
    if (mc_multi_create(&bad, 1, 0, &nope) == MC_OK) ++failures;
    printf("error text for a bad device: %s\n", mc_multi_last_error());
    /* a repeated device with the RCCL reduction is refused up front */
    mc_multi *dup;
    const int d00[2] = {0, 0};
    CHECK(mc_multi_create(d00, 2, 0, &dup));
    CHECK(mc_multi_set_reduce(dup, MC_REDUCE_RCCL));
    if (!mock_collective) {
        if (mc_multi_vanilla_run_f64(dup, &v64, seed, 0, 1000, &got) == MC_OK) ++failures;
        printf("error text for a repeated device under RCCL: %s\n", mc_multi_last_error());
    } else {   /* the duplicate check is lifted for the test double: the call goes through */
        if (mc_multi_vanilla_run_f64(dup, &v64, seed, 0, 1000, &got) != MC_OK || got.n != 1000) ++failures;
    }
    mc_multi_destroy(dup);

    mc_multi_destroy(m_one);
    mc_multi_destroy(m_three);
    mc_multi_destroy(m_all);
    mc_context_destroy(one);
    printf("%s (%d failure(s))\n", failures ? "FAILED" : "all checks passed", failures);
    return failures ? 1 : 0;
}
