/* sanitize_cpu.c -- exercises every CPU code path of the repo under AddressSanitizer + UBSan:
 * the oracle (test infrastructure) and the product's host-side C (host_path.c, both precisions are
 * separate builds; this file is compiled once per precision).  GPU sanitizers are not available on
 * the pool, so this is the memory-safety net for everything that is not a kernel.
 * Built and run by tests/test_sanitizers.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "MonteCarlo.h"
#include "mc_oracle.h"

#ifdef MC_SINGLE_PRECISION
#define ORC(name) name##_f32
#else
#define ORC(name) name##_f64
#endif

int main(void)
{
    orc_result r;
    /* oracle, reference stream */
    ORC(orc_host_vanilla)(100, 100, (mc_real)0.04879, (mc_real)0.2, 1, 20000, 12345u, &r);
    printf("host vanilla %.6f\n", r.expected);
    mc_real corr[N][N], fac[N][N];
    mc_real s[N], v[N], d[N], w[N];
    for (int i = 0; i < N; ++i) {
        s[i] = 100, v[i] = (mc_real)(i % 2 ? 0.2 : 0.3), d[i] = 0, w[i] = (mc_real)1 / N;
        for (int j = 0; j < N; ++j)
            corr[i][j] = (mc_real)(i == j ? 1.0 : 0.5);
    }
    ORC(orc_chol)(N, &corr[0][0], &fac[0][0]);
    ORC(orc_host_basket)(N, s, v, &fac[0][0], d, w, 100, 1, (mc_real)0.05, 5000, 7u, 1, &r);
    printf("host basket %.6f\n", r.expected);
    ORC(orc_host_cva)(100, 100, (mc_real)0.05, (mc_real)0.2, 1, (mc_real)0.03, (mc_real)0.6, 50, 300, 3u, &r);
    printf("host cva %.6f\n", r.expected);
    /* oracle, engine stream (unaligned ranges, per-path output) */
    mc_real *buf = (mc_real *)malloc(sizeof(mc_real) * 1001);
    ORC(orc_dev_vanilla)(100, 100, (mc_real)0.04879, (mc_real)0.2, 1, 99ull, 3, 1001, 0, buf, &r);
    ORC(orc_dev_basket)(N, s, v, &fac[0][0], d, w, 100, 1, (mc_real)0.05, 99ull, (1ull << 32) - 500, 1001, 1, buf, &r);
    ORC(orc_dev_cva)(100, 100, (mc_real)0.05, (mc_real)0.2, 1, (mc_real)0.03, (mc_real)0.6, 37, 99ull, 5, 301, 1, buf, &r);
    printf("dev cva %.6f\n", r.expected);
    free(buf);
    /* product host path (OpenMP CPU twin + closed forms + printers) */
    OptionData o = {100, 100, (mc_real)0.04879, (mc_real)0.2, 1};
    printf("bs %.6f\n", (double)host_bsCall(o));
    OptionValue ov = host_vanillaOpt(o, 200001);
    printf("twin vanilla %.6f %.6f\n", (double)ov.Expected, (double)ov.Confidence);
    MultiOptionData m;
    memcpy(m.s, s, sizeof s), memcpy(m.v, v, sizeof v), memcpy(m.d, d, sizeof d), memcpy(m.w, w, sizeof w);
    Chol(corr, m.p);
    m.k = 100, m.t = 1, m.r = (mc_real)0.05;
    ov = host_basketOpt(&m, 70001);
    printf("twin basket %.6f\n", (double)ov.Expected);
    CVA c = {(mc_real)0.03, (mc_real)0.6, 0, o, 25};
    ov = host_cvaEquityOption(&c, 3001);
    printf("twin cva %.6f\n", (double)ov.Expected);
    printOption(o);
    printMultiOpt(&m);
    return 0;
}
