/*
 * basketOpt.c -- European basket call on N correlated assets: CPU vs GPU Monte Carlo.
 *
 * Plain-C counterpart of the reference driver double_precision/basketOpt.cu:27-144.
 * N = 3 uses the reference's data verbatim (vols 0.2/0.3/0.2, all correlations -0.5, :34-61 -- a
 * singular matrix: Chol leaves the last column zero).  For N != 3 the reference's generator
 * (:160-177, +-0.5 by column parity) is NOT positive definite; this driver uses
 * vols alternating 0.3/0.2 (:147-158) with equicorrelation 0.5 instead and says so.
 * The correlation matrix is factorised on the host and stored back into option.p before either
 * path runs, exactly as the reference does (:96-99).
 */
#include "driver_util.h"
#include "MonteCarlo.h"

#define THREADS 128
#define BLOCKS 512
#define SIMPB 131072

int main(int argc, char **argv)
{
    int mult, run_cpu;
    if (!parse_args(argc, argv, &mult, &run_cpu))
        return 1;
    MultiOptionData option;
    const mc_real weight = (mc_real)1 / N;
    for (int i = 0; i < N; ++i) {
        option.s[i] = 100;
        option.w[i] = weight;
        option.d[i] = 0;
        if (N == 3)
            option.v[i] = (mc_real)(i == 1 ? 0.3 : 0.2);
        else
            option.v[i] = (mc_real)(i % 2 == 0 ? 0.3 : 0.2);
        for (int j = 0; j < N; ++j)
            option.p[i][j] = (mc_real)(i == j ? 1.0 : (N == 3 ? -0.5 : 0.5));
    }
    option.k = 100.f;
    option.r = (mc_real)0.048790164;
    option.t = 1.f;
    const int sims = mult * SIMPB;

    printf("Basket Option Pricing\n\nMonte Carlo scenarios: %d\n", sims);
    if (N != 3)
        printf("(N = %d: equicorrelation 0.5 -- the reference's +-0.5 pattern is not positive definite)\n", N);
    if (N < 7)
        printMultiOpt(&option);
    else
        printf("\nBasket Option with %d underlyings\n", N);

    mc_real factor[N][N];
    Chol(option.p, factor);
    int zero_pivots = 0;
    for (int i = 0; i < N; ++i) {
        zero_pivots += factor[i][i] == 0;
        for (int j = 0; j < N; ++j)
            option.p[i][j] = factor[i][j];
    }
    if (zero_pivots)
        printf("note: %d zero pivot(s) in the Cholesky factor (input not positive definite)\n", zero_pivots);

    OptionValue cpu = {0, 0};
    double cpu_s = 0;
    if (run_cpu) {
        printf("\nMonte Carlo execution on CPU...\n");
        double t0 = now_s();
        cpu = host_basketOpt(&option, sims);
        cpu_s = now_s() - t0;
    }
    printf("\nMonte Carlo execution on GPU...\nMonte Carlo for (%d,%d) x %d simulations per thread\n", BLOCKS, THREADS,
           sims / BLOCKS / THREADS);
    (void)dev_basketOpt(&option, BLOCKS, THREADS, SIMPB);
    double t0 = now_s();
    OptionValue gpu = dev_basketOpt(&option, BLOCKS, THREADS, sims);
    const double gpu_s = now_s() - t0;

    printf("\n-\tResults:\t-\n");
    if (run_cpu)
        printf("Simulated price for the option with CPU: Expected price, I.C., time [s]\n%f \n%f \n%f \n", (double)cpu.Expected,
               (double)cpu.Confidence, cpu_s);
    printf("Simulated price for the option with GPU:\n  : NumThreads : Price : Confidence Interval : Difference from CPU price :  Time [s] : Speedup :\n");
    printf("%d \n%f \n%f \n%f \n%f \n%.2f \n---\n", THREADS, (double)gpu.Expected, (double)gpu.Confidence,
           run_cpu ? fabs((double)gpu.Expected - (double)cpu.Expected) : 0.0, gpu_s, run_cpu ? cpu_s / gpu_s : 0.0);
    return 0;
}
