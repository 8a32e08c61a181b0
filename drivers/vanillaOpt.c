/*
 * vanillaOpt.c -- European vanilla call: closed form, CPU Monte Carlo, GPU Monte Carlo.
 *
 * Plain-C counterpart of the reference driver double_precision/vanillaOpt.cu:28-107:
 * same market data (:22-26), same launch arguments (512 blocks x 128 threads, :12-15), same
 * printed fields (Black-Scholes price; CPU price, CI, time; per GPU run: threads, price, CI,
 * |price - BS|, time, speed-up).  Non-interactive: the path multiplier comes from argv.
 * Links libmcgpu_<prec> (dev_vanillaOpt) and libmchost_<prec> (host_*).
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
    OptionData option = {.s = 100, .k = 100, .r = (mc_real)0.048790, .v = (mc_real)0.2, .t = 1};
    const int sims = mult * SIMPB;

    printf("Vanilla Option Pricing\n\nMonte Carlo scenarios: %d\n", sims);
    printOption(option);
    const double bs = (double)host_bsCall(option);
    printf("\nBlack & Scholes price: %f\n", bs);

    OptionValue cpu = {0, 0};
    double cpu_s = 0;
    if (run_cpu) {
        printf("\nMonte Carlo execution on CPU:\nN^ simulations: %d\n", sims);
        double t0 = now_s();
        cpu = host_vanillaOpt(option, sims);
        cpu_s = now_s() - t0;
    }

    printf("\nMonte Carlo execution on GPU:\n(NumBlocks, NumSimulations): ( %d ; %d )\n", BLOCKS, sims / BLOCKS);
    (void)dev_vanillaOpt(&option, BLOCKS, THREADS, SIMPB); /* creates the device context (reported apart) */
    double t0 = now_s();
    OptionValue gpu = dev_vanillaOpt(&option, BLOCKS, THREADS, sims);
    const double gpu_s = now_s() - t0;

    printf("\n-\tResults:\t-\n");
    if (run_cpu)
        printf("Simulated price for the option with CPU: Expected price, I.C., |price - BS|, time [s]\n%f \n%f \n%f \n%f \n",
               (double)cpu.Expected, (double)cpu.Confidence, fabs((double)cpu.Expected - bs), cpu_s);
    printf("Simulated price for the option with GPU:\n  : NumThreads : Price : Confidence Interval : Difference from BS price :  Time [s] : Speedup :\n");
    printf("%d \n%f \n%f \n%f \n%f \n%.2f \n---\n", THREADS, (double)gpu.Expected, (double)gpu.Confidence,
           fabs((double)gpu.Expected - bs), gpu_s, run_cpu ? cpu_s / gpu_s : 0.0);
    return 0;
}
