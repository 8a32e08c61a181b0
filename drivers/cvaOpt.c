/*
 * cvaOpt.c -- CVA of one European call over several exposure grids: CPU vs GPU Monte Carlo.
 *
 * Plain-C counterpart of the reference driver double_precision/cvaOpt.cu:30-111:
 * same market data (S=K=100, r=5%, v=20%, T=1, default intensity 3%, recovery 40% -> LGD 0.6,
 * :22-34), 1024 blocks, grids {25, 50, 75, 250, 500} x thread counts {128, 256, 512, 1024}
 * (:70-108; the thread count does not change this engine's result, it is looped for output
 * parity), 131072 paths x multiplier.  Unlike the reference (whose CPU CVA call is commented
 * out, :80-91, so its speed-up prints 0/x) the CPU leg actually runs unless --no-cpu is given.
 */
#include "driver_util.h"
#include "MonteCarlo.h"

#define BLOCKS 1024
#define SIMPB 131072

int main(int argc, char **argv)
{
    int mult, run_cpu;
    if (!parse_args(argc, argv, &mult, &run_cpu))
        return 1;
    if (argc < 2)
        mult = 1; /* the reference runs exactly 131072 paths */
    const int grids[5] = {25, 50, 75, 250, 500};
    const int threads[4] = {128, 256, 512, 1024};
    const int sims = mult * SIMPB;
    const mc_real recovery = (mc_real)0.4;
    CVA cva;
    cva.defInt = (mc_real)0.03;
    cva.lgd = 1 - recovery;
    cva.ns = 0;
    cva.option = (OptionData){.s = 100, .k = 100, .r = (mc_real)0.05, .v = (mc_real)0.2, .t = 1};

    printf("CVA of a European call option\n\nMonte Carlo scenarios: %d\n", sims);
    printOption(cva.option);
    printf("Default intensity: %.2f %%   Loss given default: %.2f %%\n", (double)cva.defInt * 100, (double)cva.lgd * 100);
    cva.n = grids[0];
    (void)dev_cvaEquityOption(&cva, BLOCKS, threads[0], SIMPB); /* creates the device context */

    for (int g = 0; g < 5; ++g) {
        cva.n = grids[g];
        printf("\n--- exposure dates: %d ---\n", cva.n);
        OptionValue cpu = {0, 0};
        double cpu_s = 0;
        if (run_cpu) {
            double t0 = now_s();
            cpu = host_cvaEquityOption(&cva, sims);
            cpu_s = now_s() - t0;
            printf("CPU: CVA %f  I.C. %f  time [s] %f\n", (double)cpu.Expected, (double)cpu.Confidence, cpu_s);
        }
        printf("GPU:  : NumThreads : CVA : Confidence Interval : Difference from CPU :  Time [s] : Speedup :\n");
        for (int t = 0; t < 4; ++t) {
            double t0 = now_s();
            OptionValue gpu = dev_cvaEquityOption(&cva, BLOCKS, threads[t], sims);
            const double gpu_s = now_s() - t0;
            printf("%d \n%f \n%f \n%f \n%f \n%.2f \n---\n", threads[t], (double)gpu.Expected, (double)gpu.Confidence,
                   run_cpu ? fabs((double)gpu.Expected - (double)cpu.Expected) : 0.0, gpu_s, run_cpu ? cpu_s / gpu_s : 0.0);
        }
    }
    return 0;
}
