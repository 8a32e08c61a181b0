/* driver_util.h -- shared by the three demo drivers: wall-clock timer and argument parsing.
 * The reference drivers time with cudaEvent pairs (vanillaOpt.cu:58-83) and read one integer with
 * scanf (vanillaOpt.cu:51-53); these read it from argv (default 8) and use clock_gettime. */
#ifndef DRIVER_UTIL_H_
#define DRIVER_UTIL_H_
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* <prog> [multiplier] [--no-cpu]   : simulations = multiplier x 131072, as in the reference */
__attribute__((unused)) static int parse_args(int argc, char **argv, int *multiplier, int *run_cpu)
{
    *multiplier = 8;
    *run_cpu = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--no-cpu"))
            *run_cpu = 0;
        else if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
            printf("usage: %s [multiplier] [--no-cpu]\n  simulations = multiplier x 131072 (default 8)\n", argv[0]);
            return 0;
        } else
            *multiplier = atoi(argv[i]);
    }
    if (*multiplier < 1 || *multiplier > 16383) {
        fprintf(stderr, "multiplier must be in [1, 16383] (sims is a 32-bit int, as in the reference)\n");
        return 0;
    }
    return 1;
}
#endif
