// ab_libs.cpp -- two builds of libmc_mi355x.so in ONE process on ONE box, alternating: box-to-box differences of +-3 % (clock state,
// silicon) are larger than most kernel changes, so a variant is only believed when it wins here.
//   hipcc -O2 -Iinclude tools/c/ab_libs.cpp -ldl -o tools/c/ab_libs
//   tools/c/ab_libs <libA.so> <libB.so> [cva64|cva32|van64|bsk64] [paths] [rounds]      (AB_ANTITHETIC=1: the antithetic estimator)
// Each round: 24 back-to-back launches of the workload through A, then through B (stream-event brackets per launch, median of the
// round); prints the per-round medians and the ratio B / A of the medians over all rounds.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mc_mi355x.h"

struct Lib {
    void *h;
    decltype(&mc_context_create) create;
    decltype(&mc_context_destroy) destroy;
    decltype(&mc_context_stream) stream;
    decltype(&mc_cva_launch_f64) cva64;
    decltype(&mc_cva_launch_f32) cva32;
    decltype(&mc_vanilla_launch_f64) van64;
    decltype(&mc_basket_launch_f64) bsk64;
    decltype(&mc_chol_f64) chol;
    decltype(&mc_last_error) err;
    mc_context *ctx;
};
static bool load(const char *path, Lib &l)
{
    l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "%s\n", dlerror()); return false; }
#define SYM(f, n) l.f = (decltype(l.f))dlsym(l.h, n); if (!l.f) { fprintf(stderr, "%s: no %s\n", path, n); return false; }
    SYM(create, "mc_context_create") SYM(destroy, "mc_context_destroy") SYM(stream, "mc_context_stream") SYM(cva64, "mc_cva_launch_f64")
    SYM(cva32, "mc_cva_launch_f32") SYM(van64, "mc_vanilla_launch_f64") SYM(bsk64, "mc_basket_launch_f64") SYM(chol, "mc_chol_f64") SYM(err, "mc_last_error")
    if (l.create(0, 0, &l.ctx) != MC_OK) return false;
    if (getenv("AB_ANTITHETIC")) {
        auto set = (decltype(&mc_context_set_antithetic))dlsym(l.h, "mc_context_set_antithetic");
        if (!set || set(l.ctx, 1) != MC_OK) return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s libA.so libB.so [cva64|cva32|van64|bsk64] [paths] [rounds]\n", argv[0]); return 2; }
    const char *what = argc > 3 ? argv[3] : "cva64";
    const uint64_t n = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1245184ull;
    const int rounds = argc > 5 ? atoi(argv[5]) : 12;
    Lib L[2];
    if (!load(argv[1], L[0]) || !load(argv[2], L[1])) return 1;
    static const mc_cva_f64 c64 = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};
    static const mc_cva_f32 c32 = {0.03f, 0.6f, {100.f, 100.f, 0.05f, 0.2f, 1.f}, 256};
    static const mc_option_f64 v64 = {100., 100., 0.048790, 0.2, 1.};
    static double corr[256], Lm[256], s[16], v[16], d[16], w[16];
    for (int i = 0; i < 16; ++i) {
        s[i] = 100, v[i] = i % 2 ? 0.2 : 0.3, d[i] = 0, w[i] = 1.0 / 16;
        for (int j = 0; j < 16; ++j) corr[16 * i + j] = i == j ? 1.0 : 0.5;
    }
    L[0].chol(16, corr, Lm);
    const mc_basket_f64 b64 = {16, s, v, Lm, d, w, 100., 1., 0.048790164};
    double *triple;
    if (hipMalloc(&triple, 24) != hipSuccess) return 1;
    const int reps = 24;
    std::vector<hipEvent_t> e0(reps), e1(reps);
    for (int i = 0; i < reps; ++i) { (void)hipEventCreate(&e0[i]); (void)hipEventCreate(&e1[i]); }
    auto burst = [&](Lib &l, uint64_t base) -> double {
        hipStream_t st = (hipStream_t)l.stream(l.ctx);
        for (int i = 0; i < reps; ++i) {
            (void)hipEventRecord(e0[i], st);
            int rc;
            if (!strcmp(what, "cva64")) rc = l.cva64(l.ctx, &c64, MC_DEFAULT_SEED, base + (uint64_t)i * n, n, triple, st);
            else if (!strcmp(what, "cva32")) rc = l.cva32(l.ctx, &c32, MC_DEFAULT_SEED, base + (uint64_t)i * n, n, triple, st);
            else if (!strcmp(what, "van64")) rc = l.van64(l.ctx, &v64, MC_DEFAULT_SEED, base + (uint64_t)i * n, n, triple, st);
            else rc = l.bsk64(l.ctx, &b64, MC_DEFAULT_SEED, base + (uint64_t)i * n, n, triple, st);
            if (rc != MC_OK) { fprintf(stderr, "launch failed: %s\n", l.err()); exit(1); }
            (void)hipEventRecord(e1[i], st);
        }
        (void)hipStreamSynchronize(st);
        std::vector<double> us;
        for (int i = 0; i < reps; ++i) { float ms; (void)hipEventElapsedTime(&ms, e0[i], e1[i]); us.push_back(ms * 1e3); }
        std::sort(us.begin(), us.end());
        return us[reps / 2];
    };
    for (int i = 0; i < 6; ++i) { burst(L[0], 1ull << 40); burst(L[1], 1ull << 40); }   // pre-heat both
    std::vector<double> a, b;
    printf("%s, %llu paths, %d rounds of %d launches each, A = %s, B = %s\n", what, (unsigned long long)n, rounds, reps, argv[1], argv[2]);
    for (int r = 0; r < rounds; ++r) {
        a.push_back(burst(L[0], (uint64_t)r << 32));
        b.push_back(burst(L[1], (uint64_t)r << 32));
        printf("  round %2d   A %9.2f us   B %9.2f us   B/A %.4f\n", r, a.back(), b.back(), b.back() / a.back());
    }
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("median of rounds: A %.2f us, B %.2f us, B / A = %.4f\n", a[rounds / 2], b[rounds / 2], b[rounds / 2] / a[rounds / 2]);
    L[0].destroy(L[0].ctx);
    L[1].destroy(L[1].ctx);
    return 0;
}
