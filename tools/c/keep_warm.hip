// keep_warm.hip -- VERDICT r04 "next" #4: can a resident, tiny, low-priority kernel hold the shader clock through the gaps between
// one-off pricing calls, and what does it cost?  (C5's shard of 8 takes 0.98 ms back to back and 1.17 ms when calls are >= 5 ms
// apart: profiles/r04_shard_clock_quantisation_vs_dvfs.log.)
//
// One context on device 0; the CVA kernel of BASELINE configs[4] (256 dates, fp64), 1.25e6 paths = C5's shard of 8, as synchronous
// calls (mc_cva_run_f64, kernel_ms from events) separated by host sleeps.  For every keep-warm MODE the same gap sweep:
//     off          nothing resident (round 4's section B)
//     sleep1       ONE wave (64 lanes) on a lowest-priority stream that s_sleeps between looks at the clock and the stop word
//     busy1        ONE wave that issues fp32 FMAs back to back
//     busy8        8 workgroups of one wave (one per XCD by the dispatcher's round-robin), busy
//     busy256      256 workgroups of one wave (one per CU), busy
//     busy1024     256 workgroups of four waves (one wave on every SIMD of the chip), busy, co-resident with the pricing kernel
//     full_yield   2048 workgroups of four waves, busy -- a fully loaded GPU between the calls; the host stops it (stop word, drain)
//                  right before each pricing call and starts it again right after: "pre-heat all the time"
// The resident kernel leaves when the host sets a stop word in pinned memory OR after its time limit (every wave reaches both
// tests each iteration: the grid always drains).  Board power and clocks: `rocm-smi` run as a child process in the middle of
// a 1.5 s idle stretch of each mode.
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/c/keep_warm.hip -Lmontecarlocuda_amd/csrc -lmc_mi355x
//         -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -o /tmp/keep_warm
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
#define HIP(call)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                      \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

static const mc_cva_f64 CVA = {0.03, 0.6, {100., 100., 0.05, 0.2, 1.}, 256};

// wall_clock64(): the constant 100 MHz counter.  `ticks` bounds the stay whatever the host does.
__global__ void keep_warm_kernel(const volatile int *stop, long long ticks, int busy, float *sink)
{
    const long long t0 = wall_clock64();
    float x = (float)threadIdx.x;
    for (;;) {
        if (wall_clock64() - t0 >= ticks)
            break;
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
            break;
        if (busy == 2) {        // eight independent chains per lane: the vector pipe is kept full (a high-power filler)
            float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                a0 = __builtin_fmaf(a0, 1.0000001f, 0.5f), a1 = __builtin_fmaf(a1, 1.0000001f, 0.5f);
                a2 = __builtin_fmaf(a2, 1.0000001f, 0.5f), a3 = __builtin_fmaf(a3, 1.0000001f, 0.5f);
                a4 = __builtin_fmaf(a4, 1.0000001f, 0.5f), a5 = __builtin_fmaf(a5, 1.0000001f, 0.5f);
                a6 = __builtin_fmaf(a6, 1.0000001f, 0.5f), a7 = __builtin_fmaf(a7, 1.0000001f, 0.5f);
            }
            x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        } else if (busy) {
#pragma unroll
            for (int i = 0; i < 256; ++i)
                x = __builtin_fmaf(x, 1.0000001f, 0.5f);
        } else {
            __builtin_amdgcn_s_sleep(127);
        }
    }
    if (x == 12345.678f)
        *sink = x;
}

// A stand-in for the pricing load that can be told to leave: fp64 FMAs, 32 x 32 -> 64 multiplies and fp32 transcendentals in about
// the kernels' proportions, on every SIMD, until `ticks` of the 100 MHz clock have passed OR a word in DEVICE memory no longer holds
// the value the launcher saw (the owner's stream writes it right before its own kernel: hipStreamWriteValue32).
__global__ __launch_bounds__(256) void pulse_kernel(const uint32_t *word, uint32_t seen, long long ticks, double *sink)
{
    const long long t0 = wall_clock64();
    double a = 1.0 + threadIdx.x * 1e-6, b = 0.5, c = 0.25, d = 2.0;
    uint64_t x = 0x9E3779B97F4A7C15ull + threadIdx.x + blockIdx.x * 256u;
    float f = 1.0f + threadIdx.x * 1e-3f;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a = __builtin_fma(a, 0.9999999, b);
            b = __builtin_fma(b, 1.0000001, -c);
            c = __builtin_fma(c, 0.9999998, d * 1e-9);
            d = __builtin_fma(d, 1.0000002, -a * 1e-9);
            x = (uint64_t)(uint32_t)x * 0xD2511F53ull + (x >> 32);
            x = (uint64_t)(uint32_t)x * 0xCD9E8D57ull + (x >> 32);
            f = __builtin_amdgcn_exp2f(f * 0.001f) + __builtin_amdgcn_logf(f + 2.0f);
        }
        if (wall_clock64() - t0 >= ticks)
            break;
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seen)
            break;
    }
    if (a + b + c + d + (double)x + f == 12345.678)
        *sink = a;
}

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

static void smi(const char *tag)
{
    FILE *p = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk|fclk' | tr -s ' ' | tr '\\n' ';'", "r");
    char buf[2048] = "";
    if (p) {
        size_t n = fread(buf, 1, sizeof buf - 1, p);
        buf[n] = 0;
        pclose(p);
    }
    printf("   rocm-smi (%s): %s\n", tag, buf[0] ? buf : "no output");
}

int main(int argc, char **argv)
{
    mc_context *c;
    MC(mc_context_create(0, 0, &c));
    double *d_triple = nullptr;
    float *d_sink = nullptr;
    HIP(hipMalloc(&d_triple, 3 * sizeof(double)));
    HIP(hipMalloc(&d_sink, sizeof(float)));
    int *stop = nullptr;
    HIP(hipHostMalloc(&stop, sizeof(int), hipHostMallocDefault));
    int lo = 0, hi = 0;
    HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t side;
    HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));   // `lo` = the numerically greatest = the LOWEST priority
    printf("stream priorities: lowest %d, highest %d; the resident kernel runs on a stream of priority %d\n", lo, hi, lo);
    mc_context_set_timing(c, 1);

    struct Mode { const char *name; int blocks, busy, lanes, yield; };
    const bool second = argc > 1 && !strcmp(argv[1], "heavy");
    const Mode first_set[] = {{"off", 0, 0, 64, 0}, {"sleep1", 1, 0, 64, 0}, {"busy1", 1, 1, 64, 0}, {"busy8", 8, 1, 64, 0}, {"busy256", 256, 1, 64, 0}, {"off", 0, 0, 64, 0}};
    const Mode heavy_set[] = {{"off", 0, 0, 64, 0}, {"busy1024", 256, 1, 256, 0}, {"full_yield", 2048, 1, 256, 1}, {"off", 0, 0, 64, 0}};
    // third set: a filler that keeps the vector pipes FULL between the calls (8 independent FMA chains per lane, 8 waves per SIMD)
    const Mode full_set[] = {{"off", 0, 0, 64, 0}, {"valu_full_yield", 4096, 2, 256, 1}, {"off", 0, 0, 64, 0}};
    const bool third = argc > 1 && !strcmp(argv[1], "full");
    // fourth set ("pulse"): no resident kernel at all -- during the gap the host launches a SHORT burst of the pricing kernel itself
    // (65 536 paths = one wave-trip on every SIMD, ~50 us at full intensity) every `period` us: does a few per cent of duty hold the state?
    // fifth set ("pulse2"): the same pulsing with the custom kernel above instead of the CVA kernel -- does a stand-in hold the state, and
    // does the owner's stream-ordered write make it leave?  Bursts of `burst_us` every `period_us`; the real call first writes the word.
    if (argc > 1 && !strcmp(argv[1], "pulse2")) {
        const uint64_t n = 1250000ull;
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        uint32_t *d_word = nullptr;
        double *d_sink2 = nullptr;
        HIP(hipMalloc(&d_word, 4));
        HIP(hipMalloc(&d_sink2, 8));
        HIP(hipMemset(d_word, 0, 4));
        hipStream_t own = (hipStream_t)mc_context_stream(c);
        uint32_t gen = 0;
        printf("%-26s %8s %12s %10s %10s %10s %12s\n", "mode", "gap ms", "kernel med", "min", "max", "pulses/gap", "wall med us");
        struct P { int period_us, burst_us; };
        for (P p : {P{0, 0}, P{500, 200}, P{250, 200}, P{500, 400}, P{1000, 400}, P{0, 0}}) {
            if (preheat(c, d_triple, 300)) return 1;
            for (double gap : {0.0, 5.0, 20.0, 100.0}) {
                std::vector<double> k, w;
                double pulses = 0;
                for (int i = 0; i < 10; ++i) {
                    const double g0 = now_ms();
                    int np = 0;
                    while (now_ms() - g0 < gap) {
                        if (p.period_us) {
                            hipLaunchKernelGGL(pulse_kernel, dim3(2048), dim3(256), 0, side, d_word, gen, (long long)p.burst_us * 100, d_sink2);
                            ++np;
                            usleep((useconds_t)p.period_us);
                        } else {
                            usleep((useconds_t)(gap * 1e3));
                        }
                    }
                    pulses += np;
                    // the real call: a pulse may be in flight (it was launched up to `period` ago and lasts `burst`): tell it to leave
                    const double tc = now_ms();
                    ++gen;
                    HIP(hipStreamWriteValue32(own, d_word, gen, 0));
                    mc_result r;
                    MC(mc_cva_run_f64(c, &CVA, MC_DEFAULT_SEED, (uint64_t)i * n, n, &r));
                    w.push_back((now_ms() - tc) * 1e3);
                    k.push_back(r.kernel_ms * 1e3);
                }
                char name[64];
                snprintf(name, sizeof name, p.period_us ? "custom burst %d / %d us" : "off", p.burst_us, p.period_us);
                printf("%-26s %8.1f %12.1f %10.1f %10.1f %10.1f %12.1f\n", name, gap, med(k), *std::min_element(k.begin(), k.end()),
                       *std::max_element(k.begin(), k.end()), pulses / 10, med(w));
                fflush(stdout);
            }
            if (p.period_us) {     // board power while pulsing
                const double t0p = now_ms();
                bool told = false;
                while (now_ms() - t0p < 1500) {
                    hipLaunchKernelGGL(pulse_kernel, dim3(2048), dim3(256), 0, side, d_word, gen, (long long)p.burst_us * 100, d_sink2);
                    usleep((useconds_t)p.period_us);
                    if (!told && now_ms() - t0p > 750) {
                        char tag[64];
                        snprintf(tag, sizeof tag, "custom burst %d / %d us", p.burst_us, p.period_us);
                        smi(tag);
                        told = true;
                    }
                }
                HIP(hipStreamSynchronize(side));
            }
        }
        hipStreamDestroy(side);
        mc_context_destroy(c);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "pulse")) {
        const uint64_t n = 1250000ull;
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%-22s %8s %12s %10s %10s %10s\n", "mode", "gap ms", "kernel med", "min", "max", "pulses/gap");
        for (int period_us : {0, 2000, 1000, 500, 250, 0}) {
            for (uint64_t pulse_paths : {65536ull, 262144ull}) {
                if (period_us == 0 && pulse_paths != 65536ull) continue;
                if (preheat(c, d_triple, 300)) return 1;
                for (double gap : {0.0, 5.0, 20.0, 100.0}) {
                    std::vector<double> k;
                    double pulses = 0;
                    for (int i = 0; i < 10; ++i) {
                        const double g0 = now_ms();
                        int np = 0;
                        while (now_ms() - g0 < gap) {
                            if (period_us) {
                                MC(mc_cva_launch_f64(c, &CVA, MC_DEFAULT_SEED, (1ull << 41) + (uint64_t)np * pulse_paths, pulse_paths, d_triple, mc_context_stream(c)));
                                ++np;
                                usleep((useconds_t)period_us);
                            } else {
                                usleep((useconds_t)(gap * 1e3));
                            }
                        }
                        pulses += np;
                        mc_result r;
                        MC(mc_cva_run_f64(c, &CVA, MC_DEFAULT_SEED, (uint64_t)i * n, n, &r));
                        k.push_back(r.kernel_ms * 1e3);
                    }
                    char name[64];
                    snprintf(name, sizeof name, period_us ? "pulse %llu / %d us" : "off", (unsigned long long)pulse_paths, period_us);
                    printf("%-22s %8.1f %12.1f %10.1f %10.1f %10.1f\n", name, gap, med(k), *std::min_element(k.begin(), k.end()),
                           *std::max_element(k.begin(), k.end()), pulses / 10);
                    fflush(stdout);
                }
                usleep(300000);
                smi(period_us ? "after a pulsed sweep (idle now)" : "off");
            }
        }
        // what the pulses cost: board power in the middle of 1.5 s of pulsing (rocm-smi as a child process), and of the kernel back to back
        for (int period_us : {0, 1000, 500, 250, -1}) {
            const double t0p = now_ms();
            bool told = false;
            int np = 0;
            while (now_ms() - t0p < 1500) {
                if (period_us != 0) {
                    MC(mc_cva_launch_f64(c, &CVA, MC_DEFAULT_SEED, (1ull << 42) + (uint64_t)np * 262144ull, period_us < 0 ? 10000000ull : 262144ull, d_triple,
                                         mc_context_stream(c)));
                    ++np;
                    if (period_us > 0) usleep((useconds_t)period_us);
                    else HIP(hipStreamSynchronize((hipStream_t)mc_context_stream(c)));
                } else {
                    usleep(1000);
                }
                if (!told && now_ms() - t0p > 750) {
                    char tag[96];
                    snprintf(tag, sizeof tag, period_us < 0 ? "1e7-path launches back to back (full load)" : period_us ? "262144-path pulse every %d us sleep" : "idle", period_us);
                    smi(tag);
                    told = true;
                }
            }
            HIP(hipStreamSynchronize((hipStream_t)mc_context_stream(c)));
            printf("   ... %d launches in 1.5 s\n", np);
        }
        hipStreamDestroy(side);
        hipFree(d_triple), hipFree(d_sink), hipHostFree(stop);
        mc_context_destroy(c);
        return 0;
    }
    const std::vector<Mode> modes = third ? std::vector<Mode>(full_set, full_set + 3)
                                  : second ? std::vector<Mode>(heavy_set, heavy_set + 4) : std::vector<Mode>(first_set, first_set + 6);
    const uint64_t n = 1250000ull;
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-15s %8s %12s %10s %10s %12s\n", "mode", "gap ms", "kernel med", "min", "max", "wall med us");
    for (const Mode &m : modes) {
        if (preheat(c, d_triple, 300)) return 1;
        *stop = 0;
        auto start_resident = [&]() {
            *stop = 0;
            if (m.blocks)
                hipLaunchKernelGGL(keep_warm_kernel, dim3(m.blocks), dim3(m.lanes), 0, side, stop, 60ll * 100000000ll /* 60 s at most */, m.busy, d_sink);
            return hipGetLastError();
        };
        HIP(start_resident());
        for (double gap : {0.0, 1.0, 5.0, 20.0, 100.0, 500.0}) {
            std::vector<double> k, w;
            const int reps = gap >= 100 ? 6 : 12;
            for (int i = 0; i < reps; ++i) {
                if (gap > 0) usleep((useconds_t)(gap * 1e3));
                mc_result r;
                const double t_call = now_ms();
                if (m.yield) {      // the heavy filler leaves before the call ...
                    *stop = 1;
                    HIP(hipStreamSynchronize(side));
                }
                MC(mc_cva_run_f64(c, &CVA, MC_DEFAULT_SEED, (uint64_t)i * n, n, &r));
                const double wall_us = (now_ms() - t_call) * 1e3;
                if (m.yield)        // ... and comes back after it
                    HIP(start_resident());
                k.push_back(r.kernel_ms * 1e3), w.push_back(m.yield ? wall_us : r.wall_ms * 1e3);
            }
            printf("%-15s %8.1f %12.1f %10.1f %10.1f %12.1f\n", m.name, gap, med(k), *std::min_element(k.begin(), k.end()),
                   *std::max_element(k.begin(), k.end()), med(w));
            fflush(stdout);
        }
        usleep(750000);
        smi(m.name);
        usleep(750000);
        // the long launch next to the resident kernel: what it costs the real work (10 launches of 1e7 paths back to back)
        {
            std::vector<double> k;
            if (m.yield) {
                *stop = 1;
                HIP(hipStreamSynchronize(side));
            }
            for (int i = 0; i < 6; ++i) {
                mc_result r;
                MC(mc_cva_run_f64(c, &CVA, MC_DEFAULT_SEED, (uint64_t)i * 10000000ull, 10000000ull, &r));
                k.push_back(r.kernel_ms * 1e3);
            }
            printf("%-15s  1e7 paths back to back: kernel med %.1f us (min %.1f)\n", m.name, med(k), *std::min_element(k.begin(), k.end()));
        }
        *stop = 1;
        HIP(hipStreamSynchronize(side));
    }
    hipStreamDestroy(side);
    hipFree(d_triple), hipFree(d_sink), hipHostFree(stop);
    mc_context_destroy(c);
    return 0;
}
