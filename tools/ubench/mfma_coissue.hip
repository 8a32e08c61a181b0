// mfma_coissue.hip -- do the f64 / f32-input matrix instructions run BESIDE the vector pipe on gfx950, or in it?
// The question behind basket_mfma_f64_kernel: a kernel bound by vector issue gains from moving its
// mat-vec to v_mfma_* only if the matrix instruction leaves the SIMD's vector issue free while it executes.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue [workgroups per CU]
//
// Each kernel runs a loop of M matrix instructions on 4 independent accumulators interleaved with V independent vector
// fmas (8 chains); cycles per loop trip per SIMD = shader clock x time / (trips x waves per SIMD).  If the pipes overlap,
// trip(M, V) ~ max(M x t_mfma, V x t_valu); if they share, ~ the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TRIPS = 4000;

template <int M, int V>
__global__ __launch_bounds__(256) void k_f64(float *out, unsigned long long *clk)
{
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){1.0 + i, 2.0, 3.0, 4.0 + threadIdx.x};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.999, r[8];
    for (int i = 0; i < 8; ++i) r[i] = 1.0 + i + threadIdx.x * 1e-3;
    const double c = 1.0001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < TRIPS; ++it) {
#pragma unroll
        for (int u = 0; u < (M > V ? M : V); ++u) {
            if (u < M) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
            if (u < V) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[u & 7]) : "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 123.456) out[0] = (float)s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int M, int V>
__global__ __launch_bounds__(256) void k_f32(float *out, unsigned long long *clk)
{
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){1.0f + i, 2.0f, 3.0f, 4.0f + threadIdx.x};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.999f;
    double r[8];   // register pairs for v_pk_fma_f32
    for (int i = 0; i < 8; ++i) r[i] = 1.0 + i + threadIdx.x * 1e-3;
    const double c = 1.0001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < TRIPS; ++it) {
#pragma unroll
        for (int u = 0; u < (M > V ? M : V); ++u) {
            if (u < M) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
            if (u < V) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[u & 7]) : "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += (float)r[i];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

struct Entry { const char *name; void (*fn)(float *, unsigned long long *); int m, v; };

int main(int argc, char **argv)
{
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2;   // 256-thread workgroups per CU = waves per SIMD
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * per_cu;
    float *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Entry ks[] = {
        {"f64: 8 mfma_f64_16x16x4", k_f64<8, 0>, 8, 0},
        {"f64: 16 v_fma_f64", k_f64<0, 16>, 0, 16},
        {"f64: 8 mfma + 16 v_fma_f64", k_f64<8, 16>, 8, 16},
        {"f64: 8 mfma + 64 v_fma_f64", k_f64<8, 64>, 8, 64},
        {"f64: 4 mfma + 64 v_fma_f64", k_f64<4, 64>, 4, 64},
        {"f64: 64 v_fma_f64", k_f64<0, 64>, 0, 64},
        {"f32: 8 mfma_f32_16x16x4", k_f32<8, 0>, 8, 0},
        {"f32: 16 v_pk_fma_f32", k_f32<0, 16>, 0, 16},
        {"f32: 8 mfma + 16 v_pk_fma_f32", k_f32<8, 16>, 8, 16},
        {"f32: 8 mfma + 64 v_pk_fma_f32", k_f32<8, 64>, 8, 64},
        {"f32: 4 mfma + 64 v_pk_fma_f32", k_f32<4, 64>, 4, 64},
        {"f32: 64 v_pk_fma_f32", k_f32<0, 64>, 0, 64},
    };
    printf("device: %s, %d CUs, %d workgroups of 256 (%d waves/SIMD), %d trips\n", prop.gcnArchName, prop.multiProcessorCount, blocks, per_cu, TRIPS);
    printf("%-34s %9s %9s %14s %22s\n", "loop body", "time_ms", "clock_MHz", "cyc/trip/SIMD", "cyc per mfma beyond valu");
    double valu_cost[2] = {0, 0};
    for (const Entry &k : ks) {
        float best = 1e30f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, clk);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); }
        }
        const double mhz = (double)h[0] / ((double)h[1] / 100.0);   // s_memrealtime ticks at 100 MHz
        const double cyc = mhz * 1e3 * best / ((double)TRIPS * per_cu);
        const int is32 = k.name[1] == '3';
        if (k.m == 0 && k.v == 16) valu_cost[is32] = cyc / 16;
        char extra[64] = "";
        if (k.m && valu_cost[is32] > 0) snprintf(extra, sizeof extra, "%.1f", (cyc - k.v * valu_cost[is32]) / k.m);
        printf("%-34s %9.3f %9.0f %14.1f %22s\n", k.name, best, mhz, cyc, extra);
    }
    return 0;
}
