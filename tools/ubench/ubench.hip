// ubench.hip -- measured VALU issue rates on gfx950 for the instructions the Monte Carlo kernels
// are made of.  Feeds the issue-slot ceiling in DESIGN.md ("Roofline") and bench.py.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
//
// Method: a kernel issues a long unrolled stream of ONE instruction on 8 independent register
// chains (so neither dependency latency nor memory matters), 8 waves per SIMD resident (2048
// workgroups of 256 on 256 CUs).  cycles per wave-instruction per SIMD =
//     shader_clock * time / (instructions per wave * waves per SIMD)
// with shader_clock measured in the same kernel from s_memtime / s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int ITERS = 2000;  // loop trips
constexpr int UNROLL = 8;    // x 8 chains = 64 instructions per trip

// One macro per shape of instruction.  "v" = 32-bit VGPR, "d" = 64-bit VGPR pair.
#define K32_2(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        float r[8];                                                                        \
        for (int i = 0; i < 8; i++) r[i] = 1.0f + threadIdx.x * 1e-3f + i;                 \
        float c = 1.0001f;                                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(r[i]) : "v"(c)); \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        float s = 0;                                                                       \
        for (int i = 0; i < 8; i++) s += r[i];                                             \
        if (s == 123.456f) out[0] = s;                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }

#define K64_2(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        double r[8];                                                                       \
        for (int i = 0; i < 8; i++) r[i] = 1.0 + threadIdx.x * 1e-3 + i;                   \
        double c = 1.0001;                                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(r[i]) : "v"(c)); \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        double s = 0;                                                                      \
        for (int i = 0; i < 8; i++) s += r[i];                                             \
        if (s == 123.456) out[0] = (float)s;                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }

// 32-bit source -> 64-bit destination pair (v_mad_u64_u32, cvt_f64)
#define KMAD64(NAME)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        unsigned long long r[8];                                                           \
        unsigned a[8];                                                                     \
        for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i; r[i] = 0; }    \
        unsigned m = 0xD2511F53u;                                                          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++)                              \
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r[i]) : "v"(a[i]), "s"(m) : "vcc"); \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        unsigned long long s = 0;                                                          \
        for (int i = 0; i < 8; i++) s += r[i];                                             \
        if (s == 123456ull) out[0] = 1.0f;                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }

K32_2(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
K32_2(k_add_f32, "v_add_f32 %0, %0, %1")
K32_2(k_mul_f32, "v_mul_f32 %0, %0, %1")
K32_2(k_max_f32, "v_max_f32 %0, %0, %1")
K32_2(k_xor_b32, "v_xor_b32 %0, %0, %1")
K32_2(k_add_u32, "v_add_u32 %0, %0, %1")
K32_2(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
K32_2(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
K32_2(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
K32_2(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
K32_2(k_exp_f32, "v_exp_f32 %0, %0")
K32_2(k_log_f32, "v_log_f32 %0, %0")
K32_2(k_sin_f32, "v_sin_f32 %0, %0")
K32_2(k_cos_f32, "v_cos_f32 %0, %0")
K32_2(k_sqrt_f32, "v_sqrt_f32 %0, %0")
K32_2(k_rcp_f32, "v_rcp_f32 %0, %0")
K32_2(k_rsq_f32, "v_rsq_f32 %0, %0")
K32_2(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
K32_2(k_alignbit, "v_alignbit_b32 %0, %0, %0, 13")
K32_2(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
K64_2(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
K64_2(k_add_f64, "v_add_f64 %0, %0, %1")
K64_2(k_mul_f64, "v_mul_f64 %0, %0, %1")
K64_2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
K64_2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
K64_2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
K64_2(k_rcp_f64, "v_rcp_f64 %0, %0")
K64_2(k_sqrt_f64, "v_sqrt_f64 %0, %0")
K64_2(k_lshl_b64, "v_lshlrev_b64 %0, 1, %0")
KMAD64(k_mad_u64_u32)

struct Entry {
    const char *name;
    void (*fn)(float *, unsigned long long *);
};

int main(int argc, char **argv)
{
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * blocks_per_cu;
    float *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Entry> ks = {
        {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_max_f32", k_max_f32},
        {"v_xor_b32", k_xor_b32}, {"v_add_u32", k_add_u32}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_alignbit_b32", k_alignbit}, {"v_mov_b32_dpp", k_mov_dpp},
        {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_exp_f32", k_exp_f32}, {"v_log_f32", k_log_f32}, {"v_sin_f32", k_sin_f32}, {"v_cos_f32", k_cos_f32},
        {"v_sqrt_f32", k_sqrt_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_rsq_f32", k_rsq_f32},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32},
        {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64},
        {"v_rcp_f64", k_rcp_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_lshlrev_b64", k_lshl_b64},
    };
    printf("device: %s, %d CUs, %d workgroups of 256 (%d waves/SIMD), %d instr/wave\n", prop.gcnArchName, cus, blocks,
           blocks_per_cu, ITERS * UNROLL * 8);
    printf("%-16s %10s %10s %14s %18s\n", "instruction", "time_ms", "clock_MHz", "cyc/wave-instr", "Glane-ops/s (chip)");
    for (auto &k : ks) {
        for (int rep = 0; rep < 2; ++rep) {  // first rep warms up
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double clock_mhz = (double)h[0] / (double)h[1] * 100.0;
        const double instr_per_wave = (double)ITERS * UNROLL * 8;
        const double waves_per_simd = blocks_per_cu;  // 4 waves per workgroup over 4 SIMDs
        const double cyc = clock_mhz * 1e6 * (ms * 1e-3) / (instr_per_wave * waves_per_simd);
        const double lane_ops = (double)blocks * 256.0 * instr_per_wave / (ms * 1e-3) / 1e9;
        printf("%-16s %10.3f %10.0f %14.2f %18.0f\n", k.name, ms, clock_mhz, cyc, lane_ops);
    }
    return 0;
}
