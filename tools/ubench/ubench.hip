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
// the non-arithmetic fp64 helpers of mc_math_f64.hpp's table forms (round 3): scaling, exponent / mantissa extraction,
// conversions, the 1/sqrt seed, max, 64-bit move, and the lane <-> scalar moves of an SGPR spill
K64_2(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
K64_2(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
K64_2(k_rsq_f64, "v_rsq_f64 %0, %0")
K64_2(k_max_f64, "v_max_f64 %0, %0, %1")
K64_2(k_mov_b64, "v_mov_b64 %0, %1")
K64_2(k_fmac_f64, "v_fmac_f64 %0, %1, %1")
#define K64_FROM32(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        double r[8];                                                                       \
        int a[8];                                                                          \
        for (int i = 0; i < 8; i++) { r[i] = 0; a[i] = threadIdx.x + i; }                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(r[i]) : "v"(a[i])); \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        double s = 0;                                                                      \
        for (int i = 0; i < 8; i++) s += r[i];                                             \
        if (s == 123.456) out[0] = (float)s;                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }
K64_FROM32(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
#define K32_FROM64(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        double r[8];                                                                       \
        int a[8];                                                                          \
        for (int i = 0; i < 8; i++) { r[i] = 1.0 + threadIdx.x * 1e-3 + i; a[i] = 0; }     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(a[i]) : "v"(r[i])); \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        int s = 0;                                                                         \
        for (int i = 0; i < 8; i++) s += a[i];                                             \
        if (s == 123456) out[0] = 1.0f;                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }
K32_FROM64(k_frexp_exp_f64, "v_frexp_exp_i32_f64 %0, %1")
K32_FROM64(k_cmp_f64, "v_cmp_lt_f64 vcc, %1, %1\n v_mov_b32 %0, 0")
__global__ __launch_bounds__(256) void k_readlane(float *out, unsigned long long *clk)
{
    int r[8];
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x + i;
    int sacc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
            _Pragma("unroll") for (int i = 0; i < 8; i++) {
                int s;
                asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(r[i]));
                sacc ^= s;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    if (sacc == 123456) out[0] = 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// ---- context probes: operand kinds, dependent chains, Philox-like mixes ---------------------
#define K32_S(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        float r[8];                                                                        \
        for (int i = 0; i < 8; i++) r[i] = 1.0f + threadIdx.x * 1e-3f + i;                 \
        float c = __builtin_amdgcn_readfirstlane(blockIdx.x) * 1e-9f + 1.0001f;            \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(r[i]) : "s"(c)); \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        float s = 0;                                                                       \
        for (int i = 0; i < 8; i++) s += r[i];                                             \
        if (s == 123.456f) out[0] = s;                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }
// one register, fully dependent chain (64 instr per trip on r[0])
#define K32_DEP(NAME, ASM)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        float r = 1.0f + threadIdx.x * 1e-3f;                                              \
        float c = 1.0001f;                                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL * 8; ++u) asm volatile(ASM : "+v"(r) : "v"(c)); \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        if (r == 123.456f) out[0] = r;                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }
K32_S(k_xor_sgpr, "v_xor_b32 %0, %1, %0")
K32_S(k_add_f32_sgpr, "v_add_f32 %0, %1, %0")
K32_S(k_sub_clamp, "v_sub_f32_e64 %0, %0, %1 clamp")
K32_2(k_sub_f32, "v_sub_f32 %0, %0, %1")
K32_2(k_and_b32, "v_and_b32 %0, %0, %1")
K32_2(k_or_b32, "v_or_b32 %0, %0, %1")
K32_2(k_lshl_b32, "v_lshlrev_b32 %0, 1, %0")
K32_2(k_lshr_b32, "v_lshrrev_b32 %0, 1, %0")
K32_2(k_mov_b32, "v_mov_b32 %0, %1")
K32_2(k_fmac_f32, "v_fmac_f32 %0, %1, %1")
K32_2(k_min_f32, "v_min_f32 %0, %0, %1")
K32_2(k_max_i32, "v_max_i32 %0, %0, %1")
K32_2(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
K32_2(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
K32_2(k_xor3, "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1")
K32_DEP(k_xor_dep, "v_xor_b32 %0, %0, %1")
K32_DEP(k_add_dep, "v_add_f32 %0, %0, %1")
K32_DEP(k_fma_dep, "v_fma_f32 %0, %0, %1, %1")
// Philox-shaped mix: per chain {mad_u64_u32; xor; xor} dependent, 8 chains
__global__ __launch_bounds__(256) void k_philox_mix(float *out, unsigned long long *clk)
{
    unsigned a[8];
    unsigned long long p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i; p[i] = 0; }
    unsigned m = 0xD2511F53u, key = __builtin_amdgcn_readfirstlane(blockIdx.x) + 12345u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
            _Pragma("unroll") for (int i = 0; i < 8; i++) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "v"(a[i]), "s"(m) : "vcc");
                unsigned hi = (unsigned)(p[i] >> 32), lo = (unsigned)p[i];
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(hi), "v"(lo));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(key));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 123456u) out[0] = 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// same instruction multiset as k_philox_mix, but each dependent instruction is 8 instructions away
__global__ __launch_bounds__(256) void k_philox_mix_far(float *out, unsigned long long *clk)
{
    unsigned a[8];
    unsigned long long p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i; p[i] = 0; }
    unsigned m = 0xD2511F53u, key = __builtin_amdgcn_readfirstlane(blockIdx.x) + 12345u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
            _Pragma("unroll") for (int i = 0; i < 8; i++)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "v"(a[i]), "s"(m) : "vcc");
            _Pragma("unroll") for (int i = 0; i < 8; i++) {
                unsigned hi = (unsigned)(p[i] >> 32), lo = (unsigned)p[i];
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(hi), "v"(lo));
            }
            _Pragma("unroll") for (int i = 0; i < 8; i++)
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(key));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 123456u) out[0] = 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
// two distinct fresh VGPR sources per xor (8 chains, operands rotate)
__global__ __launch_bounds__(256) void k_xor_2src(float *out, unsigned long long *clk)
{
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
            _Pragma("unroll") for (int i = 0; i < 8; i++)
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 123456u) out[0] = 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
__global__ __launch_bounds__(256) void k_fma_3src(float *out, unsigned long long *clk)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 1e-3f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
            _Pragma("unroll") for (int i = 0; i < 8; i++)
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7]), "v"(a[(i + 6) & 7]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 123456.f) out[0] = 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// ---- opcode-mix probes: independent registers, fixed repeating patterns ----------------------
// PAT is a sequence of asm statements over x[8] (u32), f[8] (f32), p[8] (u64); N = statements per i
#define KMIX(NAME, BODY)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk)       \
    {                                                                                      \
        unsigned x[8]; float f[8]; unsigned long long p[8];                                \
        for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 2654435761u + i; f[i] = 1.0f + i + threadIdx.x * 1e-3f; p[i] = i; } \
        unsigned m = 0xD2511F53u, key = __builtin_amdgcn_readfirstlane(blockIdx.x) + 12345u; float c = 1.0001f; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                           \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { BODY }                     \
            }                                                                              \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        unsigned long long s = 0;                                                          \
        for (int i = 0; i < 8; i++) s += x[i] + (unsigned)f[i] + p[i];                     \
        if (s == 123456ull) out[0] = 1.0f;                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }   \
    }
#define A_XOR  asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "s"(key));
#define A_ADD  asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
#define A_CVT  asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(x[i]));
#define A_MAD  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "v"(x[i]), "s"(m) : "vcc");
#define A_EXP  asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
#define A_FMA  asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(c));
KMIX(k_mix_xor_add, A_XOR A_ADD)
KMIX(k_mix_xor_cvt, A_XOR A_CVT)
KMIX(k_mix_xor_mad, A_XOR A_MAD)
KMIX(k_mix_xor2_mad, A_XOR A_XOR A_MAD)
KMIX(k_mix_xor4_mad, A_XOR A_XOR A_XOR A_XOR A_MAD)
KMIX(k_mix_exp_xor, A_EXP A_XOR)
KMIX(k_mix_exp_xor3, A_EXP A_XOR A_XOR A_XOR)
KMIX(k_mix_exp_mad, A_EXP A_MAD)
KMIX(k_mix_exp_fma, A_EXP A_FMA)

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
        {"v_ldexp_f64", k_ldexp_f64}, {"v_frexp_mant_f64", k_frexp_mant_f64}, {"v_frexp_exp_i32_f64", k_frexp_exp_f64},
        {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_rsq_f64", k_rsq_f64}, {"v_max_f64", k_max_f64}, {"v_mov_b64", k_mov_b64},
        {"v_fmac_f64", k_fmac_f64}, {"v_cmp_lt_f64 + v_mov /2", k_cmp_f64}, {"v_readlane_b32 (+s_xor)", k_readlane},
        {"v_xor_b32(sgpr)", k_xor_sgpr}, {"v_add_f32(sgpr)", k_add_f32_sgpr}, {"v_sub_f32 clamp", k_sub_clamp},
        {"v_sub_f32", k_sub_f32}, {"v_and_b32", k_and_b32}, {"v_or_b32", k_or_b32}, {"v_lshlrev_b32", k_lshl_b32},
        {"v_lshrrev_b32", k_lshr_b32}, {"v_mov_b32", k_mov_b32}, {"v_fmac_f32", k_fmac_f32}, {"v_min_f32", k_min_f32},
        {"v_max_i32", k_max_i32}, {"v_cndmask_b32", k_cndmask}, {"v_bitop3_b32", k_bitop3}, {"2x v_xor (pair)", k_xor3},
        {"v_xor dep-chain", k_xor_dep}, {"v_add_f32 dep", k_add_dep}, {"v_fma_f32 dep", k_fma_dep},
        {"philox mix x3", k_philox_mix}, {"philox mix far x3", k_philox_mix_far}, {"v_xor 2 fresh src", k_xor_2src},
        {"v_fma 3 fresh src", k_fma_3src},
        {"[xor,add] /2", k_mix_xor_add}, {"[xor,cvt] /2", k_mix_xor_cvt}, {"[xor,mad] /2", k_mix_xor_mad},
        {"[xor,xor,mad] /3", k_mix_xor2_mad}, {"[xor x4,mad] /5", k_mix_xor4_mad}, {"[exp,xor] /2", k_mix_exp_xor},
        {"[exp,xor x3] /4", k_mix_exp_xor3}, {"[exp,mad] /2", k_mix_exp_mad}, {"[exp,fma] /2", k_mix_exp_fma},
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
