// mc_reduce.hpp -- (sum, sum2) reduction for gfx950: DPP inside the 64-lane wave, LDS across
// the waves of a workgroup, one 16-byte store per workgroup, and a small finishing kernel that
// adds the per-workgroup pairs in a fixed order (bitwise reproducible for a given grid).
//
// Replaces the reference's shared-memory tree (dp/MonteCarloKernel.cu:157-176, log2(T)
// __syncthreads rounds over 2*T reals) and its host loop over blocks (:416-419, :462-465).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

// DPP controls (gfx9 encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <- lane 7-i within each 8
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <- lane 15-i within each row of 16
constexpr int DPP_ROW_BCAST15 = 0x142;     // lane 15 of a row -> every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143;     // lane 31 -> every lane of rows 2,3

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v)
{
    // a 64-bit value moves as two 32-bit DPP movs; lanes whose row is masked off (or whose
    // source lane does not exist) receive +0.0
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes of a wave; the total is returned in every lane (broadcast from 63).
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_fetch<DPP_QUAD_XOR1, 0xF>(v);
    v += dpp_fetch<DPP_QUAD_XOR2, 0xF>(v);
    v += dpp_fetch<DPP_ROW_HALF_MIRROR, 0xF>(v);
    v += dpp_fetch<DPP_ROW_MIRROR, 0xF>(v);       // every lane: sum of its row of 16
    v += dpp_fetch<DPP_ROW_BCAST15, 0xA>(v);      // rows 1,3 += rows 0,2
    v += dpp_fetch<DPP_ROW_BCAST31, 0xC>(v);      // rows 2,3 += (rows 0+1)  -> lane 63 = total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

constexpr int MAX_WAVES_PER_GROUP = 16;

// Workgroup reduction of a (sum, sum2) pair; valid in thread 0 on return.
__device__ __forceinline__ void group_sum2(double &s, double &q)
{
    __shared__ double lds_s[MAX_WAVES_PER_GROUP];
    __shared__ double lds_q[MAX_WAVES_PER_GROUP];
    s = wave_sum(s);
    q = wave_sum(q);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_waves = (blockDim.x + 63) >> 6;
    if (lane == 0) {
        lds_s[wave] = s;
        lds_q[wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = lds_s[0], tq = lds_q[0];
        for (int w = 1; w < n_waves; ++w) {
            ts += lds_s[w];
            tq += lds_q[w];
        }
        s = ts;
        q = tq;
    }
    __syncthreads();  // the LDS slots may be reused by a following call (kernels that reduce several pairs)
}

// Per-workgroup partial: one coalescable 16-byte store.
__device__ __forceinline__ void store_partial(double2 *partials, double s, double q)
{
    if (threadIdx.x == 0)
        partials[blockIdx.x] = make_double2(s, q);
}

// Finishing kernel: one workgroup adds `count` partial pairs (fixed order) and writes the
// all-reduce payload {scale1 * sum, scale2 * sum2, n}.
__global__ __launch_bounds__(256) void finish_kernel(const double2 *__restrict__ partials, int count,
                                                     double scale1, double scale2, double n_paths,
                                                     double *__restrict__ triple)
{
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const double2 p = partials[i];
        s += p.x;
        q += p.y;
    }
    group_sum2(s, q);
    if (threadIdx.x == 0) {
        triple[0] = scale1 * s;
        triple[1] = scale2 * q;
        triple[2] = n_paths;
    }
}

}  // namespace mc
