// mc_grid.hpp -- compatibility mode: the reference's launch geometry and per-thread XORWOW normal streams
// (mc_*_run_grid_*, include/mc_mi355x.h "launch geometry"; host side: mc_api.hip grid_run).
//
// Not on the hot path: these kernels only WRITE the normals a (num_blocks x num_threads) launch of the reference draws
// into HBM, in path order; the simulation kernels of mc_kernels.hpp then price them through the external-normals policy.
#pragma once
#include "mc_rng.hpp"

namespace mc {

// ---- the reference's launch geometry (compatibility mode: mc_*_run_grid_*, include/mc_mi355x.h) -------------------
// dp/MonteCarloKernel.cu:285-290 gives every THREAD of a (num_blocks x num_threads) launch its own XORWOW state:
// curand_init(seed = blockIdx.x + gridDim.x, subsequence = threadIdx.x, offset 0).  Row b * T + t of `states` is that
// state (rocRAND's seeding: what the same call gives through hipRAND on this hardware).
__global__ __launch_bounds__(256) void xorwow_grid_init_kernel(const uint32_t *__restrict__ jump, uint32_t num_blocks, uint32_t num_threads,
                                                               uint32_t *__restrict__ states)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= num_blocks * num_threads)
        return;
    uint32_t v[5], weyl;
    xorwow_seed((uint64_t)(lane / num_threads) + num_blocks, v, weyl);
    xorwow_jump(v, lane % num_threads, jump);
#pragma unroll
    for (int k = 0; k < 5; ++k)
        states[6 * (size_t)lane + k] = v[k];
    states[6 * (size_t)lane + 5] = weyl;
}

// curand_normal / rocrand_normal on an XORWOW state (rocrand_normal.h: rocrand_normal(rocrand_state_xorwow *),
// box_muller(x, y)): two words make one Box-Muller pair, the sine member is returned first and the cosine member is kept
// for the next call.  Same expressions and the same device math calls as rocRAND's header, so the normals are the ones a
// hipRAND build of the reference draws on this GPU, bit for bit (tests/test_gpu_grid.py).
struct XorwowNormalStream {
    GenXorwow rng;
    float kept;
    bool have = false;
    __device__ __forceinline__ explicit XorwowNormalStream(const uint32_t *states) : rng(states) {}
    __device__ __forceinline__ float next()
    {
        if (have) {
            have = false;
            return kept;
        }
        const uint32_t x = rng.next(), y = rng.next();
        const float u = 2.3283064e-10f + (x * 2.3283064e-10f);
        const float v = 1.46291807e-09f + (y * 1.46291807e-09f);
        const float s = sqrtf(-2.0f * logf(u));
        float sn, cs;
        __sincosf(v, &sn, &cs);
        kept = cs * s, have = true;
        return sn * s;
    }
};

// The normals of a whole grid-geometry call, laid out for the external-normals policy (GenExternal): thread t of block b
// prices paths i = t, t + T, ... < paths_per_block of its block (dp/MonteCarloKernel.cu:146,191,240) and draws `draws`
// normals for each, one after the other, from its stream; path p = b * paths_per_block + i gets row p of `ext`
// (`per_unit` Reals: the draws, widened as `double z = curand_normal(...)` does, then zeros).
template <class Real>
__global__ __launch_bounds__(256) void grid_normals_kernel(const uint32_t *__restrict__ states, uint32_t num_blocks, uint32_t num_threads,
                                                           uint64_t paths_per_block, uint32_t draws, uint32_t per_unit,
                                                           Real *__restrict__ ext)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= num_blocks * num_threads)
        return;
    const uint32_t b = lane / num_threads, t = lane % num_threads;
    XorwowNormalStream z(states);
    for (uint64_t i = t; i < paths_per_block; i += num_threads) {
        Real *row = ext + ((uint64_t)b * paths_per_block + i) * per_unit;
        for (uint32_t j = 0; j < draws; ++j)
            row[j] = (Real)z.next();
        for (uint32_t j = draws; j < per_unit; ++j)
            row[j] = (Real)0;
    }
}

}  // namespace mc
