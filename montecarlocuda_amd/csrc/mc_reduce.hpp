// mc_reduce.hpp -- (sum, sum2) reduction for gfx950: DPP inside the 64-lane wave, LDS across
// the waves of a workgroup, one 16-byte store per workgroup, and the LAST workgroup of a pricing call to
// arrive adds the per-workgroup pairs in a fixed order (bitwise reproducible for a given grid) and writes
// the call's {sum, sum2, n} -- to HBM and, for synchronous calls, straight into pinned host memory.
// finish_kernel is the same final step as a second launch (the selectable two-launch form / A/B baseline).
//
// Replaces the reference's shared-memory tree (dp/MonteCarloKernel.cu:157-176, log2(T)
// __syncthreads rounds over 2*T reals), its D2H copy of the per-block pairs (:405) and its host loop over
// blocks (:416-419, :462-465).
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

template <int CTRL, int ROW_MASK = 0xF>
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

// -----------------------------------------------------------------------------------------
// End of a simulation kernel: publish the workgroup's pair and, if this workgroup is the LAST of the
// whole pricing call to arrive, add all pairs in index order and write the call's triple.
//
// Replaces the reference's D2H copy of the per-block pairs and its host loop over blocks
// (dp/MonteCarloKernel.cu:405,416-419 / :456,462-465) -- and this repo's own former second launch
// (finish_kernel below, kept as the selectable two-launch form and as the A/B baseline).
//
// Protocol (cdna_hip_programming.md Guideline 16, counter form; MI355X_MICROARCH.md "fanin"):
//   every workgroup, lane 0:  two 8-byte write-through (sc1) stores of its pair -> s_waitcnt vmcnt(0)
//                             -> relaxed agent-scope fetch_add on its ticket shard
//   the workgroup whose add completes its shard adds 1 to the top ticket; the one that completes
//   the top ticket is the last arriver:  lane 0 agent-scope acquire -> s_waitcnt vmcnt(0) -> barrier
//   -> all 256 lanes read the pairs with plain loads, lane t taking pairs t, t+256, ... (the order
//   finish_kernel uses: the sums are the same bits whichever workgroup happens to be last, so a
//   call is bitwise reproducible for a given grid) -> DPP/LDS reduction -> triple -> tickets back to 0.
// 2048 simultaneous arrivals on ONE word would serialise at ~11-13 ns each (~25 us); over 32 shards
// (+ 32 arrivals on the top word) the ticket costs ~1 us.  Each ticket word has a 128-byte line of
// its own.  The tickets are zero between calls (zeroed at context creation, re-zeroed by every last
// arriver); a call may span several launches (range segments, range edges): `total` counts the
// pairs of all of them, `slot_base` is this launch's first pair.
// -----------------------------------------------------------------------------------------
constexpr int TICKET_SHARDS = 32;
constexpr int TICKET_STRIDE = 32;                                   // uint32 words per 128-byte line
constexpr int TICKET_WORDS = (TICKET_SHARDS + 1) * TICKET_STRIDE;   // shard words, then the top word

struct Tail {
    double2 *partials;       // the call's pair buffer: `planes` planes of `plane_stride` pairs
    uint32_t *tickets;       // TICKET_WORDS words, all zero between calls
    double *triple;          // 3 doubles per plane: {scale1 sum, scale2 sum2, n_paths}
    double scale1, scale2, n_paths;
    uint32_t slot_base;      // index (within a plane) of this launch's first pair; workgroup (x, y) owns pair slot_base + x
    uint32_t pairs;          // pairs per plane of the whole call (what the last arriver adds up)
    uint32_t planes, plane_stride;
    // arrivals: every workgroup of every launch of the call draws one ticket; workgroup (x, y) of this launch is
    // arrival number ticket_base + y * gridDim.x + x of `total`.  One-dimensional grids: ticket_base == slot_base,
    // total == pairs.  total == 0 selects the two-launch form (plain store, finish_kernel follows).
    uint32_t ticket_base, total;
    // synchronous calls: pinned, host-coherent copy of plane 0's triple (device address of host memory), or nullptr.
    // The last arriver stores {sum, sum2} there and then n_paths with system-scope release semantics: the host, which
    // preset that word to a sentinel, polls it from user space and has the result ~1 us after the last workgroup is
    // done -- no D2H copy command, no sleeping synchronize (mc_api.hip: run_sync).
    double *host_triple;
};

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) uint32_t gu32_t;

// The Tail is the FIRST argument of every simulation kernel and is read only here, after the path loop, straight
// from the kernel-argument segment (scalar loads through the constant address space).  Named as a parameter it
// would be fetched in the kernel's prologue and pin 18 SGPRs for the whole path loop: vanilla_kernel<f64> went
// from 74 to 88 SGPRs that way, and from 82 SGPRs on the hardware admits only 7 of the grid's 8 workgroups per CU
// (MI355X_MICROARCH.md, Residency).  The empty asm ties the address to a value the loop produced, so the loads
// cannot move above the loop.
typedef const __attribute__((address_space(4))) Tail *tail_ptr;
__device__ __forceinline__ tail_ptr late_tail(double after)
{
    unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p) : "v"(after));
    return (tail_ptr)p;
}

// lane 0 of the workgroup publishes the pair of one plane (after group_sum2)
__device__ __forceinline__ void publish_pair(tail_ptr t, uint32_t plane, double s, double q)
{
    if (threadIdx.x != 0)
        return;
    double2 *slot = t->partials + (size_t)plane * t->plane_stride + t->slot_base + blockIdx.x;
    if (t->total == 0) {
        *slot = make_double2(s, q);
    } else {   // write-through: the bytes leave this XCD's L2 (which no other XCD can see into)
        gu64_t *g = (gu64_t *)slot;
        __hip_atomic_store(g, (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, (unsigned long long)__double_as_longlong(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Every lane of every workgroup calls this once, after its publish_pair calls.
__device__ __forceinline__ void arrive_and_finish(tail_ptr t)
{
    if (t->total == 0)
        return;
    __shared__ uint32_t lds_is_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pair stores have left before the ticket is drawn
        const uint32_t idx = t->ticket_base + blockIdx.y * gridDim.x + blockIdx.x;
        const uint32_t shard = idx % TICKET_SHARDS;
        const uint32_t in_shard = (t->total - shard + TICKET_SHARDS - 1) / TICKET_SHARDS;   // indices < total in this shard
        uint32_t last = 0;
        gu32_t *tk = (gu32_t *)t->tickets;
        if (__hip_atomic_fetch_add(tk + shard * TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == in_shard) {
            const uint32_t shards = t->total < (uint32_t)TICKET_SHARDS ? t->total : (uint32_t)TICKET_SHARDS;   // non-empty ones
            last = __hip_atomic_fetch_add(tk + TICKET_SHARDS * TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == shards;
        }
        lds_is_last = last;
    }
    __syncthreads();
    if (!lds_is_last)   // workgroup-uniform
        return;
    if (threadIdx.x == 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale lines of the pair buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t plane = 0; plane < t->planes; ++plane) {
        const double2 *pairs = t->partials + (size_t)plane * t->plane_stride;
        double s = 0.0, q = 0.0;
        for (uint32_t i = threadIdx.x; i < t->pairs; i += blockDim.x) {
            const double2 p = pairs[i];
            s += p.x;
            q += p.y;
        }
        group_sum2(s, q);
        if (threadIdx.x == 0) {
            t->triple[3 * plane + 0] = t->scale1 * s;
            t->triple[3 * plane + 1] = t->scale2 * q;
            t->triple[3 * plane + 2] = t->n_paths;
            if (plane == 0 && t->host_triple) {
                gu64_t *h = (gu64_t *)t->host_triple;
                __hip_atomic_store(h, (unsigned long long)__double_as_longlong(t->scale1 * s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(h + 1, (unsigned long long)__double_as_longlong(t->scale2 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(h + 2, (unsigned long long)__double_as_longlong(t->n_paths), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (threadIdx.x <= TICKET_SHARDS)   // ready for the next call
        __hip_atomic_store((gu32_t *)t->tickets + threadIdx.x * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the usual end of a kernel with one (sum, sum2) pair per workgroup
__device__ __forceinline__ void finish_group(double s, double q)
{
    const tail_ptr t = late_tail(s);
    publish_pair(t, 0, s, q);
    arrive_and_finish(t);
}

// Two-launch form: one workgroup adds `count` partial pairs (fixed order) and writes the
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

// One lane: three doubles from device memory into pinned host memory, the last one with system-scope release semantics --
// how a result that some OTHER kernel produced (the RCCL all-reduce of the multi-GPU path) reaches a polling host thread
// without a copy command and a sleeping synchronize (mc_api.hip: mc_context_publish).
__global__ void publish_kernel(const double *__restrict__ src, double *host)
{
    gu64_t *h = (gu64_t *)host;
    __hip_atomic_store(h, (unsigned long long)__double_as_longlong(src[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h + 1, (unsigned long long)__double_as_longlong(src[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h + 2, (unsigned long long)__double_as_longlong(src[2]), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace mc
