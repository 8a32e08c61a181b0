// null_call.hip -- the floor of a synchronous small call on this platform: how long from hipLaunchKernelGGL to the host
// SEEING a result that the kernel wrote into pinned host memory (polled from user space), for
//   (a) one wave writing one flag,
//   (b) 256 workgroups (one per CU), each writing its own 32-byte slot {x, y, sequence}, the host polling all 256,
//   (c) 256 workgroups, last arriver (one atomic ticket) reads the 256 device slots and writes one host flag
//       -- the shape of mc_reduce.hpp's fused finish, without any simulation work.
// What mc_vanilla_run_f32 costs above (c) is simulation + set-up; what (b) saves against (c) is what a host-side final
// reduction could buy for small synchronous calls.   hipcc -O2 --offload-arch=gfx950 null_call.hip -o null_call
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { if ((x) != hipSuccess) { printf("HIP error at %d\n", __LINE__); return 1; } } while (0)

__global__ void one_flag(volatile unsigned long long *host, unsigned long long seq)
{
    if (threadIdx.x == 0)
        __hip_atomic_store((unsigned long long *)host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Slot { double x, y; unsigned long long seq, pad; };

__global__ __launch_bounds__(256) void slots(Slot *host, unsigned long long seq)
{
    if (threadIdx.x == 0) {
        Slot *s = host + blockIdx.x;
        s->x = 1.0 + blockIdx.x;
        s->y = 2.0;
        __hip_atomic_store(&s->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void ticket(double2 *dev, unsigned int *tk, volatile unsigned long long *host, unsigned long long seq)
{
    __shared__ unsigned int last;
    if (threadIdx.x == 0) {
        __hip_atomic_store((unsigned long long *)&dev[blockIdx.x].x, __double_as_longlong(1.0 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == gridDim.x;
    }
    __syncthreads();
    if (!last)
        return;
    if (threadIdx.x == 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    double s = threadIdx.x < gridDim.x ? dev[threadIdx.x].x : 0.0;
    for (int o = 32; o; o >>= 1)
        s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
        *tk = 0;
        __hip_atomic_store((unsigned long long *)host, seq + (unsigned long long)(s > 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Slot *h = nullptr;
    CHECK(hipHostMalloc(&h, 256 * sizeof(Slot), hipHostMallocDefault));
    for (int i = 0; i < 256; ++i) h[i].seq = 0;
    double2 *dev = nullptr;
    unsigned int *tk = nullptr;
    CHECK(hipMalloc(&dev, 256 * sizeof(double2)));
    CHECK(hipMalloc(&tk, 128));
    CHECK(hipMemset(tk, 0, 128));
    const int reps = 2000;
    std::vector<double> a, b, c;
    unsigned long long seq = 1;
    for (int r = 0; r < reps + 50; ++r, seq += 2) {
        auto t0 = std::chrono::steady_clock::now();
        one_flag<<<1, 64, 0, st>>>((volatile unsigned long long *)&h[0].seq, seq);
        while (__atomic_load_n(&h[0].seq, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        if (r >= 50) a.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    for (int r = 0; r < reps + 50; ++r, seq += 2) {
        auto t0 = std::chrono::steady_clock::now();
        slots<<<256, 256, 0, st>>>(h, seq);
        double sum = 0;
        for (int i = 0; i < 256; ++i) {
            while (__atomic_load_n(&h[i].seq, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
            sum += h[i].x;
        }
        if (r >= 50) b.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        if (sum < 0) return 2;
    }
    for (int r = 0; r < reps + 50; ++r, seq += 2) {
        auto t0 = std::chrono::steady_clock::now();
        ticket<<<256, 256, 0, st>>>(dev, tk, (volatile unsigned long long *)&h[0].seq, seq);
        while (__atomic_load_n(&h[0].seq, __ATOMIC_ACQUIRE) != seq + 1) __builtin_ia32_pause();
        if (r >= 50) c.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("launch -> host sees the result, median of %d (us):\n  (a) one wave, one flag                          %.2f\n"
           "  (b) 256 workgroups, 256 host slots, host adds     %.2f\n  (c) 256 workgroups, ticket + last arriver + flag  %.2f\n",
           reps, median(a), median(b), median(c));
    return 0;
}
