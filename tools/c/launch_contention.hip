// launch_contention.hip -- do kernel launches from several host threads onto ONE device run side by side inside the HIP
// runtime, or one after the other?  libmc_multi's launcher threads (csrc/mc_multi_host.hpp: LaunchCrew) can only start
// G devices together if they do; on a one-GPU box the G contexts share the device, so this is the only part of that
// question the box can answer.
//   T threads, one stream each, every thread enqueues N empty kernels (draining its stream every 64 so that no queue
//   fills); all threads start from one spin barrier.  Printed: host time per launch CALL, per thread, for T = 1, 2, 4, 8,
//   and the time from the barrier to the moment EVERY thread has its first launch enqueued (the fan-out of one call).
//   hipcc -O2 --offload-arch=gfx950 launch_contention.hip -o launch_contention -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void empty_kernel(int *p)
{
    if (p && threadIdx.x == 12345)
        *p = 0;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        printf("no HIP device\n");
        return 1;
    }
    printf("# launch_contention: %d device(s) visible; every thread launches on device 0, one stream per thread; %d rounds\n", ndev, rounds);
    printf("# a round = all T threads released together, each enqueues ONE empty kernel; fan-out = release -> last enqueue returned\n");
    printf("%8s %22s %22s %22s\n", "threads", "launch call us (med)", "launch call us (p99)", "fan-out us (med)");
    for (int T : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st((size_t)T);
        for (auto &s : st)
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
                return 1;
        std::atomic<int> go{0}, done{0};
        std::atomic<bool> quit{false};
        std::vector<std::vector<double>> cost((size_t)T);
        std::vector<double> end_at((size_t)T, 0.0);
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                int seen = 0;
                for (;;) {
                    int cur;
                    while ((cur = go.load(std::memory_order_acquire)) == seen)
                        if (quit.load(std::memory_order_relaxed))
                            return;
                    seen = cur;
                    const double a = now_us();
                    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st[(size_t)t], (int *)nullptr);
                    const double b = now_us();
                    cost[(size_t)t].push_back(b - a);
                    end_at[(size_t)t] = b;
                    if ((seen & 63) == 0)
                        (void)hipStreamSynchronize(st[(size_t)t]);
                    done.fetch_add(1, std::memory_order_release);
                }
            });
        std::vector<double> fan;
        for (int r = 1; r <= rounds + 50; ++r) {
            done.store(0, std::memory_order_relaxed);
            const double t0 = now_us();
            go.store(r, std::memory_order_release);
            const double a = now_us();
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st[0], (int *)nullptr);
            const double b = now_us();
            if ((r & 63) == 0)
                (void)hipStreamSynchronize(st[0]);
            while (done.load(std::memory_order_acquire) != T - 1) {
            }
            double last = b;
            for (int t = 1; t < T; ++t)
                last = std::max(last, end_at[(size_t)t]);
            if (r > 50) {   // the first rounds carry code-object loading and queue creation
                cost[0].push_back(b - a);
                fan.push_back(last - t0);
            } else {
                for (int t = 1; t < T; ++t)
                    cost[(size_t)t].clear();
            }
        }
        quit.store(true);
        for (auto &x : th)
            x.join();
        std::vector<double> all;
        for (auto &c : cost) {
            all.insert(all.end(), c.begin(), c.end());
        }
        std::vector<double> sorted = all;
        std::sort(sorted.begin(), sorted.end());
        printf("%8d %22.2f %22.2f %22.2f\n", T, median(all), sorted.empty() ? 0.0 : sorted[sorted.size() * 99 / 100], median(fan));
        for (auto &s : st) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    }
    printf("# serial launches cost T x the one-thread figure; if the fan-out of T threads stays near the one-thread figure the\n"
           "# runtime runs launches of different streams side by side, if it grows like T they queue on a lock inside it.\n");
    return 0;
}
