// any_order.hip -- can a second launch on the SAME stream start while the first one is still running?
//
//   hipcc --offload-arch=gfx950 -O2 -o any_order tools/c/any_order.hip && ./any_order
//
// A pricing call that ends in a partial wave-trip wants its remainder as a finer-grained second launch that fills the
// SIMDs the first launch leaves idle (cva_dates_kernel behind cva_kernel).  Back-to-back launches on one HIP stream are
// serialised by the barrier bit of the AQL packet; hipExtLaunchKernel(..., hipExtAnyOrderLaunch) clears it.  hip_ext.h says
// the flag "is not supported on AMD GFX9xx boards" for the module-launch form: this probe measures what gfx950 does.
// Each kernel's lane 0 of workgroup 0 records s_memrealtime (100 MHz) at entry and exit.
//   A = a grid that fills the chip for `spin_us`, B = a small grid of short workgroups launched right behind it:
//   (1) plain launch, same stream   (2) any-order launch, same stream   (3) plain launch, second stream
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long *stamp)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stamp[0] = t0;
        stamp[1] = wall_clock64();
    }
}

int main(int argc, char **argv)
{
    const double a_us = argc > 1 ? atof(argv[1]) : 400.0, b_us = argc > 2 ? atof(argv[2]) : 20.0;
    unsigned long long *d, h[4];
    CHECK(hipMalloc(&d, sizeof h));
    hipStream_t s0, s1;
    CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const unsigned long long ta = (unsigned long long)(a_us * 100), tb = (unsigned long long)(b_us * 100);
    // A = one and a half rounds of resident workgroups (8 x 256 lanes per CU are resident at once): its second round leaves
    // half of every CU free from a_us to 2 a_us -- room B can only use if it does not wait for A
    const int grid_a = 256 * 8 + 256 * 4, grid_b = 256;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemsetAsync(d, 0, sizeof h, s0));
            CHECK(hipStreamSynchronize(s0));
            unsigned long long *da = d, *db = d + 2;
            hipLaunchKernelGGL(spin, dim3(grid_a), dim3(256), 0, s0, ta, da);
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(grid_b), dim3(256), 0, s0, tb, db);
            } else if (mode == 1) {
                void *args[] = {(void *)&tb, (void *)&db};
                CHECK(hipExtLaunchKernel((const void *)spin, dim3(grid_b), dim3(256), args, 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch));
            } else {
                hipLaunchKernelGGL(spin, dim3(grid_b), dim3(256), 0, s1, tb, db);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            const double a0 = 0, a1 = (double)(h[1] - h[0]) / 100, b0 = (double)((long long)(h[2] - h[0])) / 100, b1 = (double)((long long)(h[3] - h[0])) / 100;
            if (rep)
                printf("%-34s A wg0 [%.1f, %.1f] us   B wg0 [%.1f, %.1f] us   -> B %s\n",
                       mode == 0 ? "same stream, plain" : mode == 1 ? "same stream, hipExtAnyOrderLaunch" : "second stream", a0, a1, b0, b1,
                       b0 < 2 * a_us - 5 ? "STARTED BEFORE A WAS DONE" : "waited for A");
        }
    }
    return 0;
}
