// rocrand_grid_device.hip -- the reference's RNG calls as a HIP build of it makes them, on the GPU: every thread of a
// (num_blocks x num_threads) launch does rocrand_init(blockIdx.x + gridDim.x, threadIdx.x, 0, &state) and then draws
// rocrand_normal(&state) `count` times (hiprand_init / hiprand_normal are these; dp/MonteCarloKernel.cu:285-290,68).
// Prints the normals as float bit patterns, thread after thread.  tests/test_gpu_grid.py compares the engine's
// launch-geometry mode (mc_grid_normals) with this bit for bit.  Nothing of the engine is linked here.
//   usage: rocrand_grid_device num_blocks num_threads count
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_normal.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void draw(int count, float *out)
{
    rocrand_state_xorwow st;
    rocrand_init(blockIdx.x + gridDim.x, threadIdx.x, 0, &st);
    float *row = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * count;
    for (int k = 0; k < count; ++k)
        row[k] = rocrand_normal(&st);
}

int main(int argc, char **argv)
{
    if (argc != 4)
        return 2;
    const int G = atoi(argv[1]), T = atoi(argv[2]), count = atoi(argv[3]);
    if (G < 1 || T < 1 || T > 1024 || count < 1)
        return 2;
    const size_t n = (size_t)G * T * count;
    float *d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess)
        return 1;
    draw<<<G, T>>>(count, d);
    std::vector<float> h(n);
    if (hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    for (size_t i = 0; i < n; ++i) {
        unsigned int bits;
        memcpy(&bits, &h[i], 4);
        printf("%08x\n", bits);
    }
    (void)hipFree(d);
    return 0;
}
