// rocrand_normal_xcheck.cpp -- prints rocRAND's own rocrand_normal() stream of an XORWOW state (host-callable engine:
// rocrand_xorwow.h / rocrand_normal.h are __host__ __device__), as float bit patterns.  This is what hiprand_normal
// returns in a HIP build of the reference (curand_init(seed, subsequence, 0) + curand_normal, dp/MonteCarloKernel.cu:
// 285-290,68); the oracle's orc_grid_normals is compared with it bit for bit.  Built and run by tests/test_rocrand_xcheck.py.
//   stdin lines: seed subsequence count   ->   stdout: `count` hex float patterns
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_normal.h>

#include <cstdio>
#include <cstring>

int main()
{
    unsigned long long seed, sub;
    int count;
    while (scanf("%llu %llu %d", &seed, &sub, &count) == 3) {
        rocrand_state_xorwow st;
        rocrand_init(seed, sub, 0ull, &st);
        for (int i = 0; i < count; ++i) {
            const float z = rocrand_normal(&st);
            unsigned int bits;
            memcpy(&bits, &z, 4);
            printf("%08x%c", bits, i + 1 == count ? '\n' : ' ');
        }
    }
    return 0;
}
