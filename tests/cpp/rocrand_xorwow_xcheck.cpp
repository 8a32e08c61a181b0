// rocrand_xorwow_xcheck.cpp -- prints XORWOW words from rocRAND's own engine (host-callable: rocrand_xorwow.h is
// __host__ __device__, its subsequence jump uses the precomputed h_xorwow_sequence_jump_matrices).  The oracle and the
// HIP engine compute their jump matrices themselves (GF(2) squaring of the one-step matrix); this is the independent
// reference they are compared with word for word.  Built and run by tests/test_rocrand_xcheck.py.
//   stdin lines: seed subsequence count   ->   stdout: `count` hex words of rocrand_init(seed, subsequence, 0) + rocrand()
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>

#include <cstdio>

int main()
{
    unsigned long long seed, sub;
    int count;
    while (scanf("%llu %llu %d", &seed, &sub, &count) == 3) {
        rocrand_state_xorwow st;
        rocrand_init(seed, sub, 0ull, &st);
        for (int i = 0; i < count; ++i)
            printf("%08x%c", rocrand(&st), i + 1 == count ? '\n' : ' ');
    }
    return 0;
}
