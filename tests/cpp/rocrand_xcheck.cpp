// rocrand_xcheck.cpp -- prints Philox4x32-10 words from rocRAND's own engine (the vendor library the
// north star names) for counters laid out like this repo's: the engine is __host__ __device__, so the
// cross-check runs on the CPU.  Built and run by tests/test_rocrand_xcheck.py.
//
//   rocrand_init(seed, subsequence = domain << 32 | block, offset = 4 * c)
//     -> counter = {c_lo, c_hi, block, domain}, key = {seed_lo, seed_hi}   (rocrand_philox4x32_10.h:
//        seed() / discard_subsequence_impl() / discard_impl()), and rocrand4() returns that block's 4 words.
// This repo puts a unit's LOW index word in counter word 1 (mc_rng.hpp: philox_unit), i.e. c = unit with its
// two 32-bit halves exchanged; the test passes c.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    // lines of: seed c block domain (decimal), read from stdin; c = counter words 1:0 as one 64-bit number
    unsigned long long seed, unit;
    unsigned block, domain;
    while (scanf("%llu %llu %u %u", &seed, &unit, &block, &domain) == 4) {
        if (unit >> 62) {  // offset = 4 * unit must fit 64 bits in rocRAND's API
            printf("skip\n");
            continue;
        }
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, ((unsigned long long)domain << 32) | block, 4ull * unit, &st);
        const uint4 r = rocrand4(&st);
        printf("%08x %08x %08x %08x\n", r.x, r.y, r.z, r.w);
    }
    return 0;
}
