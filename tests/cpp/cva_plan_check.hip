// cva_plan_check.hip -- prints what csrc/mc_launch_shape.hpp's cva_plan decides for a list of calls (tests/test_host_logic.py reads it):
// one line per case "paths dates real_bytes forced -> main_paths tail_paths log2_lanes".  Host code only; nothing is launched.
#include <cstdio>
#include <cstdlib>

#include "mc_launch_shape.hpp"

int main(int argc, char **argv)
{
    for (int i = 1; i + 3 < argc; i += 4) {
        const uint64_t n = strtoull(argv[i], nullptr, 10);
        const int dates = atoi(argv[i + 1]), bytes = atoi(argv[i + 2]), forced = atoi(argv[i + 3]);
        const mc::CvaPlan p = mc::cva_plan(forced, n, dates, 256, true, (size_t)bytes);
        printf("%llu %d %d %d -> %llu %llu %d\n", (unsigned long long)n, dates, bytes, forced, (unsigned long long)p.main_paths,
               (unsigned long long)p.tail_paths, p.log2_lanes);
    }
    return 0;
}
