// mc_launch_shape.hpp -- every rule that decides the SHAPE of a launch: how many workgroups a call gets, which kernel
// family prices a basket of n assets, how many pieces the launch-geometry kernels cut a thread's stream into.  Host-only
// code, included by mc_api.hip alone.  It lives in a file of its own because the committed PMC profiles
// (profiles/pmc_traffic.json: instruction counts, waves, HBM bytes per launch) describe launches of exactly this shape:
// bench.py and tools/summarize_pmc.py hash this file together with the device code object, and a change here marks the
// counts stale (tests/test_host_logic.py::test_pmc_stamp_covers_the_launch).
#pragma once
#include <stdint.h>

#include <cstdlib>

#include "../../include/mc_mi355x.h"
#include "mc_grid.hpp"   // GROUP, GRID_SEG_ALIGN

namespace mc {

static constexpr int MAX_SEGMENTS = 8;   // per call: segments of <= 2^31 units, same high word
static constexpr int MAX_GRID_SCALE = 6; // the heaviest kernels launch up to this many times the context's `blocks` (grid_for)

static int env_int(const char *name, int fallback, int lo, int hi)
{
    const char *e = getenv(name);
    const int v = e ? atoi(e) : fallback;
    return v < lo ? lo : (v > hi ? hi : v);
}

// Workgroups of a launch over n_units units: one lane per unit until the grid reaches `scale` x the context's `blocks`
// (default 8 per CU = 2048), grid-stride beyond.  `scale` (halves: 2 = 1x) grows with the weight of a unit.  All workgroups
// of a launch do equal work, but they do not finish together (CUs and XCDs run at slightly different rates, and a kernel
// whose registers admit only 4-6 workgroups per CU runs the grid in several rounds): more, smaller workgroups even that
// out.  Measured at the BASELINE sizes, grids of 4 ... 64 per CU interleaved in one process
// (tools/grid_sweep_all.py, profiles/r03_grid_sweep.log), kernel time against 8 per CU:
//     vanilla f32, basket f32 <= 12 assets    8 per CU is the optimum (16: +1...3 %)
//     vanilla f64, CVA f64 / f32             12 per CU: -1.0 / -1.9 / -1.7 %  (flat beyond; vanilla worse from 32)
//     tiled basket f32 (13..32 assets)       24 per CU: -1.8 %
//     tiled basket f64 (9..32 assets)        48 per CU: -4.2 %  (C4's kernel: still improving slowly at 64)
// The XORWOW policy keeps 1x (its per-lane states are sized by `blocks`), so do the secondary (Greeks) kernels.
static int grid_for(int blocks, uint32_t n_units, int scale_halves = 2)
{
    const uint64_t need = ((uint64_t)n_units + GROUP - 1) / GROUP, cap = (uint64_t)blocks * (uint64_t)scale_halves / 2;
    return (int)(need < cap ? (need ? need : 1) : cap);
}
constexpr int GRID_SCALE_VANILLA_F64 = 3, GRID_SCALE_CVA = 3, GRID_SCALE_TILED_F32 = 6, GRID_SCALE_TILED_F64 = 2 * MAX_GRID_SCALE;

// Vanilla launches whose units are cheap (4 or 2 paths each): a SMALL call is dominated by what grows with the grid --
// dispatch, one pair and one ticket per workgroup, the last arriver's sum over the pairs -- not by the simulation.  So a
// lane gets at least 4 units before the grid grows beyond one workgroup per CU: the 8 x 131 072-path call (the reference
// drivers' smallest size) runs 256 workgroups instead of 1024 (kernel 7.9 instead of 8.8 us in fp32, 10.1 instead of 12.0 in fp64:
// profiles/r02_call_latency.log); from 2.1e6 units (8.4e6 fp32 paths) on the grid is the context's `blocks` as before (16 units per lane
// measured no better at 1e7 paths).  MC_VANILLA_UNITS_PER_LANE=1 restores
// one unit per lane.
static int vanilla_units_per_lane()
{
    static const int v = [] {
        const char *e = getenv("MC_VANILLA_UNITS_PER_LANE");
        const int x = e ? atoi(e) : 4;
        return x < 1 ? 1 : (x > 1024 ? 1024 : x);
    }();
    return v;
}
static int grid_for_vanilla(int blocks, int compute_units, uint32_t n_units, int scale_halves)
{
    const int full = grid_for(blocks, n_units, scale_halves);
    const uint64_t per = (uint64_t)GROUP * (uint64_t)vanilla_units_per_lane();
    uint64_t want = ((uint64_t)n_units + per - 1) / per;
    const uint64_t floor_wgs = (uint64_t)(compute_units > 0 ? compute_units : 256);
    if (want < floor_wgs)
        want = floor_wgs;
    return (int)(want < (uint64_t)full ? want : (uint64_t)full);
}

// Which kernel family prices a basket of n assets (measured on MI355X: tools/generic_basket_speed.py, runs
// alternated in one gpurun call):
//   n <= basket_static_max (fp32: 12, fp64: 8)
//                      constants as kernel arguments / LDS-staged (basket_f32_kernel, basket_kernel)
//   up to 32 assets    constants as scalar-loaded tiles, normals in registers (basket_tiled_f32_kernel,
//                      basket_tiled_kernel).  fp64 9..16, one kernel per size: +3 % at 9, +14...+20 % at 10..16 over
//                      the kernel-argument form; fp32 13, 14, 16 (15 runs the 16 kernel): +5...+8 %, below 12 the
//                      LDS-staged form wins; 17..32: one kernel per multiple of 4 on the zero-padded buffer,
//                      +19...+39 % over the generic kernel
//   33..64 assets      generic tiled kernel, normals in LDS (basket_dyn_kernel, basket_dyn_f32_kernel)
//   MC_BASKET_MFMA=1   fp64, 13..16 assets: the mat-vec as v_mfma_f64_16x16x4_f64 (basket_mfma_f64_kernel).  Off by
//                      default: 5-6 % slower than the tiled kernel, the f64 matrix instruction does not run beside
//                      the vector pipe on gfx950 (profiles/r02_mfma_basket.log)
// MC_BASKET_STATIC_MAX_F32 / _F64 and MC_BASKET_TILED_MIN (read once per process) move the limits for
// experiments and for the tests that compare the families.
template <class Real>
static int basket_static_max()
{
    static const int limit = sizeof(Real) == 4 ? env_int("MC_BASKET_STATIC_MAX_F32", 12, 0, MC_MAX_ASSETS)
                                               : env_int("MC_BASKET_STATIC_MAX_F64", 8, 0, MC_MAX_ASSETS);
    return limit;
}
static bool basket_mfma()
{
    static const int on = env_int("MC_BASKET_MFMA", 0, 0, 1);
    return on != 0;
}
static int basket_tiled_min()
{
    static const int limit = env_int("MC_BASKET_TILED_MIN", 9, 9, 1000);
    return limit;
}

// CVA: how a call of n paths over n_dates dates is cut between one lane per path (cva_kernel; the date walk serial, as in the
// reference) and the date-parallel form (2^log2_lanes adjacent lanes per path, CVA_DATES_CH dates per lane and round; mc_kernels.hpp).
//   main_paths   leading paths, one lane each
//   tail_paths   trailing paths, date-parallel (both > 0: one launch of cva_split_kernel; main_paths == 0: cva_dates_kernel)
// The one-lane-per-path kernel's time is a staircase in steps of one WAVE-TRIP = 64 lanes x 4 SIMDs x CUs paths (65 536 on
// MI355X): a launch pays for whole trips -- 1 250 000 paths (C5's shard of 8: 19.07 trips) cost what 1 310 720 do, 990.9 us against
// 944.8 for 19 trips -- and a wave alone on its SIMD needs ~85 us (fp64; ~49 us fp32) for a 256-date path whatever the call's size.
// Measured with tools/c/shard_clock and tools/cva_call_latency.py (profiles/r06_shard_clock_AD_fused_split.log,
// profiles/r06_cva_call_latency.log):
//   * a call of more than the small-call limit and at most 64 trips, on a grid of at least 64 dates, that ends in a partial trip
//     of at most 60 % hands that remainder to date-parallel workgroups of the same launch, with enough lanes per path to put
//     about two of their waves on every SIMD (1 250 000 paths x 256 dates: fp64 990.9 -> 953.0 us, fp32 -4 %; the gain
//     shrinks to nothing as the remainder approaches a full trip: 0.31 of a trip -21.6 us, 0.45 -15.2, 0.53 -8.0; beyond 64 trips
//     a trip is under 1.6 % of the call);
//   * a SMALL call runs date-parallel as a whole, with lanes for ~4 waves per SIMD.  What "small" is depends on how long a
//     lane's serial walk is (the one-lane kernel's second wave on a SIMD costs it a third of the first: 256 dates fp64 85 us for
//     one trip, 113 for two; fp32 49 and 59), in both precisions alike:
//         >= 64 dates: up to 7/4 trips (256 dates, kernel time in us, fp64: 4096 paths 83 -> 15, 65 536: 84 -> 68, 98 304: 113 -> 95,
//                                       114 688: 113 -> 108; 131 072 -- the reference driver's own call, dp/cvaOpt.cu:12-15 -- 113 -> 121,
//                                       worse; fp32: 4096 paths 49 -> 11, 65 536: 49 -> 35, 98 304: 58 -> 48, 114 688: 59 -> 53, 131 072: a tie)
//         <  64 dates: up to 3/4 trip  (25 dates, fp64: 16 384 paths 15.6 -> 11.6, 49 152: 16.2 -> 15.2; 65 536: 15.7 -> 17.2, worse;
//                                       fp32: 16 384 paths 11.3 -> 8.9, 49 152: 11.6 -> 11.1; 65 536: 11.7 -> 12.2, worse)
//     (the date-parallel form pays per-lane table rows and per-lane Philox counters: +6 % at 2 lanes, +11 % at 8, +19...22 % from
//     16 on at 1e6 fp64 paths);
//   * everything else keeps one lane per path.
// `forced_lanes`: 0 = this rule, 1 = one lane per path always, 2 ... 64 = the whole call date-parallel with that many lanes
// (mc_context_set_cva_date_lanes / MC_CVA_DATE_LANES; the tests sweep it).  A path cannot use more lanes than it has
// chunks of CVA_DATES_CH dates.
constexpr int CVA_DATES_CH = 8;
struct CvaPlan { uint64_t main_paths, tail_paths; int log2_lanes; };
static int cva_max_log2_lanes(int n_dates)
{
    int l = 0;
    while (l < 6 && (CVA_DATES_CH << l) < n_dates)
        ++l;
    return l;
}
static CvaPlan cva_plan(int forced_lanes, uint64_t n, int n_dates, int compute_units, bool dates_kernel_possible, size_t real_bytes)
{
    constexpr int tail_max_pct = 60, small_fill = 4, split_max_trips = 64, long_grid = 64;
    const int small_trips_x4 = n_dates >= long_grid ? 7 : 3;   // quarter trips, inclusive
    (void)real_bytes;
    const int max_l = cva_max_log2_lanes(n_dates);
    CvaPlan p = {n, 0, 0};
    if (!dates_kernel_possible || max_l == 0 || forced_lanes == 1 || n == 0)
        return p;
    const auto log2_ceil = [](uint64_t x) { int l = 0; while ((1ull << l) < x) ++l; return l; };
    if (forced_lanes >= 2) {
        int l = 0;
        while ((2 << l) <= forced_lanes) ++l;
        p.main_paths = 0, p.tail_paths = n, p.log2_lanes = l < max_l ? l : max_l;
        return p;
    }
    const uint64_t trip = 64ull * 4ull * (uint64_t)(compute_units > 0 ? compute_units : 256);
    if (4 * n <= (uint64_t)small_trips_x4 * trip) {   // small call: all of it date-parallel, ~small_fill waves per SIMD
        int l = log2_ceil(((uint64_t)small_fill * trip + n - 1) / n);
        l = l > max_l ? max_l : l;
        if (l > 0)
            p.main_paths = 0, p.tail_paths = n, p.log2_lanes = l;
        return p;
    }
    const uint64_t r = n % trip;
    if (r == 0 || 100 * r > (uint64_t)tail_max_pct * trip || n > (uint64_t)split_max_trips * trip || n_dates < long_grid)
        return p;
    int l = log2_ceil((2 * trip + r - 1) / r);
    l = l < 1 ? 1 : (l > max_l ? max_l : l);
    p.main_paths = n - r, p.tail_paths = r, p.log2_lanes = l;
    return p;
}

// How many pieces the fused kernels cut every reference thread's stream into (mc_grid.hpp "sub-streams"): enough to put
// ~8 waves on every SIMD (the reference's 512 x 128 launch alone is ONE), as long as a piece keeps >= 16 paths; a power
// of two up to 32.  MC_GRID_SUB forces a count (1 = the reference's own layout).
static void grid_pieces(int num_blocks, int num_threads, uint64_t paths_per_block, uint32_t *sub, uint32_t *seg)
{
    static const int forced = env_int("MC_GRID_SUB", 0, 0, 32);
    const uint64_t n_max = (paths_per_block + (uint64_t)num_threads - 1) / (uint64_t)num_threads;   // thread 0's paths
    const uint64_t lanes = (uint64_t)num_blocks * (uint64_t)((num_threads + 63) / 64 * 64), want = 8ull * 1024 * 64;
    uint32_t s = 1;
    if (forced > 0) {
        while (s * 2 <= (uint32_t)forced) s *= 2;
    } else {
        while (s < 32 && lanes * s * 2 <= want && n_max >= 16ull * s * 2)
            s *= 2;
    }
    const uint64_t per = (n_max + s - 1) / s;
    *sub = s;
    *seg = (uint32_t)((per + GRID_SEG_ALIGN - 1) / GRID_SEG_ALIGN * GRID_SEG_ALIGN);
}

}  // namespace mc
