// mc_api.hip -- host side of libmc_mi355x.so: the C ABI of include/mc_mi355x.h.
//
// Replaces the launcher half of the reference (dp/MonteCarloKernel.cu:296-532): context
// set-up/tear-down, option upload, kernel launch, partial-sum collection, closing formulas.
// Differences by design: buffers live in a persistent context instead of being
// allocated per call; no RNG state; the cross-workgroup reduction runs on the device and the
// host reads 24 bytes; every entry point returns a status instead of exit(1).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mc_mi355x.h"
#include "../../include/mc_mi355x_test.h"
#include "mc_hostmath.h"
#include "mc_grid.hpp"
#include "mc_kernels.hpp"
#include "mc_launch_shape.hpp"

using namespace mc;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
// The per-thread error text lives in mc_hostmath.c (the host-only object shared with the CPU twin).
#define fail(...) mc_internal_fail(__VA_ARGS__)

#define HIPCHK(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(MC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                                    \
    } while (0)

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------

struct mc_context {
    int device = 0;
    int blocks = 0;
    int compute_units = 0;
    int clock_mhz = 0;
    char name[128] = {0};
    hipStream_t stream = nullptr;
    double2 *partials = nullptr;  // MAX_SEGMENTS * blocks + 2 (vanilla edge launches)
    uint32_t *tickets = nullptr;  // arrival tickets of the in-kernel final reduction (mc_reduce.hpp: Tail), zero between calls
    bool fused = true;            // final reduction inside the simulation kernel (false: second launch, finish_kernel)
    hipEvent_t last_use = nullptr;       // recorded behind every enqueued call
    hipStream_t last_stream = nullptr;   // stream of the most recent call
    bool used = false;
    double *d_triple = nullptr;   // result slot of the synchronous runs
    double2 *g_pairs = nullptr;   // multi-plane calls (Greeks): g_planes planes of 2 * blocks + 2 pairs, allocated on first use
    double *g_triples = nullptr;  // one triple per plane
    int g_planes = 0;
    double *h_triple = nullptr;   // pinned
    double *h_direct = nullptr;   // pinned + host-coherent, 3 slots of 4 doubles: [0] the triple of a synchronous call, written by
                                  // its last workgroup; [1] the same for an ARMED launch (mc_context_arm_direct); [2] mc_context_publish
    double *d_direct = nullptr;   // device address of h_direct
    double *direct_target = nullptr;   // what the Tail of the call being enqueued carries (set around the enqueue)
    bool armed = false;           // the next mc_*_launch_* delivers its triple to slot [1]
    bool timing = true;           // synchronous calls bracket their kernels with HIP events (mc_result.kernel_ms)
    void *d_out = nullptr;        // per-path dump buffer (tests), grown on demand
    size_t d_out_bytes = 0;
    void *d_table = nullptr;      // CVA per-date table
    void *h_table = nullptr;      // pinned staging for it
    size_t table_bytes = 0;
    std::vector<char> table_key;  // inputs the cached table was built from
    std::vector<char> cva_args;   // the CvaArgs<Real> that go with a cached CVA table (a launch with the same inputs reuses both)
    hipEvent_t table_copied = nullptr;
    hipStream_t table_stream = nullptr;  // stream the cached table was uploaded on
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // generator: Philox (counter-based, stateless) or XORWOW (one sequence per lane: mc_rng.hpp GenXorwow)
    int rng = MC_RNG_PHILOX;
    uint64_t xorwow_base = 0;            // lane l of a launch runs subsequence xorwow_base + l
    uint32_t *d_xorwow = nullptr;        // start states of lanes [0, blocks * GROUP): 6 words each
    uint32_t *d_xorwow_jump = nullptr;   // jump matrices A^(2^67 2^i), i < XORWOW_JUMP_BITS, then A^(2^i), i < XORWOW_OFFSET_BITS
    bool xorwow_valid = false;           // d_xorwow holds the states of (xorwow_seed, xorwow_state_base)
    uint64_t xorwow_seed = 0, xorwow_state_base = 0;
    // launch-geometry mode: one start state per (block, thread), cached for the last GRID_CACHE geometries used
    struct GridStates { int blocks, threads; uint32_t sub, step; uint32_t *d; uint64_t last_used; };   // sub pieces per thread, step words apart
    std::vector<GridStates> grid_cache;
    uint64_t grid_clock = 0;
    int grid_form = MC_GRID_FORM_AUTO;   // fused kernels where they exist, otherwise staged through HBM (MC_GRID_FORM, mc_context_set_grid_form)
    bool normals_f32 = false;     // fp64 kernels draw fp32 normals, widened (the reference's dp arithmetic): GenPhiloxF32N
    // external normals (mc_*_from_normals_*, mc_*_run_grid_*): set around one enqueue
    const void *ext = nullptr;    // device array, ext_per_unit Reals per unit
    uint32_t ext_per_unit = 0;
    int ext_flags = 0;
    void *d_ext = nullptr;        // the context's buffer for them, grown on demand
    size_t d_ext_bytes = 0;
    bool antithetic = false;      // estimator: plain (reference) or antithetic variates
    // CVA: lanes per path (mc_launch_shape.hpp: cva_plan).  0 = by call size, 1 = cva_kernel only, 2 ... 64 = cva_dates_kernel for the whole call
    int cva_date_lanes = 0;
    bool control = false;         // baskets: geometric-basket control variate
    // sampled device timing of the simulation kernels (mc_context_profile)
    int last_grid = 0, last_group = 0;   // shape of the most recent simulation launch (mc_context_last_launch)
    int profile_every = 0;
    uint64_t launches = 0;
    std::vector<hipEvent_t> prof_start, prof_stop;
    int prof_used = 0;
    // stage breakdown of the synchronous calls (mc_context_last_call_stats): host-clock accumulators of the call in progress
    mc_call_stats stats = {};
    double acc_setup_ms = 0, acc_table_ms = 0;
    std::chrono::steady_clock::time_point call_t0;   // start of a synchronous call whose first part ran before run_sync (staged launch geometry)
    bool call_t0_valid = false;
    bool stats_timed_call = false;   // a timed synchronous call is being enqueued: set-up work that completes on the device re-records ev0
    float create_ms = 0;
    uint64_t sync_calls = 0;
};

// Host-clock time of a stage of the call in progress, added to one of the context's accumulators (run_sync reads them).
struct StageTimer {
    double *acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StageTimer(double *a) : acc(a) {}
    ~StageTimer() { *acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
// Set-up that ran on the device (XORWOW states, jump matrices): waited for HERE, so that it is charged to set-up and not to the
// kernel stage, and the call's opening event is recorded again behind it.
static int setup_settle(mc_context *c, hipStream_t st);
// Defined at the end of this file.  It must not name a kernel TEMPLATE: naming one instantiates it where it is named, the order of
// instantiation is the order of the kernels in the code object, and the PMC profiles are stamped with the code object's bytes
// (a first version queried vanilla_f32_kernel's attributes and moved the stamp without changing one instruction).
static hipError_t preload_code_object();

static constexpr int PROFILE_RING = 512;

// Sampled calls time their (first) simulation kernel with the dispatch's own start/stop
// timestamps: hipExtLaunchKernelGGL attaches the two events to that one kernel, so the figure is
// the kernel's execution time as rocprofv3's kernel trace reports it (no launch latency, no
// finishing kernel).  A call split into several segments reports its first segment.
struct ProfileScope {
    mc_context *c;
    int slot = -1;
    bool used = false;
    explicit ProfileScope(mc_context *ctx) : c(ctx)
    {
        const uint64_t id = c->launches++;
        if (c->profile_every > 0 && id % (uint64_t)c->profile_every == 0 && c->prof_used < PROFILE_RING)
            slot = c->prof_used++;
    }
    ~ProfileScope()
    {
        if (slot >= 0 && !used && slot == c->prof_used - 1)
            c->prof_used--;  // nothing was launched under this sample: give the slot back
    }
};

template <class... KArgs, class... Args>
static void launch_sim_lds(ProfileScope &prof, void (*kernel)(KArgs...), int grid, size_t dynamic_lds, hipStream_t st, Args... args)
{
    prof.c->last_grid = grid, prof.c->last_group = GROUP;
    if (prof.slot >= 0 && !prof.used) {
        prof.used = true;
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(GROUP), dynamic_lds, st, prof.c->prof_start[prof.slot],
                              prof.c->prof_stop[prof.slot], 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(GROUP), dynamic_lds, st, args...);
    }
}
template <class... KArgs, class... Args>
static void launch_sim(ProfileScope &prof, void (*kernel)(KArgs...), int grid, hipStream_t st, Args... args)
{
    launch_sim_lds(prof, kernel, grid, 0, st, args...);
}

extern "C" int mc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" int mc_device_pci_bus_id(int device, char *buf, int len)
{
    if (!buf || len < 13)
        return fail(MC_ERR_INVALID, "mc_device_pci_bus_id: buffer of at least 13 bytes needed");
    const hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess)
        return fail(MC_ERR_HIP, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e));
    return MC_OK;
}

// Wait for everything the context has enqueued, on its own stream AND on the caller stream of its most recent call
// (mc_*_launch_* may run anywhere; calls on earlier streams are ordered before that one by begin_call).  Used before
// anything the kernels touch is freed or reallocated.
static int quiesce(mc_context *c)
{
    if (c->stream)
        HIPCHK(hipStreamSynchronize(c->stream));
    if (c->used && c->last_stream != c->stream)
        HIPCHK(hipStreamSynchronize(c->last_stream));
    return MC_OK;
}

static int context_allocate(mc_context *c)
{
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&c->partials, sizeof(double2) * ((size_t)MAX_SEGMENTS * c->blocks * MAX_GRID_SCALE + 2)));
    HIPCHK(hipMalloc(&c->tickets, sizeof(uint32_t) * TICKET_WORDS));
    HIPCHK(hipMemset(c->tickets, 0, sizeof(uint32_t) * TICKET_WORDS));
    HIPCHK(hipMalloc(&c->d_triple, 3 * sizeof(double)));
    HIPCHK(hipHostMalloc(&c->h_triple, 3 * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&c->h_direct, 12 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void **)&c->d_direct, c->h_direct, 0));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipEventCreateWithFlags(&c->table_copied, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->last_use, hipEventDisableTiming));
    if (const char *e = getenv("MC_FINISH"))   // "kernel": the two-launch form (A/B baseline); default: fused
        c->fused = strcmp(e, "kernel") != 0;
    if (const char *e = getenv("MC_F64_NORMALS"))   // "f32": the reference's dp arithmetic (mc_context_set_normals)
        c->normals_f32 = strcmp(e, "f32") == 0;
    c->cva_date_lanes = env_int("MC_CVA_DATE_LANES", 0, 0, 64);   // mc_context_set_cva_date_lanes
    return MC_OK;
}

extern "C" void mc_context_destroy(mc_context *c);

extern "C" int mc_context_create(int device, int blocks, mc_context **out)
{
    if (!out)
        return fail(MC_ERR_INVALID, "mc_context_create: out is NULL");
    *out = nullptr;
    const auto create0 = std::chrono::steady_clock::now();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MC_ERR_NO_DEVICE, "no HIP device visible (the HIP engine has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(MC_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    mc_context *c = new mc_context;
    c->device = device;
    c->compute_units = prop.multiProcessorCount;
    c->clock_mhz = prop.clockRate / 1000;
    snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    // 8 workgroups of 256 lanes per CU = 8 waves per SIMD: full occupancy for kernels that
    // stay within 64 VGPRs; enough workgroups (2048 on 256 CUs) to fill all 8 XCDs evenly.
    c->blocks = blocks > 0 ? blocks : 8 * c->compute_units;
    if (c->blocks > 65536) {
        delete c;
        return fail(MC_ERR_INVALID, "blocks=%d too large", blocks);
    }
    if (int rc = context_allocate(c)) {
        mc_context_destroy(c);  // frees whatever was allocated before the failure (keeps the error text)
        return rc;
    }
    // Load the device code object NOW: HIP defers it to the first launch of any kernel of the library (7-10 ms inside that launch
    // call: profiles/r05_multi_soak_300k_calls.log "first call of kernel 0"), so that the context's first pricing call costs what
    // every later one costs and the one-time work is all in mc_context_create (context_create_ms of mc_call_stats).
    if (const hipError_t e = preload_code_object(); e != hipSuccess) {
        mc_context_destroy(c);
        return fail(MC_ERR_HIP, "the device code object does not load on device %d (%s): this library is built for gfx950 only", device,
                    hipGetErrorString(e));
    }
    c->create_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - create0).count();
    if (const char *v = getenv("MC_VERBOSE"))   // 2: the resolved configuration, once per context, on stderr
        if (atoi(v) >= 2) {
            char buf[1024];
            mc_context_describe(c, buf, (int)sizeof buf);
            fprintf(stderr, "%s\n", buf);
        }
    *out = c;
    return MC_OK;
}

static int setup_settle(mc_context *c, hipStream_t st)
{
    HIPCHK(hipStreamSynchronize(st));
    if (c->stats_timed_call && st == c->stream)
        HIPCHK(hipEventRecord(c->ev0, c->stream));
    return MC_OK;
}

extern "C" int mc_context_last_call_stats(const mc_context *c, mc_call_stats *out)
{
    if (!c || !out)
        return fail(MC_ERR_INVALID, "mc_context_last_call_stats: NULL argument");
    *out = c->stats;
    return MC_OK;
}

// The resolved configuration of a context as one line of text (what MC_VERBOSE=2 prints at creation).
extern "C" int mc_context_describe(const mc_context *c, char *buf, int len)
{
    if (!c || !buf || len <= 0)
        return fail(MC_ERR_INVALID, "mc_context_describe: bad argument");
    snprintf(buf, (size_t)len,
             "mc_context config: device=%d \"%s\" CUs=%d clock_mhz=%d blocks=%d finish=%s f64_normals=%s rng=%s antithetic=%d control_variate=%d timing=%d "
             "grid_form=%s basket_static_max=f32:%d,f64:%d basket_tiled_min=%d basket_mfma=%d grid_sub=%d(0=auto) vanilla_units_per_lane=%d cva_date_lanes=%d(0=auto) "
             "created_in_ms=%.1f",
             c->device, c->name, c->compute_units, c->clock_mhz, c->blocks, c->fused ? "fused" : "kernel", c->normals_f32 ? "f32" : "native",
             c->rng == MC_RNG_XORWOW ? "xorwow" : "philox", (int)c->antithetic, (int)c->control, (int)c->timing,
             c->grid_form == MC_GRID_FORM_STAGED ? "staged" : (c->grid_form == MC_GRID_FORM_FUSED ? "fused" : "auto"), basket_static_max<float>(),
             basket_static_max<double>(), basket_tiled_min(), (int)basket_mfma(), env_int("MC_GRID_SUB", 0, 0, 32), vanilla_units_per_lane(), c->cva_date_lanes,
             c->create_ms);
    return MC_OK;
}

extern "C" void mc_context_destroy(mc_context *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)quiesce(c);   // the launch API runs on caller streams too: nothing of this context may still be in flight
    (void)hipFree(c->partials);
    (void)hipFree(c->tickets);
    (void)hipFree(c->d_xorwow);
    (void)hipFree(c->d_xorwow_jump);
    for (const mc_context::GridStates &g : c->grid_cache) (void)hipFree(g.d);
    if (c->last_use) (void)hipEventDestroy(c->last_use);
    (void)hipFree(c->d_triple);
    (void)hipFree(c->g_pairs);
    (void)hipFree(c->g_triples);
    (void)hipHostFree(c->h_triple);
    (void)hipHostFree(c->h_direct);
    (void)hipFree(c->d_out);
    (void)hipFree(c->d_ext);
    (void)hipFree(c->d_table);
    (void)hipHostFree(c->h_table);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->table_copied) (void)hipEventDestroy(c->table_copied);
    for (hipEvent_t e : c->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->prof_stop) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mc_context_device(const mc_context *c) { return c ? c->device : -1; }
extern "C" int mc_context_last_launch(const mc_context *c, int *workgroups, int *group_size)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    if (workgroups) *workgroups = c->last_grid;
    if (group_size) *group_size = c->last_group;
    return MC_OK;
}
extern "C" void *mc_context_stream(const mc_context *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int mc_context_blocks(const mc_context *c) { return c ? c->blocks : 0; }
extern "C" int mc_context_info(const mc_context *c, char *name, int name_len, int *cus, int *mhz)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    if (name && name_len > 0)
        snprintf(name, (size_t)name_len, "%s", c->name);
    if (cus) *cus = c->compute_units;
    if (mhz) *mhz = c->clock_mhz;
    return MC_OK;
}

extern "C" int mc_context_set_antithetic(mc_context *c, int on)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    c->antithetic = on != 0;
    return MC_OK;
}

extern "C" int mc_context_set_normals(mc_context *c, int mode)
{
    if (!c || (mode != MC_NORMALS_NATIVE && mode != MC_NORMALS_F32))
        return fail(MC_ERR_INVALID, "mc_context_set_normals: bad argument");
    c->normals_f32 = mode == MC_NORMALS_F32;
    return MC_OK;
}

extern "C" int mc_context_set_cva_date_lanes(mc_context *c, int lanes)
{
    if (!c || lanes < 0 || lanes > 64 || (lanes & (lanes - 1)) != 0)
        return fail(MC_ERR_INVALID, "mc_context_set_cva_date_lanes: lanes must be 0 (by call size), 1, 2, 4, ... 64");
    c->cva_date_lanes = lanes;
    return MC_OK;
}

extern "C" int mc_context_set_finish(mc_context *c, int fused)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    c->fused = fused != 0;
    return MC_OK;
}

extern "C" int mc_context_set_timing(mc_context *c, int on)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    c->timing = on != 0;
    return MC_OK;
}

extern "C" int mc_context_set_control_variate(mc_context *c, int on)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    c->control = on != 0;
    return MC_OK;
}

// E[max(G - K, 0)] of the geometric-basket control: mc_hostmath.c (closed form in fp64, see mc_mi355x.h)
static int control_mean(const mc_basket_f32 &o, double *mean) { return mc_basket_control_mean_f32(&o, mean); }
static int control_mean(const mc_basket_f64 &o, double *mean) { return mc_basket_control_mean_f64(&o, mean); }

extern "C" int mc_context_profile(mc_context *c, int every)
{
    if (!c || every < 0)
        return fail(MC_ERR_INVALID, "mc_context_profile: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (every > 0 && c->prof_start.empty()) {
        c->prof_start.resize(PROFILE_RING);
        c->prof_stop.resize(PROFILE_RING);
        for (int i = 0; i < PROFILE_RING; ++i) {
            HIPCHK(hipEventCreate(&c->prof_start[i]));
            HIPCHK(hipEventCreate(&c->prof_stop[i]));
        }
    }
    c->profile_every = every;
    c->prof_used = 0;
    c->launches = 0;
    return MC_OK;
}

extern "C" int mc_context_profile_read(mc_context *c, int *samples, double *total_ms)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    HIPCHK(hipSetDevice(c->device));
    double total = 0;
    for (int i = 0; i < c->prof_used; ++i) {
        HIPCHK(hipEventSynchronize(c->prof_stop[i]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_start[i], c->prof_stop[i]));
        total += ms;
    }
    if (samples) *samples = c->prof_used;
    if (total_ms) *total_ms = total;
    c->prof_used = 0;
    c->launches = 0;
    return MC_OK;
}

// ---------------------------------------------------------------------------------------
// work planning
// ---------------------------------------------------------------------------------------
struct Segment {
    uint64_t first;
    uint32_t count;
};

// Split units [first, first+count) so that no segment crosses a multiple of 2^32 (the high
// counter word stays wave-uniform) nor exceeds 2^31 units (32-bit strided loop cannot wrap).
static int plan_segments(uint64_t first, uint64_t count, std::vector<Segment> &out)
{
    out.clear();
    while (count) {
        uint64_t room = (1ull << 32) - (first & 0xFFFFFFFFull);
        uint64_t n = count < room ? count : room;
        if (n > (1ull << 31))
            n = 1ull << 31;
        out.push_back({first, (uint32_t)n});
        first += n;
        count -= n;
        if ((int)out.size() > MAX_SEGMENTS)
            return fail(MC_ERR_INVALID, "path range too large for one call (more than %d segments of 2^31 units); split it",
                        MAX_SEGMENTS);
    }
    return MC_OK;
}

static Work make_work(uint64_t seed, const Segment &s, uint64_t first_path, uint64_t end_path)
{
    Work w;
    w.seed_lo = (uint32_t)seed;
    w.seed_hi = (uint32_t)(seed >> 32);
    w.unit_lo = (uint32_t)s.first;
    w.unit_hi = (uint32_t)(s.first >> 32);
    w.n_units = s.count;
    w.first_path = first_path;
    w.end_path = end_path;
    w.xorwow = nullptr;
    w.ext = nullptr;
    w.ext_per_unit = 0;
    return w;
}

// Which generator policy (mc_rng.hpp) a launch runs with, and the dispatch from that run-time choice to the kernel
// instantiation: f(gen_tag<Gen>) is instantiated only for the policies in ALLOW -- each kernel family is compiled for the
// generators it supports and nothing else.
enum GenSel { GEN_PHILOX = 1, GEN_XORWOW = 2, GEN_F32N = 4, GEN_EXTERNAL = 8 };
template <class G> struct gen_tag { using type = G; };
static GenSel gen_of(const mc_context *c, const Work &w, size_t real_bytes)
{
    if (w.ext) return GEN_EXTERNAL;
    if (w.xorwow) return GEN_XORWOW;
    return (real_bytes == 8 && c->normals_f32) ? GEN_F32N : GEN_PHILOX;
}
template <unsigned ALLOW, class F>
static int with_gen(GenSel g, F f)
{
    if constexpr ((ALLOW & GEN_PHILOX) != 0)
        if (g == GEN_PHILOX) { f(gen_tag<GenPhilox>{}); return MC_OK; }
    if constexpr ((ALLOW & GEN_XORWOW) != 0)
        if (g == GEN_XORWOW) { f(gen_tag<GenXorwow>{}); return MC_OK; }
    if constexpr ((ALLOW & GEN_F32N) != 0)
        if (g == GEN_F32N) { f(gen_tag<GenPhiloxF32N>{}); return MC_OK; }
    if constexpr ((ALLOW & GEN_EXTERNAL) != 0)
        if (g == GEN_EXTERNAL) { f(gen_tag<GenExternal>{}); return MC_OK; }
    return fail(MC_ERR_UNSUPPORTED, "this kernel is not compiled for the selected generator / estimator combination");
}
// plain / antithetic x generator; the external-normals policy exists for the plain estimator only
template <unsigned ALLOW, class F>
static int with_anti_gen(bool anti, GenSel g, F f)
{
    if (anti)
        return with_gen<(ALLOW & ~(unsigned)GEN_EXTERNAL)>(g, [&](auto tag) { f(std::true_type{}, tag); });
    return with_gen<ALLOW>(g, [&](auto tag) { f(std::false_type{}, tag); });
}
// normals one block yields for precision Real under the context's settings (GenPhiloxF32N: 4 in fp64 too)
template <class Real> static uint64_t npb_of(const mc_context *c) { return (sizeof(Real) == 8 && !c->normals_f32) ? 8 : 4; }

// the Work of a launch, with the context's generator inputs attached
static Work context_work(const mc_context *c, uint64_t seed, const Segment &s, uint64_t first_path, uint64_t end_path)
{
    Work w = make_work(seed, s, first_path, end_path);
    if (c->ext) {
        w.ext = c->ext;
        w.ext_per_unit = c->ext_per_unit;
    } else if (c->rng == MC_RNG_XORWOW) {
        w.xorwow = c->d_xorwow;
    }
    return w;
}

// One pricing call = one or more simulation launches that share the context's pair buffer.  The Tail tells every
// launch where its pairs go and how many pairs the whole call has, so that the last workgroup to arrive can close the
// call inside the kernel (mc_reduce.hpp); in the two-launch form (MC_FINISH=kernel) total stays 0 and
// finish_call launches finish_kernel behind the simulation kernels.
static Tail make_tail(const mc_context *c, int total, double scale1, double scale2, uint64_t n, double *d_triple,
                      int planes = 1, int plane_stride = 0)
{
    Tail t;
    t.partials = c->partials;
    t.tickets = c->tickets;
    t.triple = d_triple;
    t.scale1 = scale1;
    t.scale2 = scale2;
    t.n_paths = (double)n;
    t.slot_base = t.ticket_base = 0;
    t.pairs = (uint32_t)total;
    t.total = c->fused ? (uint32_t)total : 0u;
    t.planes = (uint32_t)planes;
    t.plane_stride = (uint32_t)plane_stride;
    t.host_triple = c->fused ? c->direct_target : nullptr;
    return t;
}

// The context owns ONE pair buffer, ONE ticket block and ONE constant table, so its calls must execute one after the
// other.  Calls on one stream are ordered by the stream; a call on a DIFFERENT stream than the previous one is ordered
// behind everything enqueued so far on that previous stream (event record there, wait here) -- correct for any
// interleaving, and free in the usual case of one stream per context.
static int begin_call(mc_context *c, hipStream_t st)
{
    HIPCHK(hipSetDevice(c->device));
    if (c->used && st != c->last_stream) {
        HIPCHK(hipEventRecord(c->last_use, c->last_stream));
        HIPCHK(hipStreamWaitEvent(st, c->last_use, 0));
    }
    c->last_stream = st;
    c->used = true;
    return MC_OK;
}

// Make `stream` wait for everything this context has enqueued so far (on whatever stream its last call used): the
// hand-over from a launch stream to the stream of a consumer of the triple (a copy, an RCCL call).
extern "C" int mc_context_order(mc_context *c, void *stream)
{
    if (!c)
        return fail(MC_ERR_INVALID, "NULL context");
    HIPCHK(hipSetDevice(c->device));
    if (c->used && (hipStream_t)stream != c->last_stream) {
        HIPCHK(hipEventRecord(c->last_use, c->last_stream));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->last_use, 0));
    }
    return MC_OK;
}

// 1 when everything this context has enqueued has completed, 0 while work is pending, -1 on error: hipStreamQuery
// of the stream its last call used -- lets a host poll from user space instead of sleeping in a synchronize.
extern "C" int mc_context_idle(mc_context *c)
{
    if (!c)
        return -1;
    if (!c->used)
        return 1;
    if (hipSetDevice(c->device) != hipSuccess)
        return -1;
    const hipError_t e = hipStreamQuery(c->last_stream);
    return e == hipSuccess ? 1 : (e == hipErrorNotReady ? 0 : -1);
}

// ---- results straight into pinned host memory for the asynchronous launch API (the multi-GPU library's read-back) ----
static constexpr double DIRECT_SENTINEL = -1.0;   // n_paths is never negative

// Arms the context: the NEXT mc_*_launch_* call also has its last workgroup store {sum, sum2, n} into a pinned,
// host-coherent slot the context owns (n last, with system-scope release semantics).  *slot = its host address; word 2
// is preset to -1, so a host thread polls `(*slot)[2] != -1` from user space and then reads the triple -- no D2H copy
// command, no event, no sleeping synchronize.  One armed launch in flight per context.
extern "C" int mc_context_arm_direct(mc_context *c, const volatile double **slot)
{
    if (!c || !slot)
        return fail(MC_ERR_INVALID, "mc_context_arm_direct: bad argument");
    if (!c->fused)
        return fail(MC_ERR_UNSUPPORTED, "direct delivery needs the fused final reduction (MC_FINISH=kernel is set)");
    volatile double *s = c->h_direct + 4;
    s[2] = DIRECT_SENTINEL;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    c->armed = true;
    *slot = s;
    return MC_OK;
}

// Enqueues on `stream` a one-lane kernel that copies the three doubles at d_src (device memory, e.g. the output of an
// all-reduce enqueued before it on that stream) into another pinned slot of the context, same protocol as above.
extern "C" int mc_context_publish(mc_context *c, const double *d_src, void *stream, const volatile double **slot)
{
    if (!c || !d_src || !slot)
        return fail(MC_ERR_INVALID, "mc_context_publish: bad argument");
    // a call of the context like any other: ordered behind its previous call, and recorded as its last stream, so that
    // quiesce() -- before the pinned slots are freed -- waits for this kernel too whatever stream it runs on
    if (int rc = begin_call(c, (hipStream_t)stream)) return rc;
    volatile double *s = c->h_direct + 8;
    s[2] = DIRECT_SENTINEL;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    publish_kernel<<<1, 1, 0, (hipStream_t)stream>>>(d_src, c->d_direct + 8);
    HIPCHK(hipGetLastError());
    *slot = s;
    return MC_OK;
}

// moves the armed state into the enqueue that follows (the launch entry points construct one)
struct ArmScope {
    mc_context *c;
    bool on;
    explicit ArmScope(mc_context *ctx) : c(ctx), on(ctx && ctx->armed)
    {
        if (on) {
            c->direct_target = c->d_direct + 4;
            c->armed = false;
        }
    }
    ~ArmScope()
    {
        if (on)
            c->direct_target = nullptr;
    }
};

static int finish_call(mc_context *c, const Tail &t, int total, hipStream_t st)
{
    if (!c->fused)
        for (uint32_t q = 0; q < t.planes; ++q)
            finish_kernel<<<1, GROUP, 0, st>>>(t.partials + (size_t)q * t.plane_stride, total, t.scale1, t.scale2, t.n_paths,
                                               t.triple + 3 * q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        // Some launches of the call may have been enqueued and will draw tickets, but the call will never reach its
        // `total`: left alone the ticket block stays non-zero and every later call on the context would close early or
        // never.  Drain what was enqueued and put the tickets back to zero before reporting the error.
        (void)hipStreamSynchronize(st);
        (void)hipMemsetAsync(c->tickets, 0, sizeof(uint32_t) * TICKET_WORDS, st);
        (void)hipStreamSynchronize(st);
        return fail(MC_ERR_HIP, "kernel launch failed: %s (the call's tickets were reset)", hipGetErrorString(e));
    }
    return MC_OK;
}

// ---------------------------------------------------------------------------------------
// XORWOW: jump matrices and per-lane start states (device side: mc_rng.hpp)
// ---------------------------------------------------------------------------------------
// The xorshift part of XORWOW is a linear map A on 160 bits; jumping a lane to its own subsequence (2^67 words further
// per subsequence number, rocRAND's layout) is a product of matrices A^(2^67 2^i).  They are computed here, once per
// process, by squaring the one-step matrix 67 + XORWOW_JUMP_BITS - 1 times (about 3 ms) -- nothing is taken from
// rocRAND's precomputed tables; tests/test_rocrand_xcheck.py compares the resulting words with rocRAND's own engine.
namespace {
struct XwVec { uint32_t w[5]; };
struct XwMat { XwVec col[160]; };   // column c = image of state bit c (word c / 32, bit c % 32)
XwVec xw_step(XwVec v)
{
    const uint32_t t = v.w[0] ^ (v.w[0] >> 2);
    return {{v.w[1], v.w[2], v.w[3], v.w[4], (v.w[4] ^ (v.w[4] << 4)) ^ (t ^ (t << 1))}};
}
XwVec xw_apply(const XwMat &m, const XwVec &v)
{
    // branch-free (a mask per state bit): the bits are random, a conditional xor mispredicts every other time -- 3 ms
    // instead of 13 for the whole table
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
    for (int c = 0; c < 160; ++c) {
        const uint32_t mask = 0u - ((v.w[c >> 5] >> (c & 31)) & 1u);
        r0 ^= mask & m.col[c].w[0], r1 ^= mask & m.col[c].w[1], r2 ^= mask & m.col[c].w[2], r3 ^= mask & m.col[c].w[3], r4 ^= mask & m.col[c].w[4];
    }
    return {{r0, r1, r2, r3, r4}};
}
void xw_square(XwMat &m)
{
    XwMat r;
    for (int c = 0; c < 160; ++c)
        r.col[c] = xw_apply(m, m.col[c]);
    m = r;
}
const std::vector<uint32_t> &xorwow_jump_table()
{
    static const std::vector<uint32_t> table = [] {
        XwMat a;
        for (int c = 0; c < 160; ++c) {
            XwVec e = {{0, 0, 0, 0, 0}};
            e.w[c >> 5] = 1u << (c & 31);
            a.col[c] = xw_step(e);
        }
        std::vector<uint32_t> t((size_t)XORWOW_JUMP_MATRICES * 160 * 5);
        const auto store = [&](int slot) {
            for (int c = 0; c < 160; ++c)
                for (int k = 0; k < 5; ++k)
                    t[((size_t)slot * 160 + c) * 5 + k] = a.col[c].w[k];
        };
        // on the way from A to A^(2^67): the offset matrices A^(2^i), i < 32 (slots behind the subsequence matrices)
        for (int i = 0; i < 67; ++i) {
            if (i < XORWOW_OFFSET_BITS)
                store(XORWOW_JUMP_BITS + i);
            xw_square(a);
        }
        for (int i = 0; i < XORWOW_JUMP_BITS; ++i) {
            store(i);
            xw_square(a);
        }
        return t;
    }();
    return table;
}
}  // namespace

static int xorwow_jump_ready(mc_context *c)
{
    if (!c->d_xorwow_jump) {
        const std::vector<uint32_t> &t = xorwow_jump_table();
        HIPCHK(hipMalloc(&c->d_xorwow_jump, t.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(c->d_xorwow_jump, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return MC_OK;
}

// Start states of `lanes` XORWOW lanes for (seed, base) into `states` (device), enqueued on st.
static int xorwow_fill(mc_context *c, uint64_t seed, uint64_t base, uint32_t lanes, uint32_t *states, hipStream_t st)
{
    if (base + lanes > (1ull << XORWOW_JUMP_BITS) || base + lanes < base)
        return fail(MC_ERR_INVALID, "XORWOW: subsequence numbers must stay below 2^%d", XORWOW_JUMP_BITS);
    if (int rc = xorwow_jump_ready(c)) return rc;
    xorwow_init_kernel<<<(lanes + 255) / 256, 256, 0, st>>>(c->d_xorwow_jump, seed, base, lanes, states);
    HIPCHK(hipGetLastError());
    return MC_OK;
}

// The context's lane states for this call's seed (all blocks * GROUP lanes: a smaller grid uses a prefix)
static int xorwow_ready(mc_context *c, uint64_t seed, hipStream_t st)
{
    const uint32_t lanes = (uint32_t)c->blocks * GROUP;
    if (c->d_xorwow && c->xorwow_valid && c->xorwow_seed == seed && c->xorwow_state_base == c->xorwow_base)
        return MC_OK;
    StageTimer stage(&c->acc_setup_ms);   // the reference's randomSetup (dp/MonteCarloKernel.cu:285-290, "RNG done" :317-323) -- here once per seed
    if (!c->d_xorwow)
        HIPCHK(hipMalloc(&c->d_xorwow, sizeof(uint32_t) * 6 * (size_t)lanes));
    if (int rc = xorwow_fill(c, seed, c->xorwow_base, lanes, c->d_xorwow, st)) return rc;
    c->xorwow_valid = true, c->xorwow_seed = seed, c->xorwow_state_base = c->xorwow_base;
    return c->stats_timed_call ? setup_settle(c, st) : MC_OK;
}

static int xorwow_one_segment(const std::vector<Segment> &segs)
{
    if (segs.size() != 1)
        return fail(MC_ERR_UNSUPPORTED, "XORWOW generator: one call is one launch (at most 2^31 units, not across a multiple of "
                                        "2^32): a lane's sequence restarts with every launch");
    return MC_OK;
}

extern "C" int mc_context_set_generator(mc_context *c, int generator, uint64_t subsequence_base)
{
    if (!c || (generator != MC_RNG_PHILOX && generator != MC_RNG_XORWOW))
        return fail(MC_ERR_INVALID, "mc_context_set_generator: bad argument");
    c->rng = generator;
    c->xorwow_base = subsequence_base;
    return MC_OK;
}

extern "C" int mc_xorwow_words(mc_context *c, uint64_t seed, uint64_t first_subsequence, uint32_t n_subsequences,
                               uint32_t words_each, uint32_t *h_out)
{
    if (!c || !h_out || n_subsequences == 0 || words_each == 0 || (uint64_t)n_subsequences * words_each > (1u << 26))
        return fail(MC_ERR_INVALID, "mc_xorwow_words: bad argument");
    HIPCHK(hipSetDevice(c->device));
    uint32_t *states = nullptr, *out = nullptr;
    HIPCHK(hipMalloc(&states, sizeof(uint32_t) * 6 * (size_t)n_subsequences));
    HIPCHK(hipMalloc(&out, sizeof(uint32_t) * (size_t)n_subsequences * words_each));
    int rc = xorwow_fill(c, seed, first_subsequence, n_subsequences, states, c->stream);
    if (rc == MC_OK) {
        xorwow_words_kernel<<<(n_subsequences + 255) / 256, 256, 0, c->stream>>>(states, n_subsequences, words_each, out);
        if (hipMemcpyAsync(h_out, out, sizeof(uint32_t) * (size_t)n_subsequences * words_each, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = fail(MC_ERR_HIP, "mc_xorwow_words: copy failed");
    }
    (void)hipFree(states);
    (void)hipFree(out);
    return rc;
}

// NULL is the HIP null stream, as in every HIP API (the context's own stream: mc_context_stream)
static hipStream_t pick_stream(mc_context *, void *stream) { return (hipStream_t)stream; }

static int check_common(mc_context *c, const void *opt, uint64_t first, uint64_t n, const void *dst)
{
    if (!c) return fail(MC_ERR_INVALID, "NULL context");
    if (!opt) return fail(MC_ERR_INVALID, "NULL option");
    if (!dst) return fail(MC_ERR_INVALID, "NULL output pointer");
    if (n == 0) return fail(MC_ERR_INVALID, "n_paths == 0");
    if (first + n < first) return fail(MC_ERR_INVALID, "path range overflows 64 bits");
    if (n > (1ull << 52)) return fail(MC_ERR_INVALID, "n_paths > 2^52 is not exactly representable in the fp64 triple");
    return MC_OK;
}

static int ensure_out(mc_context *c, size_t bytes)
{
    if (c->d_out_bytes >= bytes)
        return MC_OK;
    if (c->d_out) {
        if (int rc = quiesce(c)) return rc;
        HIPCHK(hipFree(c->d_out));
        c->d_out = nullptr;
        c->d_out_bytes = 0;
    }
    HIPCHK(hipMalloc(&c->d_out, bytes));
    c->d_out_bytes = bytes;
    return MC_OK;
}

static constexpr uint64_t MAX_DUMP_PATHS = 1ull << 26;

// The context's one per-call constant table (CVA per-date rows, or a generic basket's folded
// constants): rebuilt and re-uploaded only when the inputs (the key) change.
static int upload_table(mc_context *c, hipStream_t st, const std::vector<char> &key, const void *data, size_t bytes)
{
    if (key != c->table_key) {
        StageTimer stage(&c->acc_table_ms);
        if (bytes > c->table_bytes) {
            HIPCHK(hipStreamSynchronize(st));
            if (c->table_stream && c->table_stream != st) HIPCHK(hipStreamSynchronize(c->table_stream));
            if (int rc = quiesce(c)) return rc;   // a launch on another caller stream may still read the old table
            if (c->d_table) HIPCHK(hipFree(c->d_table));
            if (c->h_table) HIPCHK(hipHostFree(c->h_table));
            c->table_bytes = bytes < 4096 ? 4096 : bytes;
            HIPCHK(hipMalloc(&c->d_table, c->table_bytes));
            HIPCHK(hipHostMalloc(&c->h_table, c->table_bytes, hipHostMallocDefault));
        } else {
            HIPCHK(hipEventSynchronize(c->table_copied));  // previous upload has left the staging buffer
        }
        if (bytes) {
            memcpy(c->h_table, data, bytes);
            HIPCHK(hipMemcpyAsync(c->d_table, c->h_table, bytes, hipMemcpyHostToDevice, st));
        }
        HIPCHK(hipEventRecord(c->table_copied, st));
        c->table_stream = st;
        c->table_key = key;
    } else if (st != c->table_stream) {
        // cached table uploaded on another stream: order this stream behind that upload
        HIPCHK(hipStreamWaitEvent(st, c->table_copied, 0));
    }
    return MC_OK;
}

// ---------------------------------------------------------------------------------------
// vanilla
// ---------------------------------------------------------------------------------------
static inline bool finite_pos(double x) { return std::isfinite(x) && x > 0; }

// The fp64 kernels' exp is table-driven with a rounding trick that needs |x| < EXP_F64_ARG_LIMIT = 5e6 (mc_math_f64.hpp); any model
// whose exponent can leave the range of a double (|x| > ~700 at the generator's largest normal) is rejected up
// front instead of pricing garbage.  The 52-bit uniform bottoms out at 2^-53: |z| <= sqrt(2 * 53 ln 2) = 8.572.
constexpr double Z_MAX_F64 = 8.58;
static inline bool exponent_in_range(double bound) { return std::isfinite(bound) && bound < 700.0; }

template <class Real> struct VanillaTraits;
template <> struct VanillaTraits<float> {
    using Opt = VanillaF32;
    using In = mc_option_f32;
    static int launch_hot(mc_context *c, ProfileScope &prof, bool anti, const Opt &k, const Work &w, const Tail &tail, int grid,
                          hipStream_t st)
    {
        return with_anti_gen<GEN_PHILOX | GEN_XORWOW | GEN_EXTERNAL>(anti, gen_of(c, w, 4), [&](auto a, auto tag) {
            launch_sim(prof, vanilla_f32_kernel<decltype(a)::value, typename decltype(tag)::type>, grid, st, tail, k, w);
        });
    }
    static int prepare(const In &o, Opt &k_, double &scale1, double &scale2)
    {
        // the reference's max(S_T - K, 0) (dp/MonteCarloKernel.cu:70) is defined for any strike: K <= 0 is accepted
        if (!finite_pos(o.s) || !std::isfinite(o.k) || !(o.v >= 0) || !(o.t >= 0) || !std::isfinite(o.r))
            return fail(MC_ERR_INVALID, "vanilla: need s>0, finite k, v>=0, t>=0, finite r");
        const double log2e = 1.4426950408889634074;
        const double drift = ((double)o.r - 0.5 * (double)o.v * (double)o.v) * (double)o.t;
        const double vol = (double)o.v * std::sqrt((double)o.t);
        const double a2 = drift * log2e, b2 = vol * log2e;
        // Box-Muller on 32-bit uniforms cannot exceed |z| = sqrt(2 * 33 ln 2) = 6.764 (u >= 2^-33);
        // k makes 2^(a2 - k + b2 z) - K/(S 2^k) <= 1 for all of them, so the device's [0,1] clamp is exact.
        // A negative strike RAISES the payoff (S_T + |K|): it enters the bound like the basket's does (basket_launch_n).
        const double zmax = 6.77;
        const double k = std::ceil(std::log2(std::exp2(a2 + b2 * zmax) + ((double)o.k < 0 ? -(double)o.k / (double)o.s : 0.0)));
        if (!(std::fabs(k) < 100))
            return fail(MC_ERR_INVALID, "vanilla f32: drift/volatility out of the float range (k=%g)", k);
        const double two_k = std::ldexp(1.0, (int)k);
        k_.a2k = (float)(a2 - k);
        k_.radius2 = (float)(-2.0 * 0.69314718055994530942 * b2 * b2);
        k_.b2 = (float)b2;
        k_.kappa_k = (float)((double)o.k / (double)o.s / two_k);
        scale1 = (double)o.s * two_k;
        scale2 = scale1 * scale1;
        return MC_OK;
    }
};
template <> struct VanillaTraits<double> {
    using Opt = VanillaF64;
    using In = mc_option_f64;
    static int launch_hot(mc_context *c, ProfileScope &prof, bool anti, const Opt &k, const Work &w, const Tail &tail, int grid,
                          hipStream_t st)
    {
        return with_anti_gen<GEN_PHILOX | GEN_XORWOW | GEN_F32N | GEN_EXTERNAL>(anti, gen_of(c, w, 8), [&](auto a, auto tag) {
            launch_sim(prof, vanilla_kernel<Opt, double, decltype(a)::value, typename decltype(tag)::type>, grid, st, tail, k, w);
        });
    }
    static int prepare(const In &o, Opt &k, double &scale1, double &scale2)
    {
        if (!finite_pos(o.s) || !std::isfinite(o.k) || !(o.v >= 0) || !(o.t >= 0) || !std::isfinite(o.r))
            return fail(MC_ERR_INVALID, "vanilla: need s>0, finite k, v>=0, t>=0, finite r");
        k.drift = (o.r - 0.5 * o.v * o.v) * o.t;
        k.vol = o.v * std::sqrt(o.t);
        k.strike = o.k;
        k.spot = o.s;
        scale1 = scale2 = 1.0;
        if (!exponent_in_range(std::fabs(k.drift) + k.vol * Z_MAX_F64))
            return fail(MC_ERR_INVALID, "vanilla: (r - v^2/2) t and v sqrt(t) put the terminal spot outside the range of a double");
        return MC_OK;
    }
};

// The masked (generic) vanilla kernel of a launch: every generator, both estimators
template <class Real>
static int launch_vanilla_masked(mc_context *c, bool anti, const typename VanillaTraits<Real>::Opt &k, const Work &w, const Tail &t,
                                 int grid, hipStream_t st, Real *out, Real out_scale)
{
    using Opt = typename VanillaTraits<Real>::Opt;
    constexpr unsigned ALLOW = sizeof(Real) == 8 ? (GEN_PHILOX | GEN_XORWOW | GEN_F32N | GEN_EXTERNAL) : (GEN_PHILOX | GEN_XORWOW | GEN_EXTERNAL);
    return with_anti_gen<ALLOW>(anti, gen_of(c, w, sizeof(Real)), [&](auto a, auto tag) {
        vanilla_masked_kernel<Opt, Real, decltype(a)::value, typename decltype(tag)::type><<<grid, GROUP, 0, st>>>(t, k, w, out, out_scale);
    });
}

// Enqueue simulation + reduction of paths [first, first+n).  out != nullptr additionally stores
// every payoff (device buffer of n Reals) and forces the masked kernel for all units.
template <class Real>
static int vanilla_enqueue(mc_context *c, const typename VanillaTraits<Real>::In *opt, uint64_t seed,
                           uint64_t first, uint64_t n, double *d_triple, hipStream_t st, Real *out)
{
    using T = VanillaTraits<Real>;
    const uint64_t NPB = npb_of<Real>(c);   // paths per unit
    typename T::Opt k;
    double scale1, scale2;
    if (int rc = T::prepare(*opt, k, scale1, scale2))
        return rc;
    const bool anti = c->antithetic;
    if (anti && sizeof(Real) == 4) {  // f32 kernels hand back the SUM of the two mirrored payoffs
        scale1 *= 0.5;
        scale2 *= 0.25;
    }
    if (int rc = begin_call(c, st)) return rc;
    const uint64_t end = first + n;
    if (c->rng == MC_RNG_XORWOW && !c->ext) {
        if (sizeof(Real) == 8 && c->normals_f32)
            return fail(MC_ERR_UNSUPPORTED, "fp32 normals in the fp64 kernels are implemented for the Philox generator");
        // one masked launch over every unit the range touches (no separate edge launches: a lane's sequence would
        // restart in them); the masked kernel is the generic form
        std::vector<Segment> one;
        const uint64_t u0 = first / NPB, u1 = (end + NPB - 1) / NPB;
        if (int rc = plan_segments(u0, u1 - u0, one)) return rc;
        if (int rc = xorwow_one_segment(one)) return rc;
        if (int rc = xorwow_ready(c, seed, st)) return rc;
        const Work w = context_work(c, seed, one[0], first, end);
        const int g = grid_for(c->blocks, one[0].count);
        Tail t = make_tail(c, g, scale1, scale2, n, d_triple);
        if (!out && first % NPB == 0 && end % NPB == 0) {   // whole units only (what the legacy symbols ask for): the hot kernel,
            ProfileScope prof(c);                           // same lanes, same draws, same sums up to fp32 partial-sum order
            if (int rc = T::launch_hot(c, prof, anti, k, w, t, g, st)) return rc;
        } else if (int rc = launch_vanilla_masked<Real>(c, anti, k, w, t, g, st, out, (Real)scale1))
            return rc;
        return finish_call(c, t, g, st);
    }
    std::vector<Segment> segs;
    ProfileScope prof(c);
    // plan first: the launches of the call and their grids (every launch needs the call's total pair count)
    const uint64_t head = first / NPB, tail_unit = end / NPB;
    bool has_head = false, has_tail = false;
    if (out) {
        const uint64_t u0 = first / NPB, u1 = (end + NPB - 1) / NPB;
        if (int rc = plan_segments(u0, u1 - u0, segs)) return rc;
    } else {
        const uint64_t u_full0 = (first + NPB - 1) / NPB, u_full1 = end / NPB;
        if (u_full1 > u_full0)
            if (int rc = plan_segments(u_full0, u_full1 - u_full0, segs)) return rc;
        // partial units at the edges of the range (at most two single-unit launches)
        has_head = (first % NPB) != 0;
        has_tail = (end % NPB) != 0 && !(has_head && tail_unit == head);
    }
    if (c->ext && (segs.size() > 1 || first != 0))
        return fail(MC_ERR_INVALID, "external normals: one segment starting at path 0");
    const int scale = (sizeof(Real) == 8 && !c->ext) ? GRID_SCALE_VANILLA_F64 : 2;
    const auto grid = [&](uint32_t units) { return out ? grid_for(c->blocks, units) : grid_for_vanilla(c->blocks, c->compute_units, units, scale); };
    int total = (has_head ? 1 : 0) + (has_tail ? 1 : 0);
    for (const Segment &s : segs)
        total += grid(s.count);
    Tail t = make_tail(c, total, scale1, scale2, n, d_triple);
    int slot = 0;
    for (const Segment &s : segs) {
        const Work w = context_work(c, seed, s, first, end);
        const int g = grid(s.count);
        t.slot_base = t.ticket_base = (uint32_t)slot;
        if (out) {
            if (int rc = launch_vanilla_masked<Real>(c, anti, k, w, t, g, st, out, (Real)scale1)) return rc;
        } else if (int rc = T::launch_hot(c, prof, anti, k, w, t, g, st))
            return rc;
        slot += g;
    }
    for (int e = 0; e < 2; ++e) {
        if (!(e == 0 ? has_head : has_tail))
            continue;
        Work w = context_work(c, seed, Segment{e == 0 ? head : tail_unit, 1u}, first, end);
        if (w.ext)   // the external array is indexed from the launch's first unit
            w.ext = static_cast<const Real *>(w.ext) + w.unit_lo * (uint64_t)w.ext_per_unit;
        t.slot_base = t.ticket_base = (uint32_t)slot;
        if (int rc = launch_vanilla_masked<Real>(c, anti, k, w, t, 1, st, (Real *)nullptr, (Real)1)) return rc;
        slot += 1;
    }
    return finish_call(c, t, total, st);
}

// ---------------------------------------------------------------------------------------
// vanilla with pathwise Greeks (price, delta, vega)
// ---------------------------------------------------------------------------------------
static void greeks_prepare(const mc_option_f32 &o, GreeksF32 &k)
{
    const double log2e = 1.4426950408889634074;
    k.drift2 = (float)((((double)o.r - 0.5 * (double)o.v * (double)o.v) * (double)o.t) * log2e);
    k.vol2 = (float)((double)o.v * std::sqrt((double)o.t) * log2e);
    k.spot = o.s, k.strike = o.k;
    k.sqrt_t = (float)std::sqrt((double)o.t);
    k.sigma_t = (float)((double)o.v * (double)o.t);
    k.lr_delta = (float)(1.0 / ((double)o.s * (double)o.v * std::sqrt((double)o.t)));
    k.inv_sigma = (float)(1.0 / (double)o.v);
}
static void greeks_prepare(const mc_option_f64 &o, GreeksF64 &k)
{
    k.drift = (o.r - 0.5 * o.v * o.v) * o.t;
    k.vol = o.v * std::sqrt(o.t);
    k.spot = o.s, k.strike = o.k;
    k.sqrt_t = std::sqrt(o.t);
    k.sigma_t = o.v * o.t;
    k.lr_delta = 1.0 / (o.s * o.v * std::sqrt(o.t));
    k.inv_sigma = 1.0 / o.v;
}

// Multi-plane calls (Greeks): `planes` (sum, sum2) pairs per workgroup in a buffer of their own, one triple per plane.
// Synchronous: enqueue on the context stream, wait, close every plane with `discount`.
static int ensure_planes(mc_context *c, int planes)
{
    if (planes <= c->g_planes)
        return MC_OK;
    if (int rc = quiesce(c)) return rc;
    if (c->g_pairs) HIPCHK(hipFree(c->g_pairs));
    if (c->g_triples) HIPCHK(hipFree(c->g_triples));
    c->g_pairs = nullptr, c->g_triples = nullptr, c->g_planes = 0;
    HIPCHK(hipMalloc(&c->g_pairs, sizeof(double2) * (size_t)planes * (2 * (size_t)c->blocks + 2)));
    HIPCHK(hipMalloc(&c->g_triples, sizeof(double) * 3 * (size_t)planes));
    c->g_planes = planes;
    return MC_OK;
}

// launch(tail, segment index, segment, grid) enqueues one segment; grid_y = workgroups per x position (arrivals)
// real_bytes = sizeof the simulation type: the fp32-normals mode concerns the fp64 kernels only (mc_mi355x.h: "no effect on
// the _f32 entry points"), so only the fp64 Greeks refuse it
template <class Launch>
static int planes_run(mc_context *c, size_t real_bytes, int planes, int grid_y, uint64_t unit0, uint64_t n_units, uint64_t n, double discount,
                      mc_result **out, Launch launch)
{
    const auto wall0 = std::chrono::steady_clock::now();
    hipStream_t st = c->stream;
    if (c->rng != MC_RNG_PHILOX || (real_bytes == 8 && c->normals_f32))
        return fail(MC_ERR_UNSUPPORTED, "greeks: implemented for the Philox generator with native normals only");
    if (int rc = begin_call(c, st)) return rc;
    if (int rc = ensure_planes(c, planes)) return rc;
    std::vector<Segment> segs;
    if (int rc = plan_segments(unit0, n_units, segs)) return rc;
    if (segs.size() > 2)   // a plane holds the pairs of two full launches
        return fail(MC_ERR_INVALID, "greeks: path range too large for one call; split it");
    HIPCHK(hipMemsetAsync(c->g_triples, 0xFF, sizeof(double) * 3 * (size_t)planes, st));   // poison: see run_sync
    HIPCHK(hipEventRecord(c->ev0, st));
    int pairs = 0;
    for (const Segment &s : segs)
        pairs += grid_for(c->blocks, s.count);
    Tail t = make_tail(c, pairs, 1.0, 1.0, n, c->g_triples, planes, 2 * c->blocks + 2);
    t.partials = c->g_pairs;
    if (c->fused)
        t.total = (uint32_t)(pairs * grid_y);
    int slot = 0;
    for (const Segment &s : segs) {
        const int g = grid_for(c->blocks, s.count);
        t.slot_base = (uint32_t)slot;
        t.ticket_base = (uint32_t)(slot * grid_y);
        launch(t, s, g, st);
        slot += g;
    }
    if (int rc = finish_call(c, t, pairs, st)) return rc;
    HIPCHK(hipEventRecord(c->ev1, st));
    std::vector<double> h(3 * (size_t)planes);
    HIPCHK(hipMemcpyAsync(h.data(), c->g_triples, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    for (int q = 0; q < planes; ++q) {
        mc_result *r = out[q];
        if (!(h[3 * q + 2] == (double)n))
            return fail(MC_ERR_HIP, "device returned n=%g, expected %llu: the call's final reduction did not complete", h[3 * q + 2],
                        (unsigned long long)n);
        r->sum = h[3 * q], r->sum2 = h[3 * q + 1], r->n = (uint64_t)h[3 * q + 2], r->kernel_ms = ms, r->wall_ms = wall;
        if (r->n != n)
            return fail(MC_ERR_HIP, "device returned n=%llu, expected %llu", (unsigned long long)r->n, (unsigned long long)n);
        mc_closing(r->sum, r->sum2, r->n, discount, &r->expected, &r->confidence);
    }
    return MC_OK;
}

template <class Real, class In, class Opt, bool LR>
static int greeks_run(mc_context *c, const In *o, uint64_t seed, uint64_t first, uint64_t n, mc_vanilla_greeks *out)
{
    if (int rc = check_common(c, o, first, n, out)) return rc;
    if (!finite_pos(o->s) || !std::isfinite((double)o->k) || !(o->v >= 0) || !(o->t >= 0) || !std::isfinite((double)o->r))
        return fail(MC_ERR_INVALID, "vanilla: need s>0, finite k, v>=0, t>=0, finite r");
    if (LR && !((double)o->v > 0 && (double)o->t > 0))
        return fail(MC_ERR_INVALID, "likelihood-ratio greeks: need v>0 and t>0 (the scores divide by sigma sqrt t)");
    if (c->antithetic)
        return fail(MC_ERR_UNSUPPORTED, "greeks: only the plain estimator is implemented");
    {   // the pricing paths' exponent-range guards (VanillaTraits<>::prepare): refuse inputs whose terminal spot
        // overflows the simulation type instead of returning inf/NaN sums with MC_OK
        const double drift = ((double)o->r - 0.5 * (double)o->v * (double)o->v) * (double)o->t;
        const double vol = (double)o->v * std::sqrt((double)o->t);
        const bool ok = sizeof(Real) == 4 ? std::fabs(std::ceil((drift + vol * 6.77) * 1.4426950408889634074)) < 100 &&
                                                std::fabs(drift - vol * 6.77) * 1.4426950408889634074 < 100
                                          : exponent_in_range(std::fabs(drift) + vol * Z_MAX_F64);
        if (!ok)
            return fail(MC_ERR_INVALID, "greeks: (r - v^2/2) t and v sqrt(t) put the terminal spot outside the range of the simulation type");
    }
    constexpr uint64_t NPB = GenPhilox::npb<Real>();
    Opt k;
    greeks_prepare(*o, k);
    const uint64_t end = first + n, u0 = first / NPB, u1 = (end + NPB - 1) / NPB;
    mc_result *r[3] = {&out->price, &out->delta, &out->vega};
    return planes_run(c, sizeof(Real), 3, 1, u0, u1 - u0, n, std::exp(-(double)o->r * (double)o->t), r,
                      [&](const Tail &t, const Segment &s, int g, hipStream_t st) {
                          vanilla_greeks_kernel<Opt, Real, LR><<<g, GROUP, 0, st>>>(t, k, make_work(seed, s, first, end));
                      });
}

extern "C" int mc_vanilla_greeks_run_f32(mc_context *c, const mc_option_f32 *o, uint64_t seed, uint64_t first, uint64_t n,
                                         mc_vanilla_greeks *out)
{
    return greeks_run<float, mc_option_f32, GreeksF32, false>(c, o, seed, first, n, out);
}
extern "C" int mc_vanilla_greeks_run_f64(mc_context *c, const mc_option_f64 *o, uint64_t seed, uint64_t first, uint64_t n,
                                         mc_vanilla_greeks *out)
{
    return greeks_run<double, mc_option_f64, GreeksF64, false>(c, o, seed, first, n, out);
}
extern "C" int mc_vanilla_greeks_lr_run_f32(mc_context *c, const mc_option_f32 *o, uint64_t seed, uint64_t first, uint64_t n,
                                            mc_vanilla_greeks *out)
{
    return greeks_run<float, mc_option_f32, GreeksF32, true>(c, o, seed, first, n, out);
}
extern "C" int mc_vanilla_greeks_lr_run_f64(mc_context *c, const mc_option_f64 *o, uint64_t seed, uint64_t first, uint64_t n,
                                            mc_vanilla_greeks *out)
{
    return greeks_run<double, mc_option_f64, GreeksF64, true>(c, o, seed, first, n, out);
}

// ---------------------------------------------------------------------------------------
// basket
// ---------------------------------------------------------------------------------------
template <class Real> struct BasketIn;
template <> struct BasketIn<float> { using type = mc_basket_f32; };
template <> struct BasketIn<double> { using type = mc_basket_f64; };
template <class Real> static constexpr double exp_scale() { return 1.0; }
template <> constexpr double exp_scale<float>() { return 1.4426950408889634074; }  // log2(e): E = 2^x

// the kernel of each precision: f32 = two paths per lane in packed halves, f64 = one path per lane.  Generators: Philox;
// fp32 normals widened (fp64, up to the default static limit of 8 assets); external normals for the sizes the bridge
// tests use (3 = the reference's N, 4 = BASELINE C3).
template <int NA>
static int basket_launch_kernel(mc_context *c, ProfileScope &prof, bool anti, const BasketArgs<float, NA> &k, const Work &w,
                                const Tail &tail, float *out, double scale, int grid, hipStream_t st)
{
    constexpr unsigned ALLOW = GEN_PHILOX | ((NA == 3 || NA == 4) ? GEN_EXTERNAL : 0);
    return with_anti_gen<ALLOW>(anti, gen_of(c, w, 4), [&](auto a, auto tag) {
        launch_sim(prof, basket_f32_kernel<NA, decltype(a)::value, typename decltype(tag)::type>, grid, st, tail, k, w, out, (float)scale);
    });
}
template <int NA>
static int basket_launch_kernel(mc_context *c, ProfileScope &prof, bool anti, const BasketArgs<double, NA> &k, const Work &w,
                                const Tail &tail, double *out, double, int grid, hipStream_t st)
{
    constexpr unsigned ALLOW = GEN_PHILOX | (NA <= 8 ? GEN_F32N : 0) | ((NA == 3 || NA == 4) ? GEN_EXTERNAL : 0);
    return with_anti_gen<ALLOW>(anti, gen_of(c, w, 8), [&](auto a, auto tag) {
        launch_sim(prof, basket_kernel<double, NA, decltype(a)::value, typename decltype(tag)::type>, grid, st, tail, k, w, out);
    });
}

// The volatility that multiplies the correlated normals.  MC_FROM_NORMALS_NO_VOL (bridge tests only) replaces it by 1: the
// reference's dp CPU path forms the diffusion without it (dp/MonteCarloHost.c:180), and its goldens can
// only be met on that model.  The drift keeps the true volatility, as it does there.
static double diffusion_vol(const mc_context *c, double v) { return (c->ext && (c->ext_flags & MC_FROM_NORMALS_NO_VOL)) ? 1.0 : v; }

// Folds a basket of o.n <= NA assets into the kernel-argument constants of size NA (rows beyond o.n: zero factor, base and
// weight -- they add exactly 0 to the basket).  out_scale = what the kernel's per-path values are multiplied by to give
// currency units (fp32: the exact power of two of the [0,1] rescale; 1/2 of it under antithetic variates, whose fp32
// kernels return the SUM of the two mirrored payoffs).
template <class Real, int NA>
static int basket_fold(mc_context *c, const typename BasketIn<Real>::type &o, BasketArgs<Real, NA> &k, double &out_scale)
{
    const int n = o.n;
    double scale = 1.0;
    constexpr bool is_f32 = sizeof(Real) == 4;
    const double sc = exp_scale<Real>();
    const double sqrt_t = std::sqrt((double)o.t);
    double m[NA][NA], base[NA], coef[NA];
    for (int a = 0; a < NA; ++a) {
        base[a] = coef[a] = 0;
        for (int b = 0; b < NA; ++b)
            m[a][b] = 0;
    }
    for (int a = 0; a < n; ++a) {
        const double va = (double)o.v[a];
        const double vd = diffusion_vol(c, va);   // va, except under the bridge tests' reference-CPU-bug switch
        for (int b = 0; b <= a; ++b)
            m[a][b] = vd * sqrt_t * (double)o.p[a * n + b] * sc;
        base[a] = (((double)o.r - 0.5 * va * va) * (double)o.t + vd * sqrt_t * (double)o.d[a]) * sc;
        coef[a] = (double)o.w[a] * (double)o.s[a];
        double bound = std::fabs(base[a]);
        for (int b = 0; b <= a; ++b)
            bound += std::fabs(m[a][b]) * Z_MAX_F64;
        if (!exponent_in_range(bound / sc))
            return fail(MC_ERR_INVALID, "basket: asset %d's drift and volatility put its terminal price outside the range of a double", a);
    }
    scale = 1.0;
    if (is_f32) {
        // |z| < 6.77 for every normal the f32 generator can produce: bound the basket and rescale by
        // an exact power of two so that the device's [0,1] clamp is the payoff's max(.,0)
        const double zmax = 6.77;
        double bound = 0;
        for (int a = 0; a < n; ++a) {
            double x = base[a];
            for (int b = 0; b <= a; ++b)
                x += std::fabs(m[a][b]) * zmax;
            bound += std::fabs(coef[a]) * std::exp2(x);
        }
        // a negative strike RAISES the payoff: basket - K <= bound + |K| must still fit the [0,1] clamp
        if ((double)o.k < 0)
            bound += -(double)o.k;
        const double kk = bound > 0 ? std::ceil(std::log2(bound)) : 0;
        if (!(std::fabs(kk) < 100))
            return fail(MC_ERR_INVALID, "basket f32: inputs out of the float range (scale 2^%g)", kk);
        scale = std::ldexp(1.0, (int)kk);
    }
    out_scale = (is_f32 && c->antithetic) ? 0.5 * scale : scale;  // f32 anti: kernel returns the SUM
    for (int a = 0; a < NA; ++a) {
        for (int b = 0; b <= a; ++b)
            k.m[a * (a + 1) / 2 + b] = (Real)m[a][b];
        k.base[a] = (Real)base[a];
        k.coef[a] = (Real)(coef[a] / scale);
        k.wg[a] = 0;
    }
    k.strike = (Real)((double)o.k / scale);
    k.cg = 0;
    k.cv = 0;
    if (c->control) {
        double cv_mean;
        if (int rc = control_mean(o, &cv_mean)) return rc;  // validates w > 0, s > 0, k > 0
        double W = 0, cg = 0;
        for (int a = 0; a < n; ++a)
            W += (double)o.w[a];
        for (int a = 0; a < n; ++a) {
            k.wg[a] = (Real)((double)o.w[a] / W);
            cg += (double)o.w[a] / W * std::log((double)o.s[a]);
        }
        // ln G in the kernel's exponent units (log2 in f32), minus the power-of-two rescale
        k.cg = (Real)((cg + std::log(W)) * sc - (is_f32 ? std::log2(scale) : 0.0));
        k.cv = 1;
    }
    return MC_OK;
}

template <class Real, int NA>
static int basket_launch_n(mc_context *c, ProfileScope &prof, const typename BasketIn<Real>::type &o, uint64_t seed,
                           const std::vector<Segment> &segs, hipStream_t st, Real *out, uint64_t n_paths, double *d_triple)
{
    constexpr bool is_f32 = sizeof(Real) == 4;
    BasketArgs<Real, NA> k;
    double out_scale = 1.0;
    if (int rc = basket_fold<Real, NA>(c, o, k, out_scale)) return rc;
    int total = 0, slot = 0;
    for (const Segment &s : segs)
        total += grid_for(c->blocks, is_f32 ? (s.count + 1) / 2 : s.count);
    Tail t = make_tail(c, total, out_scale, out_scale * out_scale, n_paths, d_triple);
    uint64_t done = 0;
    for (const Segment &s : segs) {
        const Work w = context_work(c, seed, s, 0, 0);
        const int g = grid_for(c->blocks, is_f32 ? (s.count + 1) / 2 : s.count);
        t.slot_base = t.ticket_base = (uint32_t)slot;
        if (int rc = basket_launch_kernel<NA>(c, prof, c->antithetic, k, w, t, out ? out + done : (Real *)nullptr, out_scale, g, st))
            return rc;
        slot += g;
        done += s.count;
    }
    return finish_call(c, t, total, st);
}

// The table-driven families: fold the constants exactly like basket_launch_n (no power-of-two rescale:
// these kernels take the plain max), lay them out in 4 x 4 tiles, park them in the context's table
// buffer and run the tiled kernel of the size, or the generic one beyond 32 assets.
template <class Real>
static int basket_launch_dyn(mc_context *c, ProfileScope &prof, const typename BasketIn<Real>::type &o, uint64_t seed,
                             const std::vector<Segment> &segs, hipStream_t st, Real *out, uint64_t n_paths, double *d_triple)
{
    // layout of BasketDyn::consts: 4 x 4 tiles of the folded lower-triangular matrix, block-row by block-row
    // (tile (A, c4) = 16 reals, index 4 j + r = m[4A + r][4 c4 + j], zero outside the triangle / beyond n),
    // then base, coef, wg padded with zeros to 4 nb entries
    const int n = o.n, nb = (n + 3) / 4, np = 4 * nb;
    const size_t n_tiles = (size_t)8 * nb * (nb + 1);
    const double sc = exp_scale<Real>();
    const double sqrt_t = std::sqrt((double)o.t);
    // fp64, 13..16 assets: the same matrix once more as the A operands of the matrix-core kernel (basket_mfma_f64_kernel:
    // a4[s][lane] = M[lane & 15][k(s, lane >> 4)], k(s, q) = 2 q + (s & 1) + 8 (s >> 1))
    const bool with_a4 = sizeof(Real) == 8 && np == 16;
    std::vector<Real> host(n_tiles + 3 * (size_t)np + (with_a4 ? 256 : 0), (Real)0);
    if (with_a4)
        for (int s4 = 0; s4 < 4; ++s4)
            for (int lane = 0; lane < 64; ++lane) {
                const int a = lane & 15, b = 2 * (lane >> 4) + (s4 & 1) + 8 * (s4 >> 1);
                if (a < n && b <= a)
                    host[n_tiles + 3 * (size_t)np + 64 * s4 + lane] = (Real)(diffusion_vol(c, (double)o.v[a]) * sqrt_t * (double)o.p[a * n + b] * sc);
            }
    size_t t = 0;
    for (int A = 0; A < nb; ++A)
        for (int c4 = 0; c4 <= A; ++c4, t += 16)
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 4; ++r) {
                    const int a = 4 * A + r, b = 4 * c4 + j;
                    if (a < n && b <= a)
                        host[t + 4 * j + r] = (Real)(diffusion_vol(c, (double)o.v[a]) * sqrt_t * (double)o.p[a * n + b] * sc);
                }
    for (int a = 0; a < n; ++a) {
        const double va = (double)o.v[a], vd = diffusion_vol(c, va);
        host[n_tiles + a] = (Real)((((double)o.r - 0.5 * va * va) * (double)o.t + vd * sqrt_t * (double)o.d[a]) * sc);
        host[n_tiles + np + a] = (Real)((double)o.w[a] * (double)o.s[a]);
        double bound = std::fabs((double)host[n_tiles + a]);
        for (int b = 0; b <= a; ++b)
            bound += std::fabs(vd * sqrt_t * (double)o.p[a * n + b] * sc) * Z_MAX_F64;
        if (!exponent_in_range(bound / sc))
            return fail(MC_ERR_INVALID, "basket: asset %d's drift and volatility put its terminal price outside the range of a double", a);
    }
    double cg_dyn = 0;
    if (c->control) {
        double cv_mean, W = 0;
        if (int rc = control_mean(o, &cv_mean)) return rc;
        for (int a = 0; a < n; ++a)
            W += (double)o.w[a];
        for (int a = 0; a < n; ++a) {
            host[n_tiles + 2 * np + a] = (Real)((double)o.w[a] / W);
            cg_dyn += (double)o.w[a] / W * std::log((double)o.s[a]);
        }
        cg_dyn = (cg_dyn + std::log(W)) * sc;
    }
    const size_t bytes = host.size() * sizeof(Real);
    std::vector<char> key(bytes + 2);
    memcpy(key.data(), host.data(), bytes);
    key[bytes] = (char)sizeof(Real);
    key[bytes + 1] = 'B';
    if (int rc = upload_table(c, st, key, host.data(), bytes)) return rc;
    BasketDyn<Real> k;
    k.consts = (const Real *)c->d_table;
    k.n = n;
    k.strike = o.k;
    k.cg = (Real)cg_dyn;
    k.cv = c->control ? 1 : 0;
    // Pick the kernel.  `gen` = the generator policy of this call; families and what they are compiled for:
    //   generic one-path-per-lane (basket_dyn_kernel)      every generator, any n
    //   fp64 tiled (9..32)                                 Philox, fp32-normals-widened, external (16 only: BASELINE C4)
    //   fp32 tiled (13..32), fp32 generic pairs            Philox, external (16 only)
    //   fp64 matrix-core (13..16, opt-in)                  Philox
    const bool anti = c->antithetic;
    const Work probe = context_work(c, seed, segs[0], 0, 0);
    const GenSel gen = gen_of(c, probe, sizeof(Real));
    using Kernel = void (*)(const Tail, const BasketDyn<Real>, const Work, Real *);
    Kernel kernel = nullptr;
    // generic kernels: the lane's LDS column holds whole blocks of normals (4 per block in fp32 and under fp32 normals, 8 in fp64)
    const size_t per_block = gen == GEN_F32N ? 4 : GenPhilox::npb<Real>();
    size_t lds = ((size_t)np + per_block - 1) / per_block * per_block * GROUP * sizeof(Real);
    bool pairs = false;  // two paths per lane
    const bool tiled_ok = n >= basket_tiled_min() && n <= 32 && (gen == GEN_PHILOX || gen == GEN_F32N || (gen == GEN_EXTERNAL && n == 16));
    int rc = MC_OK;
    if (gen == GEN_XORWOW || (gen == GEN_EXTERNAL && !tiled_ok) || (gen == GEN_F32N && !tiled_ok)) {
        // one path per lane, normals in the lane's LDS column, any n
        rc = with_anti_gen<GEN_XORWOW | GEN_EXTERNAL | (sizeof(Real) == 8 ? GEN_F32N : 0)>(anti, gen, [&](auto a, auto tag) {
            kernel = basket_dyn_kernel<Real, decltype(a)::value, typename decltype(tag)::type>;
        });
    } else if constexpr (sizeof(Real) == 4) {
        pairs = true;
        if (n >= 13 && tiled_ok) {  // normals in registers, no dynamic LDS
            lds = 0;
            switch (n <= 14 ? n : np) {  // 13, 14: own kernels; 15..32: one per multiple of 4 (zero-padded buffer)
#define MC_CASE(NA) case NA: kernel = anti ? basket_tiled_f32_kernel<NA, true> : basket_tiled_f32_kernel<NA, false>; break;
                MC_CASE(13) MC_CASE(14) MC_CASE(16) MC_CASE(20) MC_CASE(24) MC_CASE(28) MC_CASE(32)
#undef MC_CASE
            }
            if (gen == GEN_EXTERNAL)
                kernel = anti ? nullptr : basket_tiled_f32_kernel<16, false, GenExternal>;
        } else {
            kernel = anti ? basket_dyn_f32_kernel<true> : basket_dyn_f32_kernel<false>;
            lds *= 2;  // the lane's column holds packed pairs
        }
    } else if (with_a4 && basket_mfma() && gen == GEN_PHILOX) {  // fp64, 13..16 assets: the mat-vec on the matrix cores
        lds = 0;
        if constexpr (sizeof(Real) == 8)
            kernel = anti ? basket_mfma_f64_kernel<true> : basket_mfma_f64_kernel<false>;
    } else if (tiled_ok) {  // fp64: normals in registers, no dynamic LDS
        lds = 0;
        // 9..16 assets: one kernel per size; 17..32: one per multiple of 4 (the buffer is zero-padded to whole tiles,
        // padded rows carry coef = 0 and padded columns multiply real normals by 0)
        if constexpr (sizeof(Real) == 8) {
            switch (n <= 16 ? n : np) {
#define MC_CASE(NA)                                                                                                               \
    case NA:                                                                                                                      \
        kernel = gen == GEN_F32N ? (anti ? basket_tiled_kernel<double, NA, true, GenPhiloxF32N> : basket_tiled_kernel<double, NA, false, GenPhiloxF32N>) \
                                 : (anti ? basket_tiled_kernel<double, NA, true> : basket_tiled_kernel<double, NA, false>);       \
        break;
                MC_CASE(9) MC_CASE(10) MC_CASE(11) MC_CASE(12) MC_CASE(13) MC_CASE(14) MC_CASE(15) MC_CASE(16)
                MC_CASE(20) MC_CASE(24) MC_CASE(28) MC_CASE(32)
#undef MC_CASE
            }
            if (gen == GEN_EXTERNAL)
                kernel = anti ? nullptr : basket_tiled_kernel<double, 16, false, GenExternal>;
        }
    } else {
        kernel = anti ? basket_dyn_kernel<Real, true> : basket_dyn_kernel<Real, false>;
    }
    if (rc) return rc;
    if (!kernel)
        return fail(MC_ERR_UNSUPPORTED, "basket: no kernel for this generator / estimator / size combination");
    if (lds)
        HIPCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // the tiled (register-resident) kernels launch more, smaller workgroups (grid_for); the generic ones keep 1x
    const int scale = (lds == 0 && tiled_ok && gen != GEN_XORWOW) ? (sizeof(Real) == 8 ? GRID_SCALE_TILED_F64 : GRID_SCALE_TILED_F32) : 2;
    int total = 0, slot = 0;
    for (const Segment &s : segs)
        total += grid_for(c->blocks, pairs ? (s.count + 1) / 2 : s.count, scale);
    Tail tail = make_tail(c, total, 1.0, 1.0, n_paths, d_triple);
    uint64_t done = 0;
    for (const Segment &s : segs) {
        const Work w = context_work(c, seed, s, 0, 0);
        const int g = grid_for(c->blocks, pairs ? (s.count + 1) / 2 : s.count, scale);
        tail.slot_base = tail.ticket_base = (uint32_t)slot;
        launch_sim_lds(prof, kernel, g, lds, st, tail, k, w, out ? out + done : (Real *)nullptr);
        slot += g;
        done += s.count;
    }
    return finish_call(c, tail, total, st);
}

template <class Real>
static int basket_enqueue(mc_context *c, const typename BasketIn<Real>::type *o, uint64_t seed, uint64_t first,
                          uint64_t n, double *d_triple, hipStream_t st, Real *out)
{
    if (o->n < 1 || o->n > MC_MAX_ASSETS_GENERIC)
        return fail(MC_ERR_UNSUPPORTED, "basket: n=%d outside the supported range 1..%d", o->n, MC_MAX_ASSETS_GENERIC);
    if (!o->s || !o->v || !o->p || !o->d || !o->w)
        return fail(MC_ERR_INVALID, "basket: NULL array");
    if (!(o->t >= 0) || !std::isfinite((double)o->r) || !std::isfinite((double)o->k))
        return fail(MC_ERR_INVALID, "basket: need t>=0 and finite r, k");
    if (int rc = begin_call(c, st)) return rc;
    std::vector<Segment> segs;
    if (int rc = plan_segments(first, n, segs)) return rc;
    int rc = MC_OK;
    ProfileScope prof(c);
    if (c->ext && (segs.size() > 1 || first != 0))
        return fail(MC_ERR_INVALID, "external normals: one segment starting at path 0");
    if (c->rng == MC_RNG_XORWOW && !c->ext) {   // the generic kernel is the one compiled for both generators
        if (sizeof(Real) == 8 && c->normals_f32)
            return fail(MC_ERR_UNSUPPORTED, "fp32 normals in the fp64 kernels are implemented for the Philox generator");
        if (int rc2 = xorwow_one_segment(segs)) return rc2;
        if (int rc2 = xorwow_ready(c, seed, st)) return rc2;
        return basket_launch_dyn<Real>(c, prof, *o, seed, segs, st, out, n, d_triple);
    }
    // kernel-argument / LDS-staged kernels up to the static limit; the policies beyond Philox exist for some sizes only
    int static_max = basket_static_max<Real>();
    if (sizeof(Real) == 8 && c->normals_f32 && static_max > 8)
        static_max = 8;
    if (c->ext)
        static_max = (o->n == 3 || o->n == 4) ? o->n : 0;
    switch (o->n <= static_max ? o->n : 0) {
#define MC_CASE(NA) case NA: rc = basket_launch_n<Real, NA>(c, prof, *o, seed, segs, st, out, n, d_triple); break;
        MC_CASE(1) MC_CASE(2) MC_CASE(3) MC_CASE(4) MC_CASE(5) MC_CASE(6) MC_CASE(7) MC_CASE(8)
        MC_CASE(9) MC_CASE(10) MC_CASE(11) MC_CASE(12) MC_CASE(13) MC_CASE(14) MC_CASE(15) MC_CASE(16)
#undef MC_CASE
    default:
        rc = basket_launch_dyn<Real>(c, prof, *o, seed, segs, st, out, n, d_triple);
        break;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------
// CVA
// ---------------------------------------------------------------------------------------
template <class Real> struct CvaIn;
template <> struct CvaIn<float> { using type = mc_cva_f32; };
template <> struct CvaIn<double> { using type = mc_cva_f64; };

// Per-date table, fp64 on the host, rounded once to Real.  Residual maturities follow the
// reference's rule: t -= dt in Real arithmetic, dates with t < 0 contribute nothing
// (dp/MonteCarloKernel.cu:234,249-256).
// lag = 1 (bridge tests only, MC_FROM_NORMALS_HOST_ORDER): the reference CPU loop's ordering -- the exposure of date j
// is priced at the spot of date j - 1 (dp/MonteCarloHost.c:254-261); the caller then feeds the kernel the
// path's normals delayed by one date, so that its running sum W is the lagged one.
template <class Real>
static int build_cva_table(const typename CvaIn<Real>::type &v, int lag, std::vector<CvaStep<Real>> &tab, std::vector<Real> &extra,
                           CvaArgs<Real> &args)
{
    const auto &o = v.option;
    if (!finite_pos(o.s) || !finite_pos(o.k) || !finite_pos(o.v) || !finite_pos(o.t) || !std::isfinite((double)o.r))
        return fail(MC_ERR_INVALID, "cva: need s>0, k>0, v>0, t>0, finite r");
    if (v.n_grid < 1 || v.n_grid > (1 << 20))
        return fail(MC_ERR_INVALID, "cva: n_grid=%d outside [1, 2^20]", v.n_grid);
    if (!(v.defint >= 0) || !std::isfinite((double)v.lgd))
        return fail(MC_ERR_INVALID, "cva: need defint>=0 and finite lgd");
    const double sc = exp_scale<Real>();
    const Real dt = o.t / v.n_grid;
    const Real step_drift = (Real)(((double)o.r - 0.5 * (double)o.v * (double)o.v) * (double)dt);
    const Real step_vol = (Real)((double)o.v * std::sqrt((double)dt));
    const double ln_s0 = std::log((double)o.s), ln_k = std::log((double)o.k);
    const double lam = (double)v.defint;
    tab.clear();
    extra.clear();   // first sqrt(tau_j) of every date, then sigma t_j of every date (the Greeks kernel's extra columns)
    std::vector<Real> sig_t;
    args.n_bs = 0;
    args.last_intrinsic = 0;
    Real ttm = o.t;
    for (int j = 1; j <= v.n_grid; ++j) {
        ttm -= dt;
        if (!(ttm >= 0))
            break;
        const double tau = (double)ttm;
        const double t_prev = (double)dt * (double)(j - 1), t_now = (double)dt * (double)j;
        CvaStep<Real> s;
        s.dp = (Real)(-std::exp(-lam * t_prev) * std::expm1(-lam * (t_now - t_prev)));
        const double ln_sj = ln_s0 + (double)(j - lag) * (double)step_drift;  // + step_vol * W_j on the device
        s.xk = (Real)(ln_sj * sc);
        if (tau > 0) {
            const double sig = (double)o.v * std::sqrt(tau);
            s.g = (Real)((double)step_vol / sig);
            const double num = ln_sj - ln_k + ((double)o.r + 0.5 * (double)o.v * (double)o.v) * tau;
            s.e1 = (Real)(num / sig);
            s.e2 = (Real)(num / sig - sig);
            s.disc = (Real)((double)o.k * std::exp(-(double)o.r * tau));
            args.n_bs++;
        } else {
            s.g = s.e1 = s.e2 = s.disc = 0;
            args.last_intrinsic = 1;
        }
        tab.push_back(s);
        extra.push_back((Real)std::sqrt(tau));
        sig_t.push_back((Real)((double)o.v * t_now));
        if (tau == 0)
            break;
    }
    extra.insert(extra.end(), sig_t.begin(), sig_t.end());
    args.bx = (Real)((double)step_vol * sc);
    args.lgd = v.lgd;
    args.strike = o.k;
    // W is a sum of up to n_grid normals: its worst case is astronomically unlikely, so only the hard limit of the
    // device's exp (EXP_F64_ARG_LIMIT, mc_math_f64.hpp) is enforced here; below it an honest overflow of the spot gives
    // inf, as it would on any machine (beyond it the table index would wrap and return a finite wrong spot)
    if (!(std::fabs(ln_s0) + (double)v.n_grid * (std::fabs((double)step_drift) + std::fabs((double)step_vol) * Z_MAX_F64) < EXP_F64_ARG_LIMIT))
        return fail(MC_ERR_INVALID, "cva: drift and volatility put the simulated spot outside the range of a double");
    return MC_OK;
}

// Build (or reuse) the per-date table of a CVA call in the context's table buffer, ordered on `st`.
template <class Real>
static int cva_table_ready(mc_context *c, const typename CvaIn<Real>::type *v, hipStream_t st, CvaArgs<Real> &args)
{
    // the table depends only on the inputs: rebuild and re-upload only when they change
    const int lag = (c->ext && (c->ext_flags & MC_FROM_NORMALS_HOST_ORDER)) ? 1 : 0;
    const double key_vals[10] = {(double)v->defint, (double)v->lgd, (double)v->option.s, (double)v->option.k,
                                 (double)v->option.r, (double)v->option.v, (double)v->option.t,
                                 (double)v->n_grid, (double)sizeof(Real), (double)lag};
    std::vector<char> key(sizeof key_vals);
    memcpy(key.data(), key_vals, sizeof key_vals);
    if (key == c->table_key && c->cva_args.size() == sizeof args) {   // same inputs as the table in HBM: nothing to rebuild
        memcpy(&args, c->cva_args.data(), sizeof args);
        return upload_table(c, st, key, nullptr, 0);                  // (orders `st` behind the upload if it is another stream)
    }
    static thread_local std::vector<CvaStep<Real>> tab;
    static thread_local std::vector<Real> extra, blob;
    if (int rc = build_cva_table<Real>(*v, lag, tab, extra, args)) return rc;
    // one blob: the per-date rows | fp32 only: the rows of date pairs, field by field | the Greeks kernel's extra columns
    const size_t step_reals = tab.size() * (sizeof(CvaStep<Real>) / sizeof(Real));
    const size_t pair_reals = sizeof(Real) == 4 ? (size_t)12 * (args.n_bs / 2) : 0;
    blob.resize(step_reals + pair_reals + extra.size());
    memcpy(blob.data(), tab.data(), step_reals * sizeof(Real));
    if constexpr (sizeof(Real) == 4) {
        // the closed-form rows once more, two dates per row and field by field, so that a date pair's
        // {g, g'} ... {dp, dp'} are adjacent scalars = ready-made operands of the packed instructions
        float *row = blob.data() + step_reals;
        for (int q = 0; q < args.n_bs / 2; ++q, row += 12) {
            const CvaStep<float> &a = tab[2 * q], &b = tab[2 * q + 1];
            const float vals[12] = {a.g, b.g, a.e1, b.e1, a.e2, b.e2, a.xk, b.xk, a.disc, b.disc, a.dp, b.dp};
            memcpy(row, vals, sizeof vals);
        }
    }
    memcpy(blob.data() + step_reals + pair_reals, extra.data(), extra.size() * sizeof(Real));
    if (int rc = upload_table(c, st, key, blob.data(), blob.size() * sizeof(Real))) return rc;
    args.pairs = nullptr;
    args.pairs_in_lds = 0;   // set per launch (cva_enqueue)
    if constexpr (sizeof(Real) == 4)
        args.pairs = (const float *)c->d_table + step_reals;
    args.extra = (const Real *)c->d_table + step_reals + pair_reals;
    args.steps = (const CvaStep<Real> *)c->d_table;
    c->cva_args.assign((const char *)&args, (const char *)&args + sizeof args);
    return MC_OK;
}

// workgroups of the date-parallel part of a launch over `count` paths at 2^log2_lanes lanes per path: one lane-group per path
// until the grid reaches the CVA cap (12 per CU), grid-stride beyond
static int grid_for_cva_dates(int blocks, uint32_t count, int log2_lanes)
{
    const uint64_t need = (((uint64_t)count << log2_lanes) + GROUP - 1) / GROUP, cap = (uint64_t)blocks * GRID_SCALE_CVA / 2;
    return (int)(need < cap ? (need ? need : 1) : cap);
}

template <class Real>
static int cva_enqueue(mc_context *c, const typename CvaIn<Real>::type *v, uint64_t seed, uint64_t first, uint64_t n,
                       double *d_triple, hipStream_t st, Real *out)
{
    if (int rc = begin_call(c, st)) return rc;
    CvaArgs<Real> args;
    if (int rc = cva_table_ready<Real>(c, v, st, args)) return rc;
    const bool xorwow = c->rng == MC_RNG_XORWOW && !c->ext;
    // The cut between one lane per path and the date-parallel form (mc_launch_shape.hpp: cva_plan).  The date-parallel form
    // needs a generator a path can be entered in the middle of (XORWOW is one sequence per lane) and the per-date table in
    // LDS.  A call with BOTH parts is one launch of cva_split_kernel: compiled for the plain estimator on the counter-based
    // generators, one segment each (the external array is indexed from the call's first path), and a table small enough not
    // to cost the one-lane-per-path workgroups their residency (24 KB: 4 workgroups per CU beside the fp64 math tables)
    const int n_dates = args.n_bs + args.last_intrinsic;
    const size_t table_lds = (size_t)(n_dates + (n_dates & 1)) * sizeof(CvaStep<Real>);   // whole date pairs (the fp32 layout pairs them)
    // (calls on an external array -- the from-normals hooks, the staged launch-geometry form -- are compared bit for bit with
    // one-lane-per-path kernels: they go date-parallel only when the setting forces it)
    const int lanes_setting = (c->ext && c->cva_date_lanes == 0) ? 1 : c->cva_date_lanes;
    CvaPlan plan = cva_plan(lanes_setting, n, n_dates, c->compute_units, !xorwow && table_lds <= 48 * 1024, sizeof(Real));
    std::vector<Segment> segs, tail_segs;
    if (plan.main_paths && plan.tail_paths) {
        bool split = !c->antithetic && !c->ext && table_lds <= 24 * 1024;
        if (split) {
            if (int rc = plan_segments(first, plan.main_paths, segs)) return rc;
            if (int rc = plan_segments(first + plan.main_paths, plan.tail_paths, tail_segs)) return rc;
            split = segs.size() == 1 && tail_segs.size() == 1;
        }
        if (!split)
            plan = CvaPlan{n, 0, 0}, segs.clear(), tail_segs.clear();
    }
    const bool fused_split = plan.main_paths && plan.tail_paths;
    if (!fused_split) {
        if (plan.main_paths)
            if (int rc = plan_segments(first, plan.main_paths, segs)) return rc;
        if (plan.tail_paths)
            if (int rc = plan_segments(first, plan.tail_paths, tail_segs)) return rc;
    }
    const int scale = xorwow ? 2 : GRID_SCALE_CVA;
    int total = 0, slot = 0;
    for (const Segment &s : segs)
        total += grid_for(c->blocks, s.count, scale);
    for (const Segment &s : tail_segs)
        total += grid_for_cva_dates(c->blocks, s.count, plan.log2_lanes);
    Tail t = make_tail(c, total, 1.0, 1.0, n, d_triple);
    uint64_t done = 0;
    ProfileScope prof(c);
    if (c->ext && (segs.size() + tail_segs.size() > 1 || first != 0))
        return fail(MC_ERR_INVALID, "external normals: one segment starting at path 0");
    if (xorwow) {
        if (sizeof(Real) == 8 && c->normals_f32)
            return fail(MC_ERR_UNSUPPORTED, "fp32 normals in the fp64 kernels are implemented for the Philox generator");
        if (int rc = xorwow_one_segment(segs)) return rc;
        if (int rc = xorwow_ready(c, seed, st)) return rc;
    }
    // fp32, one lane per path: the date pairs' rows through LDS instead of scalar registers (mc_kernels.hpp: cva_path<float>) while the copy
    // leaves the kernel its eight waves per SIMD (16 KB = 341 date pairs; a split launch already carries the whole table's LDS)
    const size_t pairs_lds = sizeof(Real) == 4 ? (size_t)48 * (size_t)(args.n_bs / 2) : 0;
    const bool pairs_fit = pairs_lds > 0 && pairs_lds <= 16 * 1024 && !c->antithetic;   // (the antithetic instantiation does not take them)
    if (fused_split) {
        args.pairs_in_lds = pairs_lds > 0;
        const Work w = context_work(c, seed, segs[0], 0, 0), wt = context_work(c, seed, tail_segs[0], 0, 0);
        const int g_tail = grid_for_cva_dates(c->blocks, tail_segs[0].count, plan.log2_lanes);
        Real *dst = out, *dst_tail = out ? out + segs[0].count : (Real *)nullptr;
        constexpr unsigned ALLOW = GEN_PHILOX | (sizeof(Real) == 8 ? GEN_F32N : 0);
        if (int rc = with_gen<ALLOW>(gen_of(c, w, sizeof(Real)), [&](auto tag) {
                launch_sim_lds(prof, cva_split_kernel<Real, CVA_DATES_CH, typename decltype(tag)::type>, total, table_lds, st, t, args, w, wt,
                               (uint32_t)g_tail, (uint32_t)plan.log2_lanes, dst, dst_tail);
            }))
            return rc;
        return finish_call(c, t, total, st);
    }
    args.pairs_in_lds = pairs_fit;
    for (const Segment &s : segs) {
        const Work w = context_work(c, seed, s, 0, 0);
        const int g = grid_for(c->blocks, s.count, scale);
        t.slot_base = t.ticket_base = (uint32_t)slot;
        Real *dst = out ? out + done : (Real *)nullptr;
        constexpr unsigned ALLOW = GEN_PHILOX | GEN_XORWOW | GEN_EXTERNAL | (sizeof(Real) == 8 ? GEN_F32N : 0);
        if (int rc = with_anti_gen<ALLOW>(c->antithetic, gen_of(c, w, sizeof(Real)), [&](auto a, auto tag) {
                launch_sim_lds(prof, cva_kernel<Real, decltype(a)::value, typename decltype(tag)::type>, g, pairs_fit ? pairs_lds : 0, st, t, args, w, dst);
            }))
            return rc;
        slot += g;
        done += s.count;
    }
    for (const Segment &s : tail_segs) {
        const Work w = context_work(c, seed, s, 0, 0);
        const int g = grid_for_cva_dates(c->blocks, s.count, plan.log2_lanes);
        t.slot_base = t.ticket_base = (uint32_t)slot;
        Real *dst = out ? out + done : (Real *)nullptr;
        constexpr unsigned ALLOW = GEN_PHILOX | GEN_EXTERNAL | (sizeof(Real) == 8 ? GEN_F32N : 0);
        if (int rc = with_anti_gen<ALLOW>(c->antithetic, gen_of(c, w, sizeof(Real)), [&](auto a, auto tag) {
                launch_sim_lds(prof, cva_dates_kernel<Real, CVA_DATES_CH, decltype(a)::value, typename decltype(tag)::type>, g, table_lds, st, t,
                               args, w, (uint32_t)plan.log2_lanes, dst);
            }))
            return rc;
        slot += g;
        done += s.count;
    }
    return finish_call(c, t, total, st);
}

// ---------------------------------------------------------------------------------------
// Greeks of the basket call and of the CVA: secondary kernels, plain estimator, synchronous
// ---------------------------------------------------------------------------------------
template <class Real, bool LR>
static int basket_greeks_run(mc_context *c, const typename BasketIn<Real>::type *o, uint64_t seed, uint64_t first, uint64_t n,
                             mc_result *price, mc_result *delta, mc_result *vega)
{
    if (int rc = check_common(c, o, first, n, price)) return rc;
    if (!delta || !vega)
        return fail(MC_ERR_INVALID, "basket greeks: NULL output array");
    if (o->n < 1 || o->n > MC_MAX_ASSETS_GENERIC)
        return fail(MC_ERR_UNSUPPORTED, "basket: n=%d outside the supported range 1..%d", o->n, MC_MAX_ASSETS_GENERIC);
    if (!o->s || !o->v || !o->p || !o->d || !o->w)
        return fail(MC_ERR_INVALID, "basket: NULL array");
    if (!(o->t >= 0) || !std::isfinite((double)o->r) || !std::isfinite((double)o->k))
        return fail(MC_ERR_INVALID, "basket: need t>=0 and finite r, k");
    if (c->antithetic || c->control)
        return fail(MC_ERR_UNSUPPORTED, "greeks: only the plain estimator is implemented");
    const int na = o->n;
    const double sqrt_t = std::sqrt((double)o->t);
    // table: L[n*n] | d | mu | v | w | s | 1/s | v t      (the reference's unfolded constants, dp/MonteCarloKernel.cu:74-101)
    //        | likelihood ratio only: M = L^-T [n*n] | 1 / (s v sqrt t) | 1 / v | (sqrt t d - v t) / (v sqrt t)
    std::vector<Real> host((size_t)na * na + 7 * (size_t)na + (LR ? (size_t)na * na + 3 * (size_t)na : 0), (Real)0);
    Real *L = host.data(), *d = L + (size_t)na * na, *mu = d + na, *v = mu + na, *w = v + na, *s0 = w + na, *inv_s = s0 + na, *vt = inv_s + na;
    if (LR) {
        // The scores divide by sigma sqrt t and whiten with the inverse factor: y = L^-T g, i.e. M = L^-T (upper triangular), in
        // fp64 from the caller's factor by back-substitution on the columns of L^-1, rounded once
        if (!((double)o->t > 0))
            return fail(MC_ERR_INVALID, "likelihood-ratio greeks: need t>0 (the scores divide by sigma sqrt t)");
        std::vector<double> inv((size_t)na * na, 0.0);   // L^-1, lower triangular
        for (int a = 0; a < na; ++a) {
            const double laa = (double)o->p[a * na + a];
            if (!((double)o->v[a] > 0) || !(laa > 0))
                return fail(MC_ERR_INVALID, "likelihood-ratio greeks: need v[%d] > 0 and a non-singular factor (p[%d][%d] = %g): the joint density "
                                            "of the terminal prices must exist", a, a, a, laa);
            for (int b = 0; b <= a; ++b) {
                double acc = a == b ? 1.0 : 0.0;
                for (int k = b; k < a; ++k)
                    acc -= (double)o->p[a * na + k] * inv[(size_t)k * na + b];
                inv[(size_t)a * na + b] = acc / laa;
            }
        }
        Real *M = vt + na, *inv_svt = M + (size_t)na * na, *inv_v = inv_svt + na, *mcoef = inv_v + na;
        for (int a = 0; a < na; ++a) {
            for (int b = a; b < na; ++b)
                M[(size_t)a * na + b] = (Real)inv[(size_t)b * na + a];   // (L^-T)_ab = (L^-1)_ba
            const double va = (double)o->v[a];
            inv_svt[a] = (Real)(1.0 / ((double)o->s[a] * va * sqrt_t));
            inv_v[a] = (Real)(1.0 / va);
            mcoef[a] = (Real)((sqrt_t * (double)o->d[a] - va * (double)o->t) / (va * sqrt_t));
        }
    }
    for (int a = 0; a < na; ++a) {
        if (!finite_pos((double)o->s[a]))
            return fail(MC_ERR_INVALID, "basket greeks: need s[a] > 0 (delta is taken with respect to it)");
        const double va = (double)o->v[a];
        double bound = std::fabs(((double)o->r - 0.5 * va * va) * (double)o->t + va * sqrt_t * (double)o->d[a]);
        for (int b = 0; b <= a; ++b) {
            L[(size_t)a * na + b] = o->p[a * na + b];
            bound += std::fabs(va * sqrt_t * (double)o->p[a * na + b]) * Z_MAX_F64;
        }
        if (!exponent_in_range(sizeof(Real) == 4 ? bound * 8 : bound))   // fp32: e^88 is the limit
            return fail(MC_ERR_INVALID, "basket: asset %d's drift and volatility put its terminal price outside the range of the simulation type", a);
        d[a] = o->d[a];
        mu[a] = (Real)(((double)o->r - 0.5 * va * va) * (double)o->t);
        v[a] = o->v[a], w[a] = o->w[a], s0[a] = o->s[a];
        inv_s[a] = (Real)(1.0 / (double)o->s[a]);
        vt[a] = (Real)(va * (double)o->t);
    }
    const size_t bytes = host.size() * sizeof(Real);
    std::vector<char> key(bytes + 2);
    memcpy(key.data(), host.data(), bytes);
    key[bytes] = (char)sizeof(Real);
    key[bytes + 1] = LR ? 'R' : 'G';
    if (int rc = begin_call(c, c->stream)) return rc;
    if (int rc = upload_table(c, c->stream, key, host.data(), bytes)) return rc;
    BasketGreeks<Real> k;
    k.consts = (const Real *)c->d_table;
    k.n = na;
    k.strike = o->k;
    k.sqrt_t = (Real)sqrt_t;
    constexpr int NPB = GenPhilox::npb<Real>();
    const size_t lds = (size_t)((na + NPB - 1) / NPB * NPB) * GROUP * sizeof(Real);
    HIPCHK(hipFuncSetAttribute((const void *)basket_greeks_kernel<Real, LR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<mc_result *> out(1 + 2 * (size_t)na);
    out[0] = price;
    for (int a = 0; a < na; ++a)
        out[1 + a] = delta + a, out[1 + na + a] = vega + a;
    // one pass per BASKET_GREEKS_CHUNK assets: the grid's y index is the chunk
    const int chunks = (na + BASKET_GREEKS_CHUNK - 1) / BASKET_GREEKS_CHUNK;
    return planes_run(c, sizeof(Real), 1 + 2 * na, chunks, first, n, n, std::exp(-(double)o->r * (double)o->t), out.data(),
                      [&](const Tail &t, const Segment &s, int g, hipStream_t st) {
                          hipLaunchKernelGGL((basket_greeks_kernel<Real, LR>), dim3(g, chunks), dim3(GROUP), lds, st, t, k, make_work(seed, s, 0, 0));
                      });
}
extern "C" int mc_basket_greeks_run_f32(mc_context *c, const mc_basket_f32 *o, uint64_t seed, uint64_t first, uint64_t n,
                                        mc_result *price, mc_result *delta, mc_result *vega)
{
    return basket_greeks_run<float, false>(c, o, seed, first, n, price, delta, vega);
}
extern "C" int mc_basket_greeks_run_f64(mc_context *c, const mc_basket_f64 *o, uint64_t seed, uint64_t first, uint64_t n,
                                        mc_result *price, mc_result *delta, mc_result *vega)
{
    return basket_greeks_run<double, false>(c, o, seed, first, n, price, delta, vega);
}
extern "C" int mc_basket_greeks_lr_run_f32(mc_context *c, const mc_basket_f32 *o, uint64_t seed, uint64_t first, uint64_t n,
                                           mc_result *price, mc_result *delta, mc_result *vega)
{
    return basket_greeks_run<float, true>(c, o, seed, first, n, price, delta, vega);
}
extern "C" int mc_basket_greeks_lr_run_f64(mc_context *c, const mc_basket_f64 *o, uint64_t seed, uint64_t first, uint64_t n,
                                           mc_result *price, mc_result *delta, mc_result *vega)
{
    return basket_greeks_run<double, true>(c, o, seed, first, n, price, delta, vega);
}

template <class Real, bool LR>
static int cva_greeks_run(mc_context *c, const typename CvaIn<Real>::type *v, uint64_t seed, uint64_t first, uint64_t n, mc_cva_greeks *out)
{
    if (int rc = check_common(c, v, first, n, out)) return rc;
    if (c->antithetic)
        return fail(MC_ERR_UNSUPPORTED, "greeks: only the plain estimator is implemented");
    if (int rc = begin_call(c, c->stream)) return rc;
    CvaArgs<Real> args;
    if (int rc = cva_table_ready<Real>(c, v, c->stream, args)) return rc;
    mc_result *r[3] = {&out->cva, &out->delta, &out->vega};
    const Real inv_spot = (Real)(1.0 / (double)v->option.s);
    const Real dt = v->option.t / v->n_grid;   // the table's own dt (build_cva_table)
    const Real sqrt_dt = (Real)std::sqrt((double)dt);
    const Real lr_delta = (Real)(1.0 / ((double)v->option.s * (double)v->option.v * std::sqrt((double)dt))), inv_sigma = (Real)(1.0 / (double)v->option.v);
    return planes_run(c, sizeof(Real), 3, 1, first, n, n, 1.0, r, [&](const Tail &t, const Segment &s, int g, hipStream_t st) {
        cva_greeks_kernel<Real, LR><<<g, GROUP, 0, st>>>(t, args, make_work(seed, s, 0, 0), inv_spot, sqrt_dt, lr_delta, inv_sigma);
    });
}
extern "C" int mc_cva_greeks_run_f32(mc_context *c, const mc_cva_f32 *v, uint64_t seed, uint64_t first, uint64_t n, mc_cva_greeks *out)
{
    return cva_greeks_run<float, false>(c, v, seed, first, n, out);
}
extern "C" int mc_cva_greeks_run_f64(mc_context *c, const mc_cva_f64 *v, uint64_t seed, uint64_t first, uint64_t n, mc_cva_greeks *out)
{
    return cva_greeks_run<double, false>(c, v, seed, first, n, out);
}
extern "C" int mc_cva_greeks_lr_run_f32(mc_context *c, const mc_cva_f32 *v, uint64_t seed, uint64_t first, uint64_t n, mc_cva_greeks *out)
{
    return cva_greeks_run<float, true>(c, v, seed, first, n, out);
}
extern "C" int mc_cva_greeks_lr_run_f64(mc_context *c, const mc_cva_f64 *v, uint64_t seed, uint64_t first, uint64_t n, mc_cva_greeks *out)
{
    return cva_greeks_run<double, true>(c, v, seed, first, n, out);
}

// ---------------------------------------------------------------------------------------
// closing + sync wrappers
// ---------------------------------------------------------------------------------------
// mc_closing, mc_shard_range, mc_chol_*, mc_factor_from_cov_*: mc_hostmath.c (host-only C, shared with the CPU twin)

// run = enqueue on the context stream, wait, read 24 bytes, close.
//   timing on (default): two HIP events around the kernels (mc_result.kernel_ms), a 24-byte D2H copy, a stream
//     synchronize -- about 20 us of host time around the kernel.
//   timing off (mc_context_set_timing(ctx, 0); the legacy symbols unless MC_VERBOSE is set): the last workgroup of the
//     call writes the triple straight into pinned host memory (Tail.host_triple) and this thread polls the n word from
//     user space: no event records, no copy command, no sleeping wait.  kernel_ms is reported as 0.
template <class Enq>
static int run_sync(mc_context *c, uint64_t n, double discount, mc_result *out, Enq enqueue)
{
    using clock = std::chrono::steady_clock;
    const auto ms_between = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // a caller that did part of the call's work before coming here (the staged launch-geometry form) has set call_t0
    const auto wall0 = c->call_t0_valid ? c->call_t0 : clock::now();
    if (!c->call_t0_valid)   // what earlier calls of other kinds (per-path dumps, Greeks, test hooks) left in the accumulators is not this call's
        c->acc_setup_ms = c->acc_table_ms = 0;
    c->call_t0_valid = false;
    HIPCHK(hipSetDevice(c->device));
    c->armed = false;   // mc_context_arm_direct applies to the next mc_*_launch_* only: a synchronous call in between cancels it
    float ms = 0;
    const double *h = c->h_triple;
    clock::time_point t_enqueued;
    struct Scope {   // whatever way out: the stage accumulators and flags belong to ONE call
        mc_context *c;
        ~Scope() { c->acc_setup_ms = c->acc_table_ms = 0, c->stats_timed_call = false; }
    } scope{c};
    if (!c->timing && c->fused) {
        volatile double *flag = c->h_direct + 2;
        *flag = DIRECT_SENTINEL;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        c->direct_target = c->d_direct;
        const int rc = enqueue(c->stream, c->d_triple);
        c->direct_target = nullptr;
        if (rc) return rc;
        t_enqueued = clock::now();
        // poll from user space for the first 50 ms (the calls that care about 20 us are shorter than that), then hand the
        // core back and wait in the runtime (also the way out when a kernel failed and never writes)
        bool seen = false;
        for (uint32_t spin = 0;; ++spin) {
            if (__atomic_load_n((const uint64_t *)(c->h_direct + 2), __ATOMIC_ACQUIRE) != __builtin_bit_cast(uint64_t, DIRECT_SENTINEL)) {
                seen = true;
                break;
            }
            if ((spin & 255u) == 255u && clock::now() - wall0 > std::chrono::milliseconds(50))
                break;
            __builtin_ia32_pause();
        }
        if (!seen) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (*flag == DIRECT_SENTINEL)
                return fail(MC_ERR_HIP, "the device never delivered the result of a synchronous call");
        }
        h = c->h_direct;
    } else {
        // poison the slot (all bits set: NaN): a reduction that never finishes must not hand back the previous call's
        // triple, whose n word would pass the check below whenever n is unchanged
        HIPCHK(hipMemsetAsync(c->d_triple, 0xFF, 3 * sizeof(double), c->stream));
        if (c->timing) HIPCHK(hipEventRecord(c->ev0, c->stream));
        c->stats_timed_call = c->timing;   // set-up that completes on the device inside enqueue re-records ev0 behind itself
        if (int rc = enqueue(c->stream, c->d_triple)) return rc;
        c->stats_timed_call = false;
        if (c->timing) HIPCHK(hipEventRecord(c->ev1, c->stream));
        t_enqueued = clock::now();
        HIPCHK(hipMemcpyAsync(c->h_triple, c->d_triple, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->timing) HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    }
    const auto t_result = clock::now();
    if (!(h[2] == (double)n))   // also catches the poison (NaN) of a reduction that never closed
        return fail(MC_ERR_HIP, "device returned n=%g, expected %llu: the call's final reduction did not complete", h[2],
                    (unsigned long long)n);
    out->sum = h[0];
    out->sum2 = h[1];
    out->n = (uint64_t)h[2];
    out->kernel_ms = ms;
    if (out->n != n)
        return fail(MC_ERR_HIP, "device returned n=%llu, expected %llu", (unsigned long long)out->n,
                    (unsigned long long)n);
    mc_closing(out->sum, out->sum2, out->n, discount, &out->expected, &out->confidence);
    const auto t_closed = clock::now();
    out->wall_ms = (float)ms_between(wall0, t_closed);
    // the stage breakdown (mc_context_last_call_stats): consecutive host-clock intervals, the device's kernel time taken out of the wait
    mc_call_stats &k = c->stats;
    const double before = ms_between(wall0, t_enqueued), wait = ms_between(t_enqueued, t_result);
    k.setup_ms = (float)c->acc_setup_ms;
    k.table_upload_ms = (float)c->acc_table_ms;
    k.launch_ms = (float)std::max(0.0, before - c->acc_setup_ms - c->acc_table_ms);
    // the events' figure, capped at the host's wait: the opening event is stamped when the idle device reaches it, so on the FIRST launch
    // of a kernel it also spans the code-object load the host spent inside the launch call (already counted in launch_ms)
    k.kernel_ms = (float)std::min((double)ms, wait);
    k.readback_ms = (float)std::max(0.0, wait - (double)k.kernel_ms);
    k.closing_ms = (float)ms_between(t_result, t_closed);
    k.wall_ms = out->wall_ms;
    k.context_create_ms = c->create_ms;
    k.first_call = c->sync_calls++ == 0;
    return MC_OK;
}

template <class Real, class Enq>
static int dump_sync(mc_context *c, uint64_t n, Real *h_out, Enq enqueue)
{
    if (n > MAX_DUMP_PATHS)
        return fail(MC_ERR_INVALID, "per-path dump limited to 2^26 paths");
    HIPCHK(hipSetDevice(c->device));
    if (int rc = ensure_out(c, n * sizeof(Real))) return rc;
    if (int rc = enqueue(c->stream, c->d_triple, (Real *)c->d_out)) return rc;
    HIPCHK(hipMemcpyAsync(h_out, c->d_out, n * sizeof(Real), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MC_OK;
}

// ---------------------------------------------------------------------------------------
// Test hooks: the simulation kernels on a caller-supplied normal stream (include/mc_mi355x.h "test hooks")
// ---------------------------------------------------------------------------------------
// Uploads the normals into the context's buffer (zero-padded to `padded` Reals), runs `enqueue` with the external-normals
// policy switched on, synchronously, and -- when h_values is given -- once more through the per-path dump path.
// the context's external-normals buffer, at least `bytes`, with no earlier kernel still reading it
static int ensure_ext(mc_context *c, size_t bytes)
{
    if (c->d_ext_bytes < bytes) {
        if (int rc = quiesce(c)) return rc;
        if (c->d_ext) HIPCHK(hipFree(c->d_ext));
        c->d_ext = nullptr, c->d_ext_bytes = 0;
        if (hipMalloc(&c->d_ext, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return fail(MC_ERR_HIP, "external normals: cannot allocate %zu bytes of device memory", bytes);
        }
        c->d_ext_bytes = bytes;
    }
    HIPCHK(hipStreamSynchronize(c->stream));   // a previous call's kernels may still read the buffer
    return MC_OK;
}

template <class Real, class Enq>
static int from_normals_run(mc_context *c, const Real *h_normals, size_t count, size_t padded, uint32_t per_unit, int flags,
                            uint64_t n, double discount, Real *h_values, mc_result *out, Enq enqueue)
{
    if (!h_normals)
        return fail(MC_ERR_INVALID, "NULL normals");
    if (n > MAX_DUMP_PATHS || padded > (size_t)1 << 31)
        return fail(MC_ERR_INVALID, "from_normals: at most 2^26 paths and 2^31 normals");
    if (c->antithetic || c->control || c->rng != MC_RNG_PHILOX || c->normals_f32)
        return fail(MC_ERR_UNSUPPORTED, "from_normals: plain estimator with the context's default settings only");
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = padded * sizeof(Real);
    if (int rc = ensure_ext(c, bytes)) return rc;
    if (padded > count)
        HIPCHK(hipMemsetAsync(c->d_ext, 0, bytes, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_ext, h_normals, count * sizeof(Real), hipMemcpyHostToDevice, c->stream));
    c->ext = c->d_ext, c->ext_per_unit = per_unit, c->ext_flags = flags;
    int rc = run_sync(c, n, discount, out, [&](hipStream_t st, double *t) { return enqueue(st, t, (Real *)nullptr); });
    if (rc == MC_OK && h_values)
        rc = dump_sync<Real>(c, n, h_values, [&](hipStream_t st, double *t, Real *d) { return enqueue(st, t, d); });
    c->ext = nullptr, c->ext_per_unit = 0, c->ext_flags = 0;
    c->table_key.clear();   // a table built under the test switches must not be reused by a pricing call
    return rc;
}

#define MC_DEFINE_FROM_NORMALS(X, Real)                                                                       \
    extern "C" int mc_vanilla_from_normals_##X(mc_context *c, const mc_option_##X *o, const Real *h_normals,  \
                                               uint64_t n, Real *h_values, mc_result *out)                    \
    {                                                                                                         \
        if (int rc = check_common(c, o, 0, n, out)) return rc;                                                \
        const uint64_t NPB = GenPhilox::npb<Real>(), units = (n + NPB - 1) / NPB;                             \
        return from_normals_run<Real>(c, h_normals, n, units * NPB, (uint32_t)NPB, 0, n,                      \
                                      std::exp(-(double)o->r * (double)o->t), h_values, out,                  \
                                      [&](hipStream_t st, double *t, Real *d) {                               \
                                          return vanilla_enqueue<Real>(c, o, 0, 0, n, t, st, d);              \
                                      });                                                                     \
    }                                                                                                         \
    extern "C" int mc_basket_from_normals_##X(mc_context *c, const mc_basket_##X *o, const Real *h_normals,   \
                                              uint64_t n, int flags, Real *h_values, mc_result *out)          \
    {                                                                                                         \
        if (int rc = check_common(c, o, 0, n, out)) return rc;                                                \
        if (o->n < 1 || o->n > MC_MAX_ASSETS_GENERIC) return fail(MC_ERR_INVALID, "basket: bad n");           \
        /* the two-paths-per-lane kernels read the pair's second unit even where it does not exist: pad */    \
        const size_t count = (size_t)n * (size_t)o->n;                                                        \
        return from_normals_run<Real>(c, h_normals, count, count + (size_t)o->n, (uint32_t)o->n,              \
                                      flags & MC_FROM_NORMALS_NO_VOL, n, std::exp(-(double)o->r * (double)o->t), \
                                      h_values, out, [&](hipStream_t st, double *t, Real *d) {                \
                                          return basket_enqueue<Real>(c, o, 0, 0, n, t, st, d);               \
                                      });                                                                     \
    }                                                                                                         \
    extern "C" int mc_cva_from_normals_##X(mc_context *c, const mc_cva_##X *o, const Real *h_normals,         \
                                           uint64_t n, int flags, Real *h_values, mc_result *out)             \
    {                                                                                                         \
        if (int rc = check_common(c, o, 0, n, out)) return rc;                                                \
        if (o->n_grid < 1 || o->n_grid > (1 << 20)) return fail(MC_ERR_INVALID, "cva: bad n_grid");           \
        const size_t g = (size_t)o->n_grid, count = (size_t)n * g;                                            \
        const Real *src = h_normals;                                                                          \
        std::vector<Real> delayed;                                                                            \
        if (flags & MC_FROM_NORMALS_HOST_ORDER) { /* date j sees the normals of dates 1 .. j-1 */             \
            if (!h_normals) return fail(MC_ERR_INVALID, "NULL normals");                                      \
            delayed.resize(count);                                                                            \
            for (uint64_t i = 0; i < n; ++i) {                                                                \
                delayed[i * g] = 0;                                                                           \
                for (size_t j = 1; j < g; ++j)                                                                \
                    delayed[i * g + j] = h_normals[i * g + j - 1];                                            \
            }                                                                                                 \
            src = delayed.data();                                                                             \
        }                                                                                                     \
        return from_normals_run<Real>(c, src, count, count, (uint32_t)o->n_grid, flags & MC_FROM_NORMALS_HOST_ORDER, \
                                      n, 1.0, h_values, out, [&](hipStream_t st, double *t, Real *d) {        \
                                          return cva_enqueue<Real>(c, o, 0, 0, n, t, st, d);                  \
                                      });                                                                     \
    }

// ---------------------------------------------------------------------------------------
// Compatibility mode: the reference's launch geometry and per-thread XORWOW streams (include/mc_mi355x.h, "launch
// geometry").  The normals of the whole call are drawn the reference's way -- one XORWOW state per (block, thread),
// curand_init(blockIdx.x + gridDim.x, threadIdx.x, 0), curand_normal() one after the other, thread t of a block taking
// paths t, t + T, ... (dp/MonteCarloKernel.cu:285-290,146-150,191-196,240-262) -- into HBM by grid_normals_kernel, and
// the simulation kernels then price them through the external-normals policy: the payoff, accumulation and closing
// code is the hot kernels' own.  Memory: one Real per draw (a 1e8-path vanilla call: 0.4 / 0.8 GB).
// ---------------------------------------------------------------------------------------
static int grid_check(mc_context *c, const void *opt, int num_blocks, int num_threads, uint64_t paths_per_block, const void *dst, uint64_t *n)
{
    if (!c) return fail(MC_ERR_INVALID, "NULL context");
    if (!opt) return fail(MC_ERR_INVALID, "NULL option");
    if (!dst) return fail(MC_ERR_INVALID, "NULL output pointer");
    if (num_blocks < 1 || num_threads < 1 || num_threads > 1024 || (uint64_t)num_blocks * (uint64_t)num_threads > (1u << 24))
        return fail(MC_ERR_INVALID, "launch geometry: need num_blocks >= 1, 1 <= num_threads <= 1024, at most 2^24 threads in all");
    if (paths_per_block == 0 || paths_per_block > (1ull << 31) / (uint64_t)num_blocks)
        return fail(MC_ERR_INVALID, "launch geometry: need 1 <= num_blocks * paths_per_block <= 2^31");
    if (c->antithetic || c->control)
        return fail(MC_ERR_UNSUPPORTED, "launch geometry: the reference's plain estimator only");
    *n = (uint64_t)num_blocks * paths_per_block;
    return MC_OK;
}

// the (num_blocks x num_threads) start states: cached in the context for the last GRID_CACHE geometries used (the set-up is
// a 48-step GF(2) jump per thread, < 1 ms for the reference's 512 x 128; the jump matrices themselves are computed once per
// process, ~3 ms on the host).  All launch-geometry work runs on the context's own stream, so a cached array is never read
// by a kernel that started before it was filled.
static constexpr size_t GRID_CACHE = 4;
static int grid_states_ready(mc_context *c, int num_blocks, int num_threads, uint32_t sub, uint32_t step, const uint32_t **states)
{
    const uint32_t lanes = (uint32_t)num_blocks * (uint32_t)num_threads * sub;
    for (mc_context::GridStates &g : c->grid_cache)
        if (g.blocks == num_blocks && g.threads == num_threads && g.sub == sub && g.step == step) {
            g.last_used = ++c->grid_clock;
            *states = g.d;
            return MC_OK;
        }
    // a miss: set-up work of the call (the reference pays its randomSetup on EVERY call, dp/MonteCarloKernel.cu:315-323)
    StageTimer stage(&c->acc_setup_ms);
    if (int rc = xorwow_jump_ready(c)) return rc;
    if (c->grid_cache.size() >= GRID_CACHE) {   // evict the least recently used geometry (no kernel may still read it)
        HIPCHK(hipStreamSynchronize(c->stream));
        size_t lru = 0;
        for (size_t i = 1; i < c->grid_cache.size(); ++i)
            if (c->grid_cache[i].last_used < c->grid_cache[lru].last_used)
                lru = i;
        uint32_t *victim = c->grid_cache[lru].d;
        c->grid_cache.erase(c->grid_cache.begin() + (long)lru);   // out of the cache BEFORE it is freed: a failing hipFree leaves no dangling entry
        HIPCHK(hipFree(victim));
    }
    uint32_t *d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(uint32_t) * 6 * (size_t)lanes));
    xorwow_grid_init_kernel<<<(lanes + 255) / 256, 256, 0, c->stream>>>(c->d_xorwow_jump, (uint32_t)num_blocks, (uint32_t)num_threads, sub, step, d);
    if (hipGetLastError() != hipSuccess) {
        (void)hipFree(d);
        return fail(MC_ERR_HIP, "launch geometry: the state set-up kernel failed to launch");
    }
    c->grid_cache.push_back({num_blocks, num_threads, sub, step, d, ++c->grid_clock});
    *states = d;
    // the states are complete before the pricing kernel is enqueued, and a timed call's opening event is recorded again behind
    // them: kernel_ms is the pricing kernel's, the set-up is reported as set-up (it used to sit inside ev0..ev1)
    return setup_settle(c, c->stream);
}

extern "C" int mc_context_set_grid_form(mc_context *c, int form)
{
    if (!c || (form != MC_GRID_FORM_AUTO && form != MC_GRID_FORM_STAGED && form != MC_GRID_FORM_FUSED))
        return fail(MC_ERR_INVALID, "mc_context_set_grid_form: bad argument");
    c->grid_form = form;
    return MC_OK;
}

// ---- staged form (round 3): normals into HBM, then the engine's kernels through the external-normals policy -------------
// `draws` normals per path into rows of `row` Reals (the buffer zero-padded to `padded` Reals in all), then `enqueue` with
// the external-normals policy reading `per_unit` Reals per unit.  h_values: per-path dump (tests) instead of the estimate.
template <class Real, class Enq>
static int grid_run_staged(mc_context *c, int num_blocks, int num_threads, uint64_t paths_per_block, uint32_t draws, uint32_t row,
                           uint32_t per_unit, size_t padded, double discount, Real *h_values, mc_result *out, Enq enqueue)
{
    const uint64_t n = (uint64_t)num_blocks * paths_per_block;
    c->call_t0 = std::chrono::steady_clock::now(), c->call_t0_valid = !h_values;   // the normals pass below is part of the call's wall time
    c->acc_setup_ms = c->acc_table_ms = 0;
    HIPCHK(hipSetDevice(c->device));
    if (int rc = ensure_ext(c, padded * sizeof(Real))) return rc;
    const uint32_t *states = nullptr;
    if (int rc = grid_states_ready(c, num_blocks, num_threads, 1, 0, &states)) return rc;
    if (padded > n * row)
        HIPCHK(hipMemsetAsync((Real *)c->d_ext + n * row, 0, (padded - n * row) * sizeof(Real), c->stream));
    const uint32_t lanes = (uint32_t)num_blocks * (uint32_t)num_threads;
    grid_normals_kernel<Real><<<(lanes + 255) / 256, 256, 0, c->stream>>>(states, (uint32_t)num_blocks, (uint32_t)num_threads,
                                                                          paths_per_block, draws, row, (Real *)c->d_ext);
    HIPCHK(hipGetLastError());
    const int rng = c->rng, nf32 = c->normals_f32;
    c->rng = MC_RNG_PHILOX, c->normals_f32 = 0;   // the external policy replaces the generator whatever the context selects
    c->ext = c->d_ext, c->ext_per_unit = per_unit, c->ext_flags = 0;
    int rc;
    if (h_values)
        rc = dump_sync<Real>(c, n, h_values, [&](hipStream_t st, double *t, Real *d) { return enqueue(st, t, d); });
    else
        rc = run_sync(c, n, discount, out, [&](hipStream_t st, double *t) { return enqueue(st, t, (Real *)nullptr); });
    c->ext = nullptr, c->ext_per_unit = 0;
    c->rng = rng, c->normals_f32 = nf32;
    return rc;
}

// ---- fused form (round 4): the reference's launch itself, every thread's XORWOW stream in registers (mc_grid.hpp) -------
// launch(tail, work, geo, workgroups, workgroup size, stream, d_out) starts the product's grid kernel.  One (sum, sum2)
// pair per workgroup (reference block x piece), closed by the last arriver like every other call.
static bool grid_fused_fits(const mc_context *c, int num_blocks)
{
    return (uint64_t)num_blocks * 32 <= (uint64_t)MAX_SEGMENTS * (uint64_t)c->blocks * MAX_GRID_SCALE;   // the context's pair buffer (<= 32 pieces)
}
template <class Real, class Launch>
static int grid_run_fused(mc_context *c, int num_blocks, int num_threads, uint64_t paths_per_block, uint32_t draws, double scale1,
                          double scale2, double discount, Real *h_values, mc_result *out, Launch launch)
{
    const uint64_t n = (uint64_t)num_blocks * paths_per_block;
    HIPCHK(hipSetDevice(c->device));
    const auto enqueue = [&](hipStream_t st, double *d_triple, Real *d_out) -> int {
        if (int rc = begin_call(c, st)) return rc;
        GridGeom geo;
        grid_pieces(num_blocks, num_threads, paths_per_block, &geo.sub, &geo.seg);
        if ((uint64_t)geo.sub * geo.seg * draws >= (1ull << XORWOW_OFFSET_BITS)) {   // a piece's offset must fit the offset matrices: one piece
            geo.seg = (uint32_t)(((paths_per_block + (uint64_t)num_threads - 1) / (uint64_t)num_threads + GRID_SEG_ALIGN - 1) / GRID_SEG_ALIGN * GRID_SEG_ALIGN);
            geo.sub = 1;
        }
        if (int rc = grid_states_ready(c, num_blocks, num_threads, geo.sub, geo.sub > 1 ? geo.seg * draws : 0u, &geo.states)) return rc;
        geo.num_threads = (uint32_t)num_threads;
        geo.paths_per_block = (uint32_t)paths_per_block;
        Work w = make_work(0, Segment{0, 1u}, 0, 0);
        w.ext_per_unit = draws;
        const int groups = num_blocks * (int)geo.sub;     // one (sum, sum2) pair per workgroup
        const Tail t = make_tail(c, groups, scale1, scale2, n, d_triple);
        const int group = (num_threads + 63) / 64 * 64;   // whole waves; the lanes beyond num_threads idle
        c->last_grid = groups, c->last_group = group;
        if (int rc = launch(t, w, geo, groups, group, st, d_out)) return rc;
        return finish_call(c, t, groups, st);
    };
    if (h_values)
        return dump_sync<Real>(c, n, h_values, enqueue);
    return run_sync(c, n, discount, out, [&](hipStream_t st, double *t) { return enqueue(st, t, (Real *)nullptr); });
}

// which form a call takes: the context's choice, else fused where a fused kernel exists for the shape
static int grid_pick(mc_context *c, bool fused_possible, bool *fused)
{
    if (c->grid_form == MC_GRID_FORM_FUSED && !fused_possible)
        return fail(MC_ERR_UNSUPPORTED, "launch geometry: no fused kernel for this shape (baskets beyond 16 assets, more blocks than the pair "
                                        "buffer holds); the staged form covers it");
    *fused = c->grid_form == MC_GRID_FORM_STAGED ? false : fused_possible;
    return MC_OK;
}

// dates of a CVA path that draw a normal: `t -= dt >= 0` in Real arithmetic (dp/MonteCarloKernel.cu:249; the per-date
// table of build_cva_table stops at the same date)
template <class Real>
static uint32_t cva_draws(Real t, int n_grid)
{
    const Real dt = t / n_grid;
    uint32_t draws = 0;
    for (int j = 1; j <= n_grid; ++j) {
        t -= dt;
        if (!(t >= 0)) break;
        ++draws;
    }
    return draws;
}

extern "C" int mc_grid_normals(mc_context *c, int num_blocks, int num_threads, uint32_t count, float *h_out)
{
    uint64_t n;
    if (!h_out) return fail(MC_ERR_INVALID, "mc_grid_normals: NULL output pointer");
    if (int rc = grid_check(c, h_out, num_blocks, num_threads, 1, h_out, &n)) return rc;
    const uint64_t lanes = (uint64_t)num_blocks * (uint64_t)num_threads;
    if (count == 0 || lanes * count > (1ull << 28))
        return fail(MC_ERR_INVALID, "mc_grid_normals: need 1 <= threads * count <= 2^28");
    HIPCHK(hipSetDevice(c->device));
    if (int rc = ensure_ext(c, lanes * count * sizeof(float))) return rc;
    const uint32_t *states = nullptr;
    if (int rc = grid_states_ready(c, num_blocks, num_threads, 1, 0, &states)) return rc;
    // every thread as the only path of its own row: paths_per_block = num_threads, one path per thread, `count` draws
    grid_normals_kernel<float><<<((uint32_t)lanes + 255) / 256, 256, 0, c->stream>>>(states, (uint32_t)num_blocks, (uint32_t)num_threads,
                                                                                    (uint64_t)num_threads, count, count, (float *)c->d_ext);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_out, c->d_ext, lanes * count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MC_OK;
}

// the fused basket kernels are compiled for 4, 8 and 16 assets; a smaller basket runs the next size zero-padded
template <class Real, int NA>
static int grid_basket_fused(mc_context *c, const typename BasketIn<Real>::type *o, int nb, int nt, uint64_t ppb, Real *h_values, mc_result *out)
{
    BasketArgs<Real, NA> k;
    double out_scale = 1.0;
    if (int rc = basket_fold<Real, NA>(c, *o, k, out_scale)) return rc;
    return grid_run_fused<Real>(c, nb, nt, ppb, (uint32_t)o->n, out_scale, out_scale * out_scale, std::exp(-(double)o->r * (double)o->t), h_values, out,
                                [&](const Tail &t, const Work &w, const GridGeom &geo, int groups, int group, hipStream_t st, Real *d_out) -> int {
                                    if (d_out)
                                        grid_basket_kernel<Real, NA, true><<<groups, group, 0, st>>>(t, k, w, geo, d_out, (Real)out_scale);
                                    else
                                        grid_basket_kernel<Real, NA, false><<<groups, group, 0, st>>>(t, k, w, geo, d_out, (Real)out_scale);
                                    return MC_OK;
                                });
}

template <class Real>
static int grid_vanilla(mc_context *c, const typename VanillaTraits<Real>::In *o, int nb, int nt, uint64_t ppb, Real *h_values, mc_result *out)
{
    uint64_t n;
    if (int rc = grid_check(c, o, nb, nt, ppb, h_values ? (const void *)h_values : (const void *)out, &n)) return rc;
    bool fused;
    if (int rc = grid_pick(c, grid_fused_fits(c, nb), &fused)) return rc;
    const double disc = std::exp(-(double)o->r * (double)o->t);
    if (!fused) {
        const uint64_t NPB = GenPhilox::npb<Real>(), units = (n + NPB - 1) / NPB;
        /* a vanilla unit is NPB paths: rows of one normal, read NPB at a time */
        return grid_run_staged<Real>(c, nb, nt, ppb, 1, 1, (uint32_t)NPB, units * NPB, disc, h_values, out,
                                     [&](hipStream_t st, double *t, Real *d) { return vanilla_enqueue<Real>(c, o, 0, 0, n, t, st, d); });
    }
    typename VanillaTraits<Real>::Opt k;
    double scale1, scale2;
    if (int rc = VanillaTraits<Real>::prepare(*o, k, scale1, scale2)) return rc;
    return grid_run_fused<Real>(c, nb, nt, ppb, 1u, scale1, scale2, disc, h_values, out,
                                [&](const Tail &t, const Work &w, const GridGeom &geo, int groups, int group, hipStream_t st, Real *d_out) -> int {
                                    if constexpr (sizeof(Real) == 4) {
                                        if (d_out)
                                            grid_vanilla_f32_kernel<true><<<groups, group, 0, st>>>(t, k, w, geo, d_out, (float)scale1);
                                        else
                                            grid_vanilla_f32_kernel<false><<<groups, group, 0, st>>>(t, k, w, geo, d_out, (float)scale1);
                                    } else {
                                        if (d_out)
                                            grid_vanilla_f64_kernel<true><<<groups, group, 0, st>>>(t, k, w, geo, d_out, 1.0);
                                        else
                                            grid_vanilla_f64_kernel<false><<<groups, group, 0, st>>>(t, k, w, geo, d_out, 1.0);
                                    }
                                    return MC_OK;
                                });
}

template <class Real>
static int grid_basket(mc_context *c, const typename BasketIn<Real>::type *o, int nb, int nt, uint64_t ppb, Real *h_values, mc_result *out)
{
    uint64_t n;
    if (int rc = grid_check(c, o, nb, nt, ppb, h_values ? (const void *)h_values : (const void *)out, &n)) return rc;
    if (o->n < 1 || o->n > MC_MAX_ASSETS_GENERIC) return fail(MC_ERR_INVALID, "basket: bad n");
    if (!o->s || !o->v || !o->p || !o->d || !o->w) return fail(MC_ERR_INVALID, "basket: NULL array");
    if (!(o->t >= 0) || !std::isfinite((double)o->r) || !std::isfinite((double)o->k)) return fail(MC_ERR_INVALID, "basket: need t>=0 and finite r, k");
    bool fused;
    if (int rc = grid_pick(c, grid_fused_fits(c, nb) && o->n <= 16, &fused)) return rc;
    if (!fused)
        return grid_run_staged<Real>(c, nb, nt, ppb, (uint32_t)o->n, (uint32_t)o->n, (uint32_t)o->n, ((size_t)n + 1) * (size_t)o->n,
                                     std::exp(-(double)o->r * (double)o->t), h_values, out,
                                     [&](hipStream_t st, double *t, Real *d) { return basket_enqueue<Real>(c, o, 0, 0, n, t, st, d); });
    if (o->n <= 4) return grid_basket_fused<Real, 4>(c, o, nb, nt, ppb, h_values, out);
    if (o->n <= 8) return grid_basket_fused<Real, 8>(c, o, nb, nt, ppb, h_values, out);
    return grid_basket_fused<Real, 16>(c, o, nb, nt, ppb, h_values, out);
}

template <class Real>
static int grid_cva(mc_context *c, const typename CvaIn<Real>::type *o, int nb, int nt, uint64_t ppb, Real *h_values, mc_result *out)
{
    uint64_t n;
    if (int rc = grid_check(c, o, nb, nt, ppb, h_values ? (const void *)h_values : (const void *)out, &n)) return rc;
    if (o->n_grid < 1 || o->n_grid > (1 << 20) || !(o->option.t > 0))
        return fail(MC_ERR_INVALID, "cva: bad n_grid or maturity");
    bool fused;
    if (int rc = grid_pick(c, grid_fused_fits(c, nb), &fused)) return rc;
    const uint32_t draws = cva_draws<Real>(o->option.t, o->n_grid);
    if (!fused)
        return grid_run_staged<Real>(c, nb, nt, ppb, draws, (uint32_t)o->n_grid, (uint32_t)o->n_grid, (size_t)n * (size_t)o->n_grid, 1.0, h_values, out,
                                     [&](hipStream_t st, double *t, Real *d) { return cva_enqueue<Real>(c, o, 0, 0, n, t, st, d); });
    return grid_run_fused<Real>(c, nb, nt, ppb, draws, 1.0, 1.0, 1.0, h_values, out,
                                [&](const Tail &t, const Work &w, const GridGeom &geo, int groups, int group, hipStream_t st, Real *d_out) -> int {
                                    CvaArgs<Real> args;
                                    if (int rc = cva_table_ready<Real>(c, o, st, args)) return rc;
                                    if (d_out)
                                        grid_cva_kernel<Real, true><<<groups, group, 0, st>>>(t, args, w, geo, d_out);
                                    else
                                        grid_cva_kernel<Real, false><<<groups, group, 0, st>>>(t, args, w, geo, d_out);
                                    return MC_OK;
                                });
}

#define MC_DEFINE_GRID(X, Real)                                                                                                   \
    extern "C" int mc_vanilla_run_grid_##X(mc_context *c, const mc_option_##X *o, int num_blocks, int num_threads,               \
                                           uint64_t paths_per_block, mc_result *out)                                             \
    {                                                                                                                             \
        if (!out) return fail(MC_ERR_INVALID, "NULL output pointer");                                                             \
        return grid_vanilla<Real>(c, o, num_blocks, num_threads, paths_per_block, (Real *)nullptr, out);                         \
    }                                                                                                                             \
    extern "C" int mc_basket_run_grid_##X(mc_context *c, const mc_basket_##X *o, int num_blocks, int num_threads,                \
                                          uint64_t paths_per_block, mc_result *out)                                              \
    {                                                                                                                             \
        if (!out) return fail(MC_ERR_INVALID, "NULL output pointer");                                                             \
        if (!o) return fail(MC_ERR_INVALID, "NULL option");                                                                       \
        return grid_basket<Real>(c, o, num_blocks, num_threads, paths_per_block, (Real *)nullptr, out);                          \
    }                                                                                                                             \
    extern "C" int mc_cva_run_grid_##X(mc_context *c, const mc_cva_##X *o, int num_blocks, int num_threads,                      \
                                       uint64_t paths_per_block, mc_result *out)                                                 \
    {                                                                                                                             \
        if (!out) return fail(MC_ERR_INVALID, "NULL output pointer");                                                             \
        if (!o) return fail(MC_ERR_INVALID, "NULL option");                                                                       \
        return grid_cva<Real>(c, o, num_blocks, num_threads, paths_per_block, (Real *)nullptr, out);                             \
    }                                                                                                                             \
    /* per-path values of a launch-geometry call, in the call's path order (block-major): test hooks, mc_mi355x_test.h */         \
    extern "C" int mc_vanilla_paths_grid_##X(mc_context *c, const mc_option_##X *o, int num_blocks, int num_threads,             \
                                             uint64_t paths_per_block, Real *h_out)                                              \
    {                                                                                                                             \
        if (!h_out) return fail(MC_ERR_INVALID, "NULL output pointer");                                                           \
        return grid_vanilla<Real>(c, o, num_blocks, num_threads, paths_per_block, h_out, (mc_result *)nullptr);                  \
    }                                                                                                                             \
    extern "C" int mc_basket_paths_grid_##X(mc_context *c, const mc_basket_##X *o, int num_blocks, int num_threads,              \
                                            uint64_t paths_per_block, Real *h_out)                                               \
    {                                                                                                                             \
        if (!h_out) return fail(MC_ERR_INVALID, "NULL output pointer");                                                           \
        if (!o) return fail(MC_ERR_INVALID, "NULL option");                                                                       \
        return grid_basket<Real>(c, o, num_blocks, num_threads, paths_per_block, h_out, (mc_result *)nullptr);                   \
    }                                                                                                                             \
    extern "C" int mc_cva_paths_grid_##X(mc_context *c, const mc_cva_##X *o, int num_blocks, int num_threads,                    \
                                         uint64_t paths_per_block, Real *h_out)                                                  \
    {                                                                                                                             \
        if (!h_out) return fail(MC_ERR_INVALID, "NULL output pointer");                                                           \
        if (!o) return fail(MC_ERR_INVALID, "NULL option");                                                                       \
        return grid_cva<Real>(c, o, num_blocks, num_threads, paths_per_block, h_out, (mc_result *)nullptr);                      \
    }

#define MC_DEFINE_PRODUCT(X, Real)                                                                           \
    extern "C" int mc_vanilla_launch_##X(mc_context *c, const mc_option_##X *o, uint64_t seed, uint64_t first, \
                                         uint64_t n, double *d_triple, void *stream)                         \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, d_triple)) return rc;                                      \
        ArmScope arm(c);                                                                                     \
        return vanilla_enqueue<Real>(c, o, seed, first, n, d_triple, pick_stream(c, stream), nullptr);       \
    }                                                                                                        \
    extern "C" int mc_basket_launch_##X(mc_context *c, const mc_basket_##X *o, uint64_t seed, uint64_t first, \
                                        uint64_t n, double *d_triple, void *stream)                          \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, d_triple)) return rc;                                      \
        ArmScope arm(c);                                                                                     \
        return basket_enqueue<Real>(c, o, seed, first, n, d_triple, pick_stream(c, stream), nullptr);        \
    }                                                                                                        \
    extern "C" int mc_cva_launch_##X(mc_context *c, const mc_cva_##X *o, uint64_t seed, uint64_t first,      \
                                     uint64_t n, double *d_triple, void *stream)                             \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, d_triple)) return rc;                                      \
        ArmScope arm(c);                                                                                     \
        return cva_enqueue<Real>(c, o, seed, first, n, d_triple, pick_stream(c, stream), nullptr);           \
    }                                                                                                        \
    extern "C" int mc_vanilla_run_##X(mc_context *c, const mc_option_##X *o, uint64_t seed, uint64_t first,  \
                                      uint64_t n, mc_result *out)                                            \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, out)) return rc;                                           \
        return run_sync(c, n, std::exp(-(double)o->r * (double)o->t), out, [&](hipStream_t st, double *t) {  \
            return vanilla_enqueue<Real>(c, o, seed, first, n, t, st, nullptr);                              \
        });                                                                                                  \
    }                                                                                                        \
    extern "C" int mc_basket_run_##X(mc_context *c, const mc_basket_##X *o, uint64_t seed, uint64_t first,   \
                                     uint64_t n, mc_result *out)                                             \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, out)) return rc;                                           \
        const double disc = std::exp(-(double)o->r * (double)o->t);                                          \
        if (int rc = run_sync(c, n, disc, out, [&](hipStream_t st, double *t) {                              \
                return basket_enqueue<Real>(c, o, seed, first, n, t, st, nullptr);                           \
            }))                                                                                              \
            return rc;                                                                                       \
        if (c->control) { /* the simulated quantity was payoff - control: add the control's closed-form mean */ \
            double cv_mean;                                                                                  \
            if (int rc = control_mean(*o, &cv_mean)) return rc;                                              \
            out->expected += disc * cv_mean;                                                                 \
        }                                                                                                    \
        return MC_OK;                                                                                        \
    }                                                                                                        \
    extern "C" int mc_cva_run_##X(mc_context *c, const mc_cva_##X *o, uint64_t seed, uint64_t first,         \
                                  uint64_t n, mc_result *out)                                                \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, out)) return rc;                                           \
        return run_sync(c, n, 1.0, out, [&](hipStream_t st, double *t) {                                     \
            return cva_enqueue<Real>(c, o, seed, first, n, t, st, nullptr);                                  \
        });                                                                                                  \
    }                                                                                                        \
    extern "C" int mc_vanilla_paths_##X(mc_context *c, const mc_option_##X *o, uint64_t seed, uint64_t first, \
                                        uint64_t n, Real *h_out)                                             \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, h_out)) return rc;                                         \
        return dump_sync<Real>(c, n, h_out, [&](hipStream_t st, double *t, Real *d) {                        \
            return vanilla_enqueue<Real>(c, o, seed, first, n, t, st, d);                                    \
        });                                                                                                  \
    }                                                                                                        \
    extern "C" int mc_basket_paths_##X(mc_context *c, const mc_basket_##X *o, uint64_t seed, uint64_t first, \
                                       uint64_t n, Real *h_out)                                              \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, h_out)) return rc;                                         \
        return dump_sync<Real>(c, n, h_out, [&](hipStream_t st, double *t, Real *d) {                        \
            return basket_enqueue<Real>(c, o, seed, first, n, t, st, d);                                     \
        });                                                                                                  \
    }                                                                                                        \
    extern "C" int mc_cva_paths_##X(mc_context *c, const mc_cva_##X *o, uint64_t seed, uint64_t first,       \
                                    uint64_t n, Real *h_out)                                                 \
    {                                                                                                        \
        if (int rc = check_common(c, o, first, n, h_out)) return rc;                                         \
        return dump_sync<Real>(c, n, h_out, [&](hipStream_t st, double *t, Real *d) {                        \
            return cva_enqueue<Real>(c, o, seed, first, n, t, st, d);                                        \
        });                                                                                                  \
    }                                                                                                        \
    extern "C" int mc_normals_##X(mc_context *c, uint64_t seed, uint32_t domain, uint64_t first_unit,        \
                                  uint64_t n_units, uint32_t block, Real *h_out)                             \
    {                                                                                                        \
        if (int rc = check_common(c, h_out, first_unit, n_units, h_out)) return rc;                          \
        const uint64_t NPB = npb_of<Real>(c);                                                                \
        return dump_sync<Real>(c, n_units * NPB, h_out, [&](hipStream_t st, double *, Real *d) -> int {      \
            std::vector<Segment> segs;                                                                       \
            if (int rc = plan_segments(first_unit, n_units, segs)) return rc;                                \
            uint64_t done = 0;                                                                               \
            for (const Segment &s : segs) {                                                                  \
                const Work w = make_work(seed, s, 0, 0);                                                     \
                if (sizeof(Real) == 8 && c->normals_f32)                                                     \
                    normals_kernel<Real, GenPhiloxF32N><<<grid_for(c->blocks, s.count), GROUP, 0, st>>>(w, block, domain, d + done * NPB); \
                else                                                                                         \
                    normals_kernel<Real><<<grid_for(c->blocks, s.count), GROUP, 0, st>>>(w, block, domain, d + done * NPB); \
                done += s.count;                                                                             \
            }                                                                                                \
            HIPCHK(hipGetLastError());                                                                       \
            return MC_OK;                                                                                    \
        });                                                                                                  \
    }

MC_DEFINE_PRODUCT(f32, float)
MC_DEFINE_PRODUCT(f64, double)
MC_DEFINE_FROM_NORMALS(f32, float)
MC_DEFINE_GRID(f32, float)
MC_DEFINE_FROM_NORMALS(f64, double)
MC_DEFINE_GRID(f64, double)

// A query of one kernel's attributes makes the runtime load the library's whole code object for the current device.  The kernel
// named is a plain (non-template) one that xorwow_fill above already launches: naming it again instantiates nothing and moves nothing.
static hipError_t preload_code_object()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&xorwow_init_kernel));
}
