// mc_multi.cpp -- libmc_multi.so: one pricing call over several MI355X from one host process (include/mc_multi.h).
//
// A client of the single-device C ABI (include/mc_mi355x.h: one mc_context per device, mc_*_launch_* on that
// device's stream) plus RCCL's C API, linked directly (librccl.so; no PyTorch anywhere near this file).  The
// reference has no counterpart: it is single-device, default stream, synchronous; the entry points
// fanned out here are dp/MonteCarloKernel.cu:483 (basket), :500 (vanilla), :517 (CVA).
//
// Host-only translation unit: no device code.  Everything a call enqueues -- G launches, one grouped all-reduce, one
// one-lane publish kernel -- is asynchronous; the results come back through pinned host memory that the devices write
// themselves and the calling thread polls (run_sharded).  With G > 1 every device's launch is issued by that device's
// own launcher thread (mc_multi_host.hpp: LaunchCrew), so the devices start together; the collective stays ONE grouped
// call on the calling thread.  NOTE: the grouped all-reduce has only ever run with a communicator of ONE rank (one-GPU
// test boxes; RCCL refuses a repeated device): the G > 1 collective is unexercised.  The G > 1 CALL SEQUENCE of this file
// (buffers, streams and communicators per rank, publish, cross-check) does run in the GPU suite, with three ranks on one
// device against a test double of the six RCCL entry points (tests/cpp/rccl_mock.hip, MC_MULTI_ALLOW_REPEATED_DEVICES=1).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mc_multi.h"
#include "mc_multi_host.hpp"

static thread_local std::string g_multi_error;

extern "C" const char *mc_multi_last_error(void) { return g_multi_error.c_str(); }

static int fail(int code, const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_multi_error = buf;
    return code;
}

#define HIPCHK(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(MC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NCCLCHK(call)                                                                                      \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return fail(MC_ERR_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
// a failing single-device call: pass its status and its own text on
#define MCCHK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != MC_OK)                                                  \
            return fail(rc_, "device %d: %s", m->devices[g], mc_last_error()); \
    } while (0)

constexpr int MAX_DEVICES = 64;   // mc_multi_create's limit

struct mc_multi {
    std::vector<int> devices;
    std::vector<mc_context *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<double *> d_send, d_recv;      // per device: its own triple, the all-reduced triple
    std::vector<hipEvent_t> ev0, ev1;
    std::vector<ncclComm_t> comm;              // created on first use (ncclCommInitAll), kept for the handle's life
    double *h_send = nullptr;                  // pinned, 3 doubles per device
    double *h_recv = nullptr;                  // pinned, 3 doubles
    int reduce = MC_REDUCE_RCCL;
    bool control = false;
    bool timing = true;                        // two HIP events per device and call (mc_result.kernel_ms)
    bool readback_copy = false;                // MC_MULTI_READBACK=copy: round 2's copy + synchronize read-back also with timing off (A/B)
    double last_reduce_error = 0.0;
    // one launcher thread per device (G > 1; MC_MULTI_THREADS=0 keeps the serial fan-out of rounds 2-3 for A/B)
    std::unique_ptr<mc_host::LaunchCrew> crew;
    // the job of the call in progress lives HERE, not on run_sharded's stack: a launcher thread that comes back from a launch after
    // its call has timed out (LaunchCrew::run_all) still reads it
    alignas(16) unsigned char job_storage[512];
    const volatile double *slot_storage[MAX_DEVICES + 1] = {};   // (the same goes for the call's pinned-slot addresses)
    std::vector<std::string> worker_error;     // text of a worker's failed launch (mc_last_error is per thread)
    double last_fanout_us = 0.0;               // call entry -> the last device's launch enqueued, of the last call
    double last_collective_us = -1.0;          // last device triple seen on the host -> all-reduced triple seen, of the last call (-1: not measured)
    double last_device_us[MAX_DEVICES] = {};   // call entry -> device g's triple seen on the host, of the last call
    double last_seen_us[MAX_DEVICES] = {};     // ... per device: when its launcher thread saw the call (-1: the caller ran it because the
    double last_at_us[MAX_DEVICES] = {};       //     thread was late, -2: because it was parked; serial fan-out: -3), when its launch was enqueued
    long linger_us = 0;                        // resolved configuration, for mc_multi_describe
    int cpus = 0;
    const char *threads_why = "";
    bool trace = false;                        // MC_MULTI_TRACE=1: one stderr line per call with every device's hand-off and enqueue time
};

// The calling thread's current HIP device is the caller's business: every entry point that visits the handle's devices
// (creation, destruction, a sharded run -- serially, or when the caller takes over a late launcher thread's job) puts it
// back on the way out.  The launcher threads have their own current device and never touch the caller's.
struct CallerDevice {
    int prev = -1;
    CallerDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~CallerDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
    CallerDevice(const CallerDevice &) = delete;
    CallerDevice &operator=(const CallerDevice &) = delete;
};

extern "C" void mc_multi_destroy(mc_multi *m)
{
    if (!m)
        return;
    CallerDevice keep;
    if (m->crew && m->crew->broken()) {
        // a launcher thread never returned from a launch it had claimed (run_sharded reported MC_ERR_HIP): it may still be inside the
        // HIP runtime with pointers into this handle -- nothing it could touch is freed; the threads are detached and the handle leaked
        mc_host::LaunchCrew::retire(m->crew);
        return;
    }
    m->crew.reset();   // joins the launcher threads: none of them is inside a launch after this
    for (size_t g = 0; g < m->ctx.size(); ++g) {
        (void)hipSetDevice(m->devices[g]);
        if (g < m->stream.size() && m->stream[g]) (void)hipStreamSynchronize(m->stream[g]);
        if (g < m->comm.size() && m->comm[g]) (void)ncclCommDestroy(m->comm[g]);
        if (g < m->d_send.size()) (void)hipFree(m->d_send[g]);
        if (g < m->d_recv.size()) (void)hipFree(m->d_recv[g]);
        if (g < m->ev0.size() && m->ev0[g]) (void)hipEventDestroy(m->ev0[g]);
        if (g < m->ev1.size() && m->ev1[g]) (void)hipEventDestroy(m->ev1[g]);
        mc_context_destroy(m->ctx[g]);
    }
    (void)hipHostFree(m->h_send);
    (void)hipHostFree(m->h_recv);
    delete m;
}

static int multi_allocate(mc_multi *m, int blocks)
{
    const int G = (int)m->devices.size();
    for (int g = 0; g < G; ++g) {
        mc_context *c = nullptr;
        MCCHK(mc_context_create(m->devices[g], blocks, &c));
        m->ctx.push_back(c);
        m->stream.push_back((hipStream_t)mc_context_stream(c));
        HIPCHK(hipSetDevice(m->devices[g]));
        double *s = nullptr, *r = nullptr;
        HIPCHK(hipMalloc(&s, 3 * sizeof(double)));
        m->d_send.push_back(s);
        HIPCHK(hipMalloc(&r, 3 * sizeof(double)));
        m->d_recv.push_back(r);
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(hipEventCreate(&a));
        m->ev0.push_back(a);
        HIPCHK(hipEventCreate(&b));
        m->ev1.push_back(b);
    }
    HIPCHK(hipHostMalloc(&m->h_send, 3 * sizeof(double) * G, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&m->h_recv, 3 * sizeof(double), hipHostMallocDefault));
    return MC_OK;
}

extern "C" int mc_multi_create(const int *devices, int n_devices, int blocks, mc_multi **out)
{
    if (!out)
        return fail(MC_ERR_INVALID, "mc_multi_create: out is NULL");
    *out = nullptr;
    CallerDevice keep;
    const int visible = mc_device_count();
    if (visible <= 0)
        return fail(MC_ERR_NO_DEVICE, "no HIP device visible (the HIP engine has no CPU fallback)");
    if (!devices && n_devices <= 0)
        n_devices = visible;
    if (n_devices <= 0 || n_devices > MAX_DEVICES)
        return fail(MC_ERR_INVALID, "mc_multi_create: n_devices=%d", n_devices);
    mc_multi *m = new mc_multi;
    for (int g = 0; g < n_devices; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= visible) {
            delete m;
            return fail(MC_ERR_INVALID, "mc_multi_create: device %d out of range [0,%d)", d, visible);
        }
        m->devices.push_back(d);
    }
    if (const char *e = getenv("MC_MULTI_REDUCE"))
        m->reduce = strcmp(e, "host") == 0 ? MC_REDUCE_HOST : MC_REDUCE_RCCL;
    if (const char *e = getenv("MC_MULTI_READBACK"))
        m->readback_copy = strcmp(e, "copy") == 0;
    if (int rc = multi_allocate(m, blocks)) {
        mc_multi_destroy(m);
        return rc;
    }
    // Launcher threads: device g's launches are issued by thread g (its own hipSetDevice, arm and launch), started through
    // one call-number word the crew watches, so that device G-1 starts with device 0 instead of (G-1) x ~4 us later.
    // A worker spins for MC_MULTI_LINGER_US after its last job (default 5000: the 13 us the threads save are worth having on
    // calls of up to a few ms -- 0.3 % of a 5 ms call -- and calls shorter than the linger time that follow each other never
    // find a worker asleep), then sleeps; a sleeping worker's job is run by the caller at once and the sleepers are woken, after
    // the fan-out, only when calls come within the linger time of each other.  (Round 4's default was 100 000: a handle called
    // every < 100 ms kept G cores spinning for good -- ADVICE r04.)
    // No threads at all when the process cannot keep G + 1 threads running (affinity mask capped by the cgroup CPU quota):
    // spinners that exhaust the quota get the whole process throttled; MC_MULTI_THREADS=1 forces them, =0 forbids them.
    m->trace = getenv("MC_MULTI_TRACE") && atoi(getenv("MC_MULTI_TRACE")) != 0;
    const char *th = getenv("MC_MULTI_THREADS");
    const char *lg = getenv("MC_MULTI_LINGER_US");
    m->linger_us = lg ? atol(lg) : 5000;
    if (m->linger_us < 0)
        m->linger_us = 0;
    m->cpus = mc_host::cpus_allowed();
    const bool forced = th && atoi(th) > 0, forbidden = th && atoi(th) == 0;
    if (n_devices <= 1)
        m->threads_why = "one device";
    else if (forbidden)
        m->threads_why = "MC_MULTI_THREADS=0";
    else if (!forced && m->cpus < n_devices + 1)
        m->threads_why = "fewer CPUs granted (affinity mask, cgroup quota) than devices + 1";
    else {
        m->threads_why = forced ? "MC_MULTI_THREADS" : "default";
        m->worker_error.resize((size_t)n_devices);
        m->crew.reset(new mc_host::LaunchCrew(
            n_devices, std::chrono::microseconds(m->linger_us),
            [](void *h, int g) { (void)hipSetDevice(static_cast<mc_multi *>(h)->devices[(size_t)g]); }, m,
            std::chrono::microseconds(15), m->cpus));
    }
    if (const char *v = getenv("MC_VERBOSE"))
        if (atoi(v) >= 2) {
            char buf[1024];
            mc_multi_describe(m, buf, (int)sizeof buf);
            fprintf(stderr, "%s\n", buf);
        }
    *out = m;
    return MC_OK;
}

extern "C" int mc_multi_size(const mc_multi *m) { return m ? (int)m->devices.size() : 0; }
extern "C" mc_context *mc_multi_context(mc_multi *m, int i)
{
    return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr;
}
extern "C" double mc_multi_last_reduce_error(const mc_multi *m) { return m ? m->last_reduce_error : 0.0; }
extern "C" double mc_multi_last_fanout_us(const mc_multi *m) { return m ? m->last_fanout_us : 0.0; }
extern "C" double mc_multi_last_collective_us(const mc_multi *m) { return m ? m->last_collective_us : -1.0; }
extern "C" int mc_multi_last_device_us(const mc_multi *m, int cap, double *delivered_us)
{
    if (!m || !delivered_us || cap < 0)
        return fail(MC_ERR_INVALID, "mc_multi_last_device_us: bad argument");
    const int G = (int)m->devices.size();
    for (int g = 0; g < G && g < cap; ++g)
        delivered_us[g] = m->last_device_us[g];
    return G;
}
extern "C" int mc_multi_last_fanout_trace(const mc_multi *m, int cap, double *seen_us, double *enqueued_us)
{
    if (!m)
        return 0;
    const int G = (int)m->devices.size();
    for (int g = 0; g < G && g < cap; ++g) {
        if (seen_us) seen_us[g] = m->last_seen_us[g];
        if (enqueued_us) enqueued_us[g] = m->last_at_us[g];
    }
    return G;
}
extern "C" int mc_multi_fanout_stats(const mc_multi *m, mc_multi_fanout_counts *out)
{
    if (!m || !out)
        return fail(MC_ERR_INVALID, "mc_multi_fanout_stats: NULL argument");
    *out = mc_multi_fanout_counts{};
    if (m->crew) {
        const mc_host::LaunchCrew::Stats &s = m->crew->stats();
        out->calls = s.calls, out->by_worker = s.by_worker, out->served_parked = s.served_parked, out->stolen = s.stolen;
        out->slow_claimed = s.slow_claimed, out->wakeups = s.wakeups;
    }
    return MC_OK;
}
// The resolved configuration of a handle as one line of text (MC_VERBOSE=2 prints it at creation).
extern "C" int mc_multi_describe(const mc_multi *m, char *buf, int len)
{
    if (!m || !buf || len <= 0)
        return fail(MC_ERR_INVALID, "mc_multi_describe: bad argument");
    int k = snprintf(buf, (size_t)len, "mc_multi config: devices=[");
    for (size_t g = 0; g < m->devices.size() && k < len; ++g)
        k += snprintf(buf + k, (size_t)(len - k), "%s%d", g ? "," : "", m->devices[g]);
    if (k < len)
        snprintf(buf + k, (size_t)(len - k), "] reduce=%s readback=%s launcher_threads=%d (%s) linger_us=%ld cpus_allowed=%d (mask %d, cgroup quota %d) "
                 "spin=%s trace=%d",
                 m->reduce == MC_REDUCE_RCCL ? "rccl" : "host", m->readback_copy ? "copy" : "pinned-slot", m->crew ? m->crew->size() : 0, m->threads_why,
                 m->linger_us, m->cpus, mc_host::cpus_in_affinity_mask(), mc_host::cgroup_cpu_quota(),
                 m->crew ? (m->crew->yields() ? "yield" : "pause") : "none", (int)m->trace);
    return MC_OK;
}
extern "C" int mc_multi_launcher_threads(const mc_multi *m) { return (m && m->crew) ? m->crew->size() : 0; }

extern "C" int mc_multi_set_antithetic(mc_multi *m, int on)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    for (mc_context *c : m->ctx)
        mc_context_set_antithetic(c, on);
    return MC_OK;
}
extern "C" int mc_multi_set_control_variate(mc_multi *m, int on)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    for (mc_context *c : m->ctx)
        mc_context_set_control_variate(c, on);
    m->control = on != 0;
    return MC_OK;
}
extern "C" int mc_multi_set_normals(mc_multi *m, int mode)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    for (size_t g = 0; g < m->ctx.size(); ++g)
        if (mc_context_set_normals(m->ctx[g], mode) != MC_OK)
            return fail(MC_ERR_INVALID, "device %d: %s", m->devices[g], mc_last_error());
    return MC_OK;
}
// XORWOW runs one sequence per lane: device g's lanes take the subsequences after those of devices 0 .. g-1, so no two
// lanes of the job share one (with Philox there is nothing to partition: the counter is the global path index).
extern "C" int mc_multi_set_generator(mc_multi *m, int generator, uint64_t subsequence_base)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    uint64_t base = subsequence_base;
    for (size_t g = 0; g < m->ctx.size(); ++g) {
        if (mc_context_set_generator(m->ctx[g], generator, base) != MC_OK)
            return fail(MC_ERR_INVALID, "device %d: %s", m->devices[g], mc_last_error());
        base += (uint64_t)mc_context_blocks(m->ctx[g]) * 256u;
    }
    return MC_OK;
}

// Device timing of the calls (mc_result.kernel_ms = the slowest device's kernels).  Off: no event records -- two fewer
// runtime calls per device on the launching thread, which starts device g's work that much sooner after device 0's.
extern "C" int mc_multi_set_timing(mc_multi *m, int on)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    m->timing = on != 0;
    return MC_OK;
}

extern "C" int mc_multi_set_reduce(mc_multi *m, int mode)
{
    if (!m || (mode != MC_REDUCE_RCCL && mode != MC_REDUCE_HOST))
        return fail(MC_ERR_INVALID, "mc_multi_set_reduce: bad argument");
    m->reduce = mode;
    return MC_OK;
}

// RCCL communicators, one per device, all in this process: created by the first call that needs them.
static int ensure_comms(mc_multi *m)
{
    if (!m->comm.empty())
        return MC_OK;
    const int G = (int)m->devices.size();
    bool allow_repeated = false;
#ifdef MC_MULTI_TEST_HOOKS
    // Test build only (make libmc_multi_testhooks): MC_MULTI_ALLOW_REPEATED_DEVICES=1 is for the test double of the collective
    // (tests/cpp/rccl_mock.hip, preloaded in front of librccl.so), whose ranks may share a device; RCCL itself fails such a list
    // in ncclCommInitAll.  The shipped library has no such switch.
    if (const char *rep = getenv("MC_MULTI_ALLOW_REPEATED_DEVICES"))
        allow_repeated = atoi(rep) != 0;
#endif
    for (int a = 0; a < G && !allow_repeated; ++a)
        for (int b = a + 1; b < G; ++b)
            if (m->devices[a] == m->devices[b])
                return fail(MC_ERR_INVALID, "device %d is listed twice: RCCL needs distinct devices (MC_REDUCE_HOST accepts the list)",
                            m->devices[a]);
    std::vector<ncclComm_t> comm((size_t)G, nullptr);
    NCCLCHK(ncclCommInitAll(comm.data(), G, m->devices.data()));
    m->comm = comm;
    return MC_OK;
}

// What a call has put in flight, so that EVERY way out of run_sharded -- error returns included -- leaves the handle
// reusable: an open RCCL group is closed, every stream the call touched is drained (its kernels write d_send / the
// pinned slots, which the next call reuses).
struct InFlight {
    mc_multi *m;
    int touched = 0;        // streams 0 .. touched-1 have work of this call
    bool group_open = false;
    bool settled = false;
    explicit InFlight(mc_multi *h) : m(h) {}
    void settle()
    {
        if (settled)
            return;
        settled = true;
        if (group_open)
            (void)ncclGroupEnd();
        for (int g = 0; g < touched; ++g) {
            (void)hipSetDevice(m->devices[g]);
            (void)hipStreamSynchronize(m->stream[g]);
        }
    }
    ~InFlight() { settle(); }
};

// launch(g, ctx, first, count, d_triple, stream) enqueues device g's shard.
//
// Fan-out.  G == 1 (or MC_MULTI_THREADS=0): the calling thread enqueues every device's launch itself, one after the
// other -- an asynchronous launch costs it ~4.2 us (profiles/r02_launch_cost.log), so device 7 of 8 started ~30 us after
// device 0: 3 % of C5's 1 ms shard.  G > 1: every device has its own launcher thread (LaunchCrew), the calling thread
// bumps the crew's call number and the launches are issued concurrently; `last_fanout_us` = call entry -> the last device's launch
// enqueued (tools/c/multi_cost.c prints it).
//
// Read-back.  With timing off (what the legacy symbols and the benchmarks use) nothing is copied and nothing sleeps:
// every device's last workgroup stores its triple into a pinned host slot of its context (mc_context_arm_direct), the
// all-reduced triple follows through a one-lane kernel behind the collective on device 0's stream
// (mc_context_publish), and this thread polls the G (+ 1) flag words in ONE loop from user space
// (mc_multi_host.hpp: poll_slots).  Round 2 issued G + 1 hipMemcpyAsync and G serial hipStreamSynchronize calls
// instead (kept as MC_MULTI_READBACK=copy, the conservative fallback, and whenever timing is on: it is the form that
// can report kernel_ms).
template <class Launch>
static int run_sharded(mc_multi *m, uint64_t first, uint64_t n, double discount, double add_back, mc_result *out, Launch launch)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    if (!out) return fail(MC_ERR_INVALID, "NULL output pointer");
    if (n == 0) return fail(MC_ERR_INVALID, "n_paths == 0");
    const int G = (int)m->devices.size();
    if (m->crew && m->crew->broken())
        return fail(MC_ERR_HIP, "this handle is unusable: a launcher thread never returned from a launch of an earlier call "
                                "(see that call's error); destroy it -- the device it hung on is probably lost");
    CallerDevice keep;   // declared before InFlight: the drain of a failed call runs first, then the device goes back
    if (m->reduce == MC_REDUCE_RCCL)
        if (int rc = ensure_comms(m)) return rc;
    const auto wall0 = std::chrono::steady_clock::now();
    InFlight fl(m);
    bool direct = !m->timing && !m->readback_copy;
    // per-device scratch of the call in the handle (mc_multi_create admits at most 64 devices): no allocation between the
    // call's entry and the first launch, and nothing a late launcher thread could touch lives on this stack frame
    const volatile double **slot = m->slot_storage;   // [G] = the all-reduced triple's slot
    for (int g = 0; g <= G; ++g)
        slot[g] = nullptr;
    // device g's part of the call; runs on the calling thread or on launcher thread g.  Returns an MC_* status; the text
    // of a failure goes to `err` (mc_last_error is thread-local: a worker's text would be lost otherwise).
    struct Job {
        mc_multi *m;
        uint64_t first, n;
        bool want_direct;
        const volatile double **slot;
        Launch launch;            // by value (the lambdas capture the option pointer and the seed by value)
        char armed[MAX_DEVICES];
    };
    static_assert(sizeof(Job) <= sizeof m->job_storage && alignof(Job) <= 16 && std::is_trivially_destructible<Job>::value, "job_storage holds the call's job");
    Job &job = *new (m->job_storage) Job{m, first, n, direct, slot, launch, {}};
    const auto device_part = [](void *jp, int g) -> int {
        Job &j = *static_cast<Job *>(jp);
        mc_multi *m = j.m;
        const int G = (int)m->devices.size();
        std::string *err = m->worker_error.empty() ? nullptr : &m->worker_error[(size_t)g];
        const auto hip_fail = [&](const char *what, hipError_t e) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s failed: %s", what, hipGetErrorString(e));
            if (err) *err = buf; else g_multi_error = buf;
            return MC_ERR_HIP;
        };
        uint64_t lo = 0, cnt = 0;
        mc_shard_range(j.n, g, G, &lo, &cnt);
        hipError_t e = hipSetDevice(m->devices[(size_t)g]);
        if (e != hipSuccess) return hip_fail("hipSetDevice", e);
        if (m->timing && (e = hipEventRecord(m->ev0[(size_t)g], m->stream[(size_t)g])) != hipSuccess) return hip_fail("hipEventRecord", e);
        if (cnt) {
            if (j.want_direct && mc_context_arm_direct(m->ctx[(size_t)g], &j.slot[g]) == MC_OK)
                j.armed[(size_t)g] = 1;   // not armed (e.g. MC_FINISH=kernel): the whole call falls back to copies
            const int rc = j.launch(g, m->ctx[(size_t)g], j.first + lo, cnt, m->d_send[(size_t)g], (void *)m->stream[(size_t)g]);
            if (rc != MC_OK) {
                if (err) *err = mc_last_error(); else g_multi_error = mc_last_error();
                return rc;
            }
        } else {  // fewer paths than devices: this one contributes {0, 0, 0}
            j.armed[(size_t)g] = 1;
            if ((e = hipMemsetAsync(m->d_send[(size_t)g], 0, 3 * sizeof(double), m->stream[(size_t)g])) != hipSuccess) return hip_fail("hipMemsetAsync", e);
        }
        if (m->timing && (e = hipEventRecord(m->ev1[(size_t)g], m->stream[(size_t)g])) != hipSuccess) return hip_fail("hipEventRecord", e);
        return MC_OK;
    };
    fl.touched = G;   // from here on any device may have work of this call
    int rcs[MAX_DEVICES] = {};
    int64_t at_ns[MAX_DEVICES] = {}, seen_ns[MAX_DEVICES] = {};
    if (m->crew) {
        if (m->crew->run_all(device_part, &job, rcs, wall0, at_ns, seen_ns) > 0) {
            // Bounded wait (mc_multi.h): a launcher thread claimed its device's launch and has not come back.  Nothing can be
            // drained (the same runtime call would be waited for); whatever the other devices still deliver lands in pinned slots
            // the handle keeps alive (it is leaked on destroy), and the handle stays broken.
            fl.touched = 0;
            for (int g = 0; g < G; ++g)
                if (rcs[(size_t)g] == mc_host::LaunchCrew::TIMED_OUT)
                    return fail(MC_ERR_HIP, "device %d: its launcher thread has not returned from the launch it claimed (stuck inside the HIP "
                                            "runtime); the handle is unusable from now on and leaks on mc_multi_destroy", m->devices[(size_t)g]);
        }
    } else {
        for (int g = 0; g < G; ++g) {
            rcs[(size_t)g] = device_part(&job, g);
            at_ns[(size_t)g] = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - wall0).count();
            if (rcs[(size_t)g] != MC_OK)
                break;
        }
    }
    int64_t last_ns = 0;
    for (int g = 0; g < G; ++g) {
        last_ns = at_ns[(size_t)g] > last_ns ? at_ns[(size_t)g] : last_ns;
        if (rcs[(size_t)g] != MC_OK)
            return fail(rcs[(size_t)g], "device %d: %s", m->devices[(size_t)g],
                        m->worker_error.empty() ? g_multi_error.c_str() : m->worker_error[(size_t)g].c_str());
        if (direct && !job.armed[(size_t)g])
            direct = false;
    }
    m->last_fanout_us = last_ns * 1e-3;
    for (int g = 0; g < G; ++g) {
        m->last_seen_us[g] = m->crew ? (seen_ns[(size_t)g] < 0 ? (double)seen_ns[(size_t)g] : seen_ns[(size_t)g] * 1e-3) : -3.0;
        m->last_at_us[g] = at_ns[(size_t)g] * 1e-3;
    }
    if (m->trace) {   // MC_MULTI_TRACE=1: when each launcher thread saw the call and when its launch had been enqueued
        fprintf(stderr, "mc_multi fan-out (us since call entry; %s):", m->crew ? (m->crew->yields() ? "launcher threads, yielding spin" : "launcher threads") : "serial");
        for (int g = 0; g < G; ++g)
            fprintf(stderr, "  [%d] %.2f -> %.2f", g, m->last_seen_us[g], m->last_at_us[g]);
        if (m->crew)
            fprintf(stderr, "  | jobs taken over by the caller so far: %llu", (unsigned long long)m->crew->stolen());
        fprintf(stderr, "\n");
    }
    if (!direct)
        for (int g = 0; g < G; ++g)
            slot[(size_t)g] = nullptr;
    if (m->reduce == MC_REDUCE_RCCL) {
        NCCLCHK(ncclGroupStart());
        fl.group_open = true;
        for (int g = 0; g < G; ++g)
            NCCLCHK(ncclAllReduce(m->d_send[g], m->d_recv[g], 3, ncclDouble, ncclSum, m->comm[g], m->stream[g]));
        fl.group_open = false;
        NCCLCHK(ncclGroupEnd());
        if (direct) {
            const int g = 0;
            MCCHK(mc_context_publish(m->ctx[0], m->d_recv[0], (void *)m->stream[0], &slot[(size_t)G]));
        }
    }
    const volatile double *rslot = slot[(size_t)G];
    m->last_collective_us = -1.0;
    float kernel_ms = 0;
    double host[3] = {0, 0, 0}, reduced[3] = {0, 0, 0};
    if (direct) {
        // one polling loop over every flag word; after 50 ms (BASELINE's C4 and C5 shards take 1-40 ms) hand the core
        // back and wait in the runtime, which is also the way out if a device faulted and will never write
        int64_t ready_ns[MAX_DEVICES + 1];
        if (!mc_host::poll_slots(slot, G + 1, wall0, std::chrono::milliseconds(50), [&] { fl.settle(); }, ready_ns))
            return fail(MC_ERR_HIP, "a device never delivered its result");
        int64_t last_device_ns = -1;
        for (int g = 0; g < G; ++g) {
            m->last_device_us[g] = ready_ns[g] < 0 ? -1.0 : ready_ns[g] * 1e-3;
            last_device_ns = ready_ns[g] > last_device_ns ? ready_ns[g] : last_device_ns;
        }
        // the collective as the host sees it: all-reduce + publish kernel, from the last device's own triple to the reduced one
        m->last_collective_us = (slot[(size_t)G] && ready_ns[G] >= 0 && last_device_ns >= 0) ? (ready_ns[G] - last_device_ns) * 1e-3 : -1.0;
        for (int g = 0; g < G; ++g)
            for (int k = 0; k < 3; ++k)
                host[k] += slot[g] ? slot[g][k] : 0.0;
        for (int k = 0; k < 3 && rslot; ++k)
            reduced[k] = rslot[k];
        fl.touched = 0;   // everything the call enqueued has delivered: nothing left to drain
    } else {
        for (int g = 0; g < G; ++g) {
            HIPCHK(hipSetDevice(m->devices[g]));
            HIPCHK(hipMemcpyAsync(m->h_send + 3 * g, m->d_send[g], 3 * sizeof(double), hipMemcpyDeviceToHost, m->stream[g]));
            if (g == 0 && m->reduce == MC_REDUCE_RCCL)
                HIPCHK(hipMemcpyAsync(m->h_recv, m->d_recv[0], 3 * sizeof(double), hipMemcpyDeviceToHost, m->stream[0]));
        }
        for (int g = 0; g < G; ++g) {
            HIPCHK(hipSetDevice(m->devices[g]));
            HIPCHK(hipStreamSynchronize(m->stream[g]));
            float ms = 0;
            if (m->timing) HIPCHK(hipEventElapsedTime(&ms, m->ev0[g], m->ev1[g]));
            kernel_ms = ms > kernel_ms ? ms : kernel_ms;
        }
        fl.touched = 0;
        // host sum in device order: the collective's cross-check, or the result itself
        for (int g = 0; g < G; ++g)
            for (int k = 0; k < 3; ++k)
                host[k] += m->h_send[3 * g + k];
        for (int k = 0; k < 3; ++k)
            reduced[k] = m->h_recv[k];
    }
    const double *tot = host;
    m->last_reduce_error = 0.0;
    if (m->reduce == MC_REDUCE_RCCL) {
        tot = reduced;
        for (int k = 0; k < 3; ++k) {
            const double err = std::fabs(reduced[k] - host[k]), ref = std::fabs(host[k]);
            if (!(err <= 1e-12 * ref))
                return fail(MC_ERR_HIP, "RCCL all-reduce disagrees with the host sum of the %d device triples: word %d %.17g vs %.17g",
                            G, k, reduced[k], host[k]);
            if (k == 0 && ref > 0)
                m->last_reduce_error = err / ref;
        }
    }
    out->sum = tot[0];
    out->sum2 = tot[1];
    out->n = (uint64_t)tot[2];
    out->kernel_ms = kernel_ms;
    if (out->n != n)
        return fail(MC_ERR_HIP, "devices returned n=%llu in total, expected %llu", (unsigned long long)out->n, (unsigned long long)n);
    mc_closing(out->sum, out->sum2, out->n, discount, &out->expected, &out->confidence);
    out->expected += discount * add_back;
    out->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (m->crew)
        m->crew->call_ended();
    return MC_OK;
}

#define MC_DEFINE_MULTI(X)                                                                                              \
    extern "C" int mc_multi_vanilla_run_##X(mc_multi *m, const mc_option_##X *o, uint64_t seed, uint64_t first, uint64_t n, \
                                            mc_result *out)                                                             \
    {                                                                                                                   \
        if (!o) return fail(MC_ERR_INVALID, "NULL option");                                                             \
        return run_sharded(m, first, n, std::exp(-(double)o->r * (double)o->t), 0.0, out,                               \
                           [=](int, mc_context *c, uint64_t f, uint64_t cnt, double *d, void *st) {                     \
                               return mc_vanilla_launch_##X(c, o, seed, f, cnt, d, st);                                 \
                           });                                                                                          \
    }                                                                                                                   \
    extern "C" int mc_multi_basket_run_##X(mc_multi *m, const mc_basket_##X *o, uint64_t seed, uint64_t first, uint64_t n, \
                                           mc_result *out)                                                              \
    {                                                                                                                   \
        if (!o) return fail(MC_ERR_INVALID, "NULL basket");                                                             \
        double mean = 0.0; /* control variate: the simulated quantity is payoff - control, add its closed-form mean back */ \
        if (m && m->control && mc_basket_control_mean_##X(o, &mean) != MC_OK)                                           \
            return fail(MC_ERR_INVALID, "%s", mc_last_error());                                                         \
        return run_sharded(m, first, n, std::exp(-(double)o->r * (double)o->t), mean, out,                              \
                           [=](int, mc_context *c, uint64_t f, uint64_t cnt, double *d, void *st) {                     \
                               return mc_basket_launch_##X(c, o, seed, f, cnt, d, st);                                  \
                           });                                                                                          \
    }                                                                                                                   \
    extern "C" int mc_multi_cva_run_##X(mc_multi *m, const mc_cva_##X *o, uint64_t seed, uint64_t first, uint64_t n,     \
                                        mc_result *out)                                                                 \
    {                                                                                                                   \
        if (!o) return fail(MC_ERR_INVALID, "NULL cva");                                                                \
        return run_sharded(m, first, n, 1.0, 0.0, out,                                                                  \
                           [=](int, mc_context *c, uint64_t f, uint64_t cnt, double *d, void *st) {                     \
                               return mc_cva_launch_##X(c, o, seed, f, cnt, d, st);                                     \
                           });                                                                                          \
    }

MC_DEFINE_MULTI(f32)
MC_DEFINE_MULTI(f64)
