// mc_multi.cpp -- libmc_multi.so: one pricing call over several MI355X from one host process (include/mc_multi.h).
//
// A client of the single-device C ABI (include/mc_mi355x.h: one mc_context per device, mc_*_launch_* on that
// device's stream) plus RCCL's C API, linked directly (librccl.so; no PyTorch anywhere near this file).  The
// reference has no counterpart: it is single-device, default stream, synchronous; the entry points
// fanned out here are dp/MonteCarloKernel.cu:483 (basket), :500 (vanilla), :517 (CVA).
//
// Host-only translation unit: no device code.  Everything a call enqueues -- G launches, one grouped all-reduce, one
// one-lane publish kernel -- is asynchronous; the results come back through pinned host memory that the devices write
// themselves and the calling thread polls (run_sharded).  NOTE: the grouped all-reduce has only ever run with a
// communicator of ONE rank (one-GPU test boxes; RCCL refuses a repeated device): the G > 1 collective is unexercised.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mc_multi.h"

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
};

extern "C" void mc_multi_destroy(mc_multi *m)
{
    if (!m)
        return;
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
    const int visible = mc_device_count();
    if (visible <= 0)
        return fail(MC_ERR_NO_DEVICE, "no HIP device visible (the HIP engine has no CPU fallback)");
    if (!devices && n_devices <= 0)
        n_devices = visible;
    if (n_devices <= 0 || n_devices > 64)
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
    *out = m;
    return MC_OK;
}

extern "C" int mc_multi_size(const mc_multi *m) { return m ? (int)m->devices.size() : 0; }
extern "C" mc_context *mc_multi_context(mc_multi *m, int i)
{
    return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr;
}
extern "C" double mc_multi_last_reduce_error(const mc_multi *m) { return m ? m->last_reduce_error : 0.0; }

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
    for (int a = 0; a < G; ++a)
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

static inline bool slot_ready(const volatile double *slot)
{
    return __atomic_load_n((const uint64_t *)(slot + 2), __ATOMIC_ACQUIRE) != 0xBFF0000000000000ull;   // bits of -1.0
}

// launch(g, ctx, first, count, d_triple, stream) enqueues device g's shard.
//
// Read-back.  With timing off (what the legacy symbols and the benchmarks use) nothing is copied and nothing sleeps:
// every device's last workgroup stores its triple into a pinned host slot of its context (mc_context_arm_direct), the
// all-reduced triple follows through a one-lane kernel behind the collective on device 0's stream
// (mc_context_publish), and this thread polls the G (+ 1) flag words in ONE loop from user space.  Round 2 issued G + 1
// hipMemcpyAsync and G serial hipStreamSynchronize calls instead.  Measured on one device, C5's shard of 8 (a 1.04 ms
// kernel), wall minus the kernel's own duration per call (profiles/r03_multi_fixed_cost.log, tools/c/multi_cost.c; the
// dispatch-bound events of the measurement itself cost ~10 us of it): host sum 13.8 us = the single-device floor (13.7),
// RCCL over one rank + publish 21.8 us; round 2's form 23.7 / 28.6 us.  With timing on the event/copy/synchronize path
// is kept: it is the one that can report kernel_ms.
template <class Launch>
static int run_sharded(mc_multi *m, uint64_t first, uint64_t n, double discount, double add_back, mc_result *out, Launch launch)
{
    if (!m) return fail(MC_ERR_INVALID, "NULL handle");
    if (!out) return fail(MC_ERR_INVALID, "NULL output pointer");
    if (n == 0) return fail(MC_ERR_INVALID, "n_paths == 0");
    const int G = (int)m->devices.size();
    if (m->reduce == MC_REDUCE_RCCL)
        if (int rc = ensure_comms(m)) return rc;
    const auto wall0 = std::chrono::steady_clock::now();
    InFlight fl(m);
    bool direct = !m->timing && !m->readback_copy;
    std::vector<const volatile double *> slot((size_t)G, nullptr);
    const volatile double *rslot = nullptr;
    for (int g = 0; g < G; ++g) {
        uint64_t lo = 0, cnt = 0;
        mc_shard_range(n, g, G, &lo, &cnt);
        HIPCHK(hipSetDevice(m->devices[g]));
        fl.touched = g + 1;
        if (m->timing) HIPCHK(hipEventRecord(m->ev0[g], m->stream[g]));
        if (cnt) {
            if (direct && mc_context_arm_direct(m->ctx[g], &slot[g]) != MC_OK)
                direct = false;   // e.g. MC_FINISH=kernel: fall back to copies for the whole call (g == 0: nothing armed yet)
            MCCHK(launch(g, m->ctx[g], first + lo, cnt, m->d_send[g], (void *)m->stream[g]));
        } else {  // fewer paths than devices: this one contributes {0, 0, 0}
            HIPCHK(hipMemsetAsync(m->d_send[g], 0, 3 * sizeof(double), m->stream[g]));
        }
        if (m->timing) HIPCHK(hipEventRecord(m->ev1[g], m->stream[g]));
    }
    if (m->reduce == MC_REDUCE_RCCL) {
        NCCLCHK(ncclGroupStart());
        fl.group_open = true;
        for (int g = 0; g < G; ++g)
            NCCLCHK(ncclAllReduce(m->d_send[g], m->d_recv[g], 3, ncclDouble, ncclSum, m->comm[g], m->stream[g]));
        fl.group_open = false;
        NCCLCHK(ncclGroupEnd());
        if (direct) {
            const int g = 0;
            MCCHK(mc_context_publish(m->ctx[0], m->d_recv[0], (void *)m->stream[0], &rslot));
        }
    }
    float kernel_ms = 0;
    double host[3] = {0, 0, 0}, reduced[3] = {0, 0, 0};
    if (direct) {
        // one polling loop over every flag word; after 50 ms (BASELINE's C4 and C5 shards take 1-40 ms) hand the core
        // back and wait in the runtime, which is also the way out if a device faulted and will never write
        bool all = false;
        for (uint32_t spin = 0; !all; ++spin) {
            all = !rslot || slot_ready(rslot);
            for (int g = 0; g < G && all; ++g)
                all = !slot[g] || slot_ready(slot[g]);
            if (all)
                break;
            if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - wall0 > std::chrono::milliseconds(50)) {
                fl.settle();
                all = !rslot || slot_ready(rslot);
                for (int g = 0; g < G && all; ++g)
                    all = !slot[g] || slot_ready(slot[g]);
                if (!all)
                    return fail(MC_ERR_HIP, "a device never delivered its result");
                break;
            }
            __builtin_ia32_pause();
        }
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
    return MC_OK;
}

#define MC_DEFINE_MULTI(X)                                                                                              \
    extern "C" int mc_multi_vanilla_run_##X(mc_multi *m, const mc_option_##X *o, uint64_t seed, uint64_t first, uint64_t n, \
                                            mc_result *out)                                                             \
    {                                                                                                                   \
        if (!o) return fail(MC_ERR_INVALID, "NULL option");                                                             \
        return run_sharded(m, first, n, std::exp(-(double)o->r * (double)o->t), 0.0, out,                               \
                           [&](int, mc_context *c, uint64_t f, uint64_t cnt, double *d, void *st) {                     \
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
                           [&](int, mc_context *c, uint64_t f, uint64_t cnt, double *d, void *st) {                     \
                               return mc_basket_launch_##X(c, o, seed, f, cnt, d, st);                                  \
                           });                                                                                          \
    }                                                                                                                   \
    extern "C" int mc_multi_cva_run_##X(mc_multi *m, const mc_cva_##X *o, uint64_t seed, uint64_t first, uint64_t n,     \
                                        mc_result *out)                                                                 \
    {                                                                                                                   \
        if (!o) return fail(MC_ERR_INVALID, "NULL cva");                                                                \
        return run_sharded(m, first, n, 1.0, 0.0, out,                                                                  \
                           [&](int, mc_context *c, uint64_t f, uint64_t cnt, double *d, void *st) {                     \
                               return mc_cva_launch_##X(c, o, seed, f, cnt, d, st);                                     \
                           });                                                                                          \
    }

MC_DEFINE_MULTI(f32)
MC_DEFINE_MULTI(f64)
