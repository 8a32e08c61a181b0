// mc_multi_host.hpp -- the two pieces of libmc_multi.so's control flow that touch no GPU API, kept apart so that a plain
// g++ program can drive them on a box without one (tests/cpp/multi_host_check.cpp, tests/test_host_logic.py):
//
//   poll_slots    the read-back loop of run_sharded: G (+ 1) pinned flag words polled from user space, a deadline after
//                 which the caller's `settle` (drain every stream the call touched) runs once and the words are looked
//                 at one last time -- the way out when a device faulted and will never write
//   LaunchCrew    one launcher thread per device: the calling thread hands every device's launch to that device's own
//                 thread through ONE call-number word the whole crew watches, so that all devices start together instead of ~4 us apart
//
// Nothing here exists in the reference (single device, default stream, synchronous: dp/MonteCarloKernel.cu:296-532).
#pragma once
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__) || defined(__i386__)
#define MC_CPU_RELAX() __builtin_ia32_pause()
#else
#define MC_CPU_RELAX() ((void)0)
#endif

namespace mc_host {

constexpr uint64_t SLOT_SENTINEL_BITS = 0xBFF0000000000000ull;   // the bits of -1.0: what an armed slot's n word holds

// word 2 of a slot ({sum, sum2, n}) no longer holds the sentinel: the device's release store has landed
static inline bool slot_ready(const volatile double *slot)
{
    return __atomic_load_n((const uint64_t *)(slot + 2), __ATOMIC_ACQUIRE) != SLOT_SENTINEL_BITS;
}

static inline bool all_ready(const volatile double *const *slots, int n)
{
    for (int i = 0; i < n; ++i)
        if (slots[i] && !slot_ready(slots[i]))
            return false;
    return true;
}

// Polls until every non-NULL slot is ready.  After `spin_for` the core is handed back: settle() -- which waits in the
// runtime for everything the call enqueued -- runs ONCE, then the slots are checked a last time.
// Returns true when all slots delivered, false when some never did (the caller reports MC_ERR_HIP).
// `ready_ns` (optional, n entries): when each slot was first SEEN ready, in ns since t0 (-1: never, or no slot) -- the host's view
// of when every device delivered and when the all-reduced triple did (run_sharded: mc_multi_last_collective_us).
template <class Settle>
static bool poll_slots(const volatile double *const *slots, int n, std::chrono::steady_clock::time_point t0,
                       std::chrono::nanoseconds spin_for, Settle settle, int64_t *ready_ns = nullptr)
{
    if (ready_ns) {
        int pending = 0;
        for (int i = 0; i < n; ++i) {
            ready_ns[i] = -1;
            pending += slots[i] ? 1 : 0;
        }
        const auto stamp = [&] {
            for (int i = 0; i < n; ++i)
                if (slots[i] && ready_ns[i] < 0 && slot_ready(slots[i])) {
                    ready_ns[i] = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                    --pending;
                }
        };
        for (uint32_t spin = 0;; ++spin) {
            stamp();
            if (pending == 0)
                return true;
            if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > spin_for) {
                settle();
                stamp();
                return pending == 0;
            }
            MC_CPU_RELAX();
        }
    }
    for (uint32_t spin = 0;; ++spin) {
        if (all_ready(slots, n))
            return true;
        if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > spin_for) {
            settle();
            return all_ready(slots, n);
        }
        MC_CPU_RELAX();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One launcher thread per device.  A call publishes its job (a callable taking the device index) and bumps the crew's
// ONE call-number word `go_`; worker g runs job(g) on its own thread -- its own hipSetDevice, arm and launch -- stores the
// job's status and bumps `done`.  The caller spins on the `done` words.  Hand-off is one cache line each way, no
// system call: a worker SPINS on that word while calls keep coming (for `linger` after its last job) and parks on a
// condition variable only when the handle has been idle for longer than that, so a sequence of calls never pays a
// wake-up and an idle handle burns no core.  When the process may run on fewer CPUs than the crew has threads (+ the
// caller), every spin iteration yields the CPU instead (sched_yield): a spinning thread that sits on the core the
// thread it waits for needs would otherwise cost a whole scheduler quantum (measured: 2 ms per hand-off with 9 threads
// on 8 CPUs, 3 us with the yield).
// A job belongs to whoever CLAIMS it (one compare-exchange on the worker's `claim` word): normally worker g, but when
// worker g has not claimed its job `steal_after` after the hand-off -- it is parked and still waking up, or the kernel
// scheduler took its core away in the middle of a spin (seen once in 41 calls on a shared 256-thread host: a 9.5 ms
// fan-out) -- the CALLER claims it and runs it on its own thread.  The worst case of a fan-out is then the serial
// fan-out of rounds 2-3, not a scheduler quantum; a handle whose workers sleep is served serially instead of through G
// wake-ups.  Every job still runs exactly once.
// ---------------------------------------------------------------------------------------------------------------
// CPUs' worth of time the cgroup grants (v2 `cpu.max` = "<quota> <period>" or "max <period>"; v1 cfs quota / period), rounded
// up; 0 = no limit or unknown.  `root` is a parameter so that a test can point it at a directory with a faked quota.
static inline int cgroup_cpu_quota(const char *root = "/sys/fs/cgroup")
{
    char path[512];
    long long quota = -1, period = 100000;
    snprintf(path, sizeof path, "%s/cpu.max", root);
    if (FILE *f = fopen(path, "r")) {
        char q[32] = "";
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0)
            quota = atoll(q);
        fclose(f);
    } else {
        snprintf(path, sizeof path, "%s/cpu/cpu.cfs_quota_us", root);
        if (FILE *g = fopen(path, "r")) {
            if (fscanf(g, "%lld", &quota) != 1)
                quota = -1;
            fclose(g);
            snprintf(path, sizeof path, "%s/cpu/cpu.cfs_period_us", root);
            if (FILE *h = fopen(path, "r")) {
                if (fscanf(h, "%lld", &period) != 1)
                    period = 100000;
                fclose(h);
            }
        }
    }
    if (quota <= 0 || period <= 0)
        return 0;
    return (int)((quota + period - 1) / period);
}

static inline int cpus_in_affinity_mask()
{
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) != 0)
        return 1;
    const int n = CPU_COUNT(&set);
    return n > 0 ? n : 1;
}

// CPUs this process can actually keep busy: the affinity mask capped by the cgroup's CPU quota.  The GPU boxes show 256
// hardware threads in the mask and grant 16 CPUs of time (cpu.max "1600000 100000"): round 4 counted the mask alone, so
// nine spinning threads on a 8-CPU grant would have looked fine -- and would have been throttled for the rest of every period.
static inline int cpus_allowed(const char *cgroup_root = "/sys/fs/cgroup")
{
    const int mask = cpus_in_affinity_mask(), quota = cgroup_cpu_quota(cgroup_root);
    return (quota > 0 && quota < mask) ? quota : mask;
}

class LaunchCrew {
public:
    typedef int (*JobFn)(void *ctx, int g);
    // what happened to the jobs of all calls so far (mc_multi_fanout_stats; tools/c/multi_soak.c prints it)
    struct Stats {
        uint64_t calls = 0;
        uint64_t by_worker = 0;        // jobs run by their own launcher thread
        uint64_t served_parked = 0;    // jobs the caller ran at once because their worker was parked (no wake-up on the critical path)
        uint64_t stolen = 0;           // jobs the caller took over `steal_after` after the hand-off: the worker was spinning but late
        uint64_t slow_claimed = 0;     // jobs that took > 1 ms from claim to return on their worker: descheduled INSIDE the job (or a
                                       // runtime call that blocked) -- the one case no take-over can bound
        uint64_t wakeups = 0;          // parked workers woken up after a call because calls had started to come in quick succession
        uint64_t timed_out = 0;        // jobs whose worker claimed them and had not returned `claimed_deadline` later (the crew is then broken)
    };
    static constexpr int TIMED_OUT = -9999;   // rc[g] of a job that was claimed by its worker and never came back (run_all)

    // `cpus` = CPUs the process can keep busy (cpus_allowed(): affinity mask capped by the cgroup quota); a test passes its own.
    LaunchCrew(int n, std::chrono::nanoseconds linger, void (*thread_init)(void *, int) = nullptr, void *init_ctx = nullptr,
               std::chrono::nanoseconds steal_after = std::chrono::microseconds(15), int cpus = cpus_allowed(),
               std::chrono::nanoseconds claimed_deadline = std::chrono::seconds(20))
        : linger_(linger), steal_after_(steal_after), claimed_deadline_(claimed_deadline), yield_(cpus < n + 1), workers_((size_t)n)
    {
        last_end_ = last_start_ = std::chrono::steady_clock::now() - std::chrono::hours(1);
        for (int g = 0; g < n; ++g) {
            workers_[(size_t)g].reset(new Worker);
            Worker *w = workers_[(size_t)g].get();
            w->th = std::thread([this, w, g, thread_init, init_ctx] {
                if (thread_init)
                    thread_init(init_ctx, g);
                run(*w, g);
            });
        }
    }
    ~LaunchCrew()
    {
        quit_.store(true, std::memory_order_seq_cst);
        for (auto &w : workers_) {
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->cv.notify_all();
            }
            w->th.join();   // (a BROKEN crew must not get here: its stuck thread never joins -- retire() below)
        }
    }
    // The end of a crew, broken or not.  A healthy crew is destroyed (its threads joined).  A broken one -- a worker is still
    // inside a job it claimed -- cannot be joined and must not be freed either: that worker will touch its words, the call
    // number and the quit flag whenever it comes back.  It is told to quit, its threads are detached and the object is LEAKED
    // on purpose (a few hundred bytes per worker; the healthy workers leave at once, the stuck one when -- if -- it returns).
    static void retire(std::unique_ptr<LaunchCrew> &crew)
    {
        if (!crew)
            return;
        if (!crew->broken()) {
            crew.reset();
            return;
        }
        LaunchCrew *c = crew.release();
        c->quit_.store(true, std::memory_order_seq_cst);
        for (auto &w : c->workers_) {
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->cv.notify_all();
            }
            w->th.detach();
        }
    }
    LaunchCrew(const LaunchCrew &) = delete;
    LaunchCrew &operator=(const LaunchCrew &) = delete;

    int size() const { return (int)workers_.size(); }
    bool yields() const { return yield_; }
    uint64_t stolen() const { return stats_.stolen + stats_.served_parked; }   // jobs the calling thread ran itself
    const Stats &stats() const { return stats_; }
    // true once a job has timed out: a worker is (or was) stuck inside a job it claimed.  The crew accepts no further calls
    // (run_all returns TIMED_OUT for every job at once); whatever the stuck job's `ctx` points to must outlive it.
    bool broken() const { return broken_; }

    // Runs fn(ctx, g) for every g -- on worker g, or on the calling thread when worker g is parked or late -- and waits for
    // all of them; rc[g] = fn's return value.  The wait is bounded: a job that its worker CLAIMED and that has not returned
    // `claimed_deadline` (default 20 s; the slowest jobs seen in 2.4 million were 7.7 ms, inside the HIP runtime) after the
    // hand-off gets rc[g] = TIMED_OUT, the crew is broken() from then on and the number of such jobs is returned (0 = all
    // came back).  The jobs of the other workers are still waited for (each within the same deadline).
    // enqueued_ns[g] (optional) = steady_clock time at which job g returned, in ns since `t0`.
    // seen_ns[g] (optional) = the time at which worker g SAW the call (hand-off latency apart from the job's own duration);
    //                         -1 = the caller ran it because the worker was late, -2 = because the worker was parked.
    int run_all(JobFn fn, void *ctx, int *rc, std::chrono::steady_clock::time_point t0 = {}, int64_t *enqueued_ns = nullptr,
                int64_t *seen_ns = nullptr)
    {
        using clock = std::chrono::steady_clock;
        if (broken_) {   // a worker may still be inside an earlier call's job: fn_ / ctx_ / t0_ are its to read
            for (size_t g = 0; g < workers_.size(); ++g)
                rc[g] = TIMED_OUT;
            return (int)workers_.size();
        }
        int timed_out = 0;
        fn_ = fn, ctx_ = ctx, t0_ = t0;
        const uint32_t seq = ++seq_;
        ++stats_.calls;
        go_.store(seq, std::memory_order_seq_cst);             // ONE word for the whole crew: every worker sees the call at once
        const auto handed = clock::now();
        // Calls in quick succession (this one began within `linger` of the last one's end, and the last one was itself shorter than
        // `linger`: a spinning worker would have lived to see this call) are worth spinning for: parked workers are woken up --
        // AFTER the fan-out, off its critical path.  A lone call after a long pause leaves them asleep, and so does a sequence of
        // calls that each outlast the linger time (C4 x 10: 34 ms per device) -- they would only spin 5 ms and park again.
        const bool busy = handed - last_end_ <= linger_ && last_duration_ <= linger_;
        last_start_ = handed;
        const auto on_caller = [&](Worker &w, size_t g, int64_t why) {
            w.seen_ns = why;
            w.rc = fn_(ctx_, (int)g);
            w.at_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - t0_).count();
            w.done.store(seq, std::memory_order_release);
        };
        // 1. parked workers first: a futex wake-up costs the caller ~3 us each and the sleeper tens of us to get going, the job
        //    itself ~3-5 us -- the caller claims it and runs it right away (Dekker with the worker's parked = true; read go)
        bool any_parked = false;
        for (size_t g = 0; g < workers_.size(); ++g) {
            Worker &w = *workers_[g];
            if (!w.parked.load(std::memory_order_seq_cst))
                continue;
            any_parked = true;
            uint32_t expect = seq - 1;
            if (w.claim.compare_exchange_strong(expect, seq, std::memory_order_acq_rel)) {
                on_caller(w, g, -2);
                ++stats_.served_parked;
            }   // else: it woke up by itself in between and has the job
        }
        // 2. the spinning ones: wait; one that has not claimed its job `steal_after` after the hand-off lost its core -- take over
        for (size_t g = 0; g < workers_.size(); ++g) {
            Worker &w = *workers_[g];
            bool lost = false;
            for (uint32_t spin = 0; w.done.load(std::memory_order_acquire) != seq; ++spin) {
                if (w.claim.load(std::memory_order_relaxed) != seq) {
                    if (clock::now() - handed > steal_after_) {
                        uint32_t expect = seq - 1;
                        if (w.claim.compare_exchange_strong(expect, seq, std::memory_order_acq_rel)) {
                            on_caller(w, g, -1);
                            ++stats_.stolen;
                            break;
                        }
                    }
                } else if ((spin & 1023u) == 1023u && clock::now() - handed > claimed_deadline_) {
                    lost = true;   // claimed by its worker, never returned: nobody can take a claimed job over
                    break;
                }
                relax();
            }
            if (lost) {
                rc[g] = TIMED_OUT;
                if (enqueued_ns) enqueued_ns[g] = -1;
                if (seen_ns) seen_ns[g] = -4;
                broken_ = true;
                ++timed_out;
                ++stats_.timed_out;
                continue;
            }
            rc[g] = w.rc;
            if (enqueued_ns)
                enqueued_ns[g] = w.at_ns;
            if (seen_ns)
                seen_ns[g] = w.seen_ns;
            if (w.seen_ns >= 0) {
                ++stats_.by_worker;
                if (w.at_ns - w.seen_ns > 1000000)
                    ++stats_.slow_claimed;
            }
        }
        if (any_parked && busy)
            for (auto &w : workers_)
                if (w->parked.load(std::memory_order_seq_cst)) {
                    std::lock_guard<std::mutex> lk(w->mu);
                    w->wake = true;
                    w->cv.notify_one();
                    ++stats_.wakeups;
                }
        return timed_out;
    }
    // The caller's side of "a call has ended" (run_sharded: the estimate is closed): the next call's distance from here decides
    // whether parked workers are worth waking.
    void call_ended()
    {
        last_end_ = std::chrono::steady_clock::now();
        last_duration_ = last_end_ - last_start_;
    }

private:
    struct alignas(128) Worker {
        alignas(128) std::atomic<uint32_t> claim{0};   // the last call whose job somebody took: worker g, or the caller when g was parked or late
        alignas(128) std::atomic<uint32_t> done{0};
        int rc = 0;
        int64_t at_ns = 0, seen_ns = 0;
        alignas(128) std::atomic<bool> parked{false};
        bool wake = false;                             // under mu: "get up and spin" (set by the caller after a call in a busy phase)
        std::mutex mu;
        std::condition_variable cv;
        std::thread th;
    };

    void run(Worker &w, int g)
    {
        uint32_t seen = 0;
        auto idle_since = std::chrono::steady_clock::now();
        for (;;) {
            uint32_t cur;
            uint32_t spin = 0;
            while ((cur = go_.load(std::memory_order_acquire)) == seen) {
                if (quit_.load(std::memory_order_relaxed))
                    return;
                if ((++spin & 1023u) == 0 && std::chrono::steady_clock::now() - idle_since > linger_) {
                    // Park.  A sleeper is not woken by the next call (the caller serves it), only by `wake` -- so it keeps
                    // sleeping through lone calls and `seen` catches up with go_ when it gets up.
                    std::unique_lock<std::mutex> lk(w.mu);
                    w.wake = false;
                    w.parked.store(true, std::memory_order_seq_cst);
                    if (go_.load(std::memory_order_seq_cst) == seen)          // Dekker: a call handed off before `parked` was visible is ours
                        while (!w.wake && !quit_.load(std::memory_order_seq_cst))
                            w.cv.wait(lk);
                    w.parked.store(false, std::memory_order_seq_cst);
                    idle_since = std::chrono::steady_clock::now();
                } else {
                    relax();
                }
            }
            seen = cur;
            uint32_t expect = cur - 1;
            if (w.claim.compare_exchange_strong(expect, cur, std::memory_order_acq_rel)) {
                w.seen_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
                w.rc = fn_(ctx_, g);
                w.at_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
                w.done.store(cur, std::memory_order_release);
            }   // else: the caller ran this one (or a later one) itself; its words are the caller's to write
            idle_since = std::chrono::steady_clock::now();
        }
    }

    void relax() const
    {
        if (yield_)
            sched_yield();
        else
            MC_CPU_RELAX();
    }

    // the call number, bumped once per run_all: all workers spin on this one (read-shared) line, so the last of them sees a
    // call as early as the first -- with one word per worker the caller's G stores put worker 7 about 1.2 us behind worker 0
    alignas(128) std::atomic<uint32_t> go_{0};
    alignas(128) std::chrono::nanoseconds linger_;
    std::chrono::nanoseconds steal_after_, claimed_deadline_;
    bool broken_ = false;
    bool yield_;
    Stats stats_;
    std::chrono::steady_clock::time_point last_end_, last_start_;
    std::chrono::nanoseconds last_duration_{0};
    std::vector<std::unique_ptr<Worker>> workers_;
    std::atomic<bool> quit_{false};
    JobFn fn_ = nullptr;
    void *ctx_ = nullptr;
    std::chrono::steady_clock::time_point t0_{};
    uint32_t seq_ = 0;
};

}  // namespace mc_host
