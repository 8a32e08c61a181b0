// multi_host_check.cpp -- the GPU-free control flow of libmc_multi.so (montecarlocuda_amd/csrc/mc_multi_host.hpp) on a box
// without a GPU: the slot-polling loop of run_sharded with FAKE slots in ordinary memory, and the launcher-thread crew.
//   g++ -O2 -std=c++17 -pthread -I montecarlocuda_amd/csrc tests/cpp/multi_host_check.cpp -o multi_host_check
// Prints one line per check, "all checks passed" and exit status 0 iff everything held (tests/test_host_logic.py).
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <unistd.h>

#include "mc_multi_host.hpp"

using namespace mc_host;
using clk = std::chrono::steady_clock;

static int failures;
#define EXPECT(cond, what)                                   \
    do {                                                     \
        const bool ok_ = (cond);                             \
        printf("%-86s %s\n", what, ok_ ? "ok" : "FAILED");   \
        failures += !ok_;                                    \
    } while (0)

struct Slot { volatile double w[4]; };
static void arm(Slot &s) { s.w[0] = s.w[1] = 0, s.w[2] = -1.0; }
// what a device's last workgroup does: {sum, sum2}, then n with release semantics
static void deliver(Slot &s, double a, double b, double n)
{
    s.w[0] = a, s.w[1] = b;
    uint64_t bits;
    memcpy(&bits, &n, 8);
    __atomic_store_n((uint64_t *)&s.w[2], bits, __ATOMIC_RELEASE);
}

int main()
{
    // ---- poll_slots ------------------------------------------------------------------------------------------------
    {
        Slot s[4];
        for (auto &x : s) arm(x);
        const volatile double *slots[4] = {s[0].w, s[1].w, nullptr, s[3].w};   // a NULL entry = a device with no slot (skipped)
        EXPECT(!slot_ready(s[0].w), "an armed slot reads as not ready");
        std::thread dev([&] {
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            deliver(s[0], 1, 2, 10);
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            deliver(s[3], 3, 4, 30);
            deliver(s[1], 5, 6, 20);
        });
        int settled = 0;
        const auto t0 = clk::now();
        const bool ok = poll_slots(slots, 4, t0, std::chrono::seconds(5), [&] { ++settled; });
        dev.join();
        EXPECT(ok && settled == 0, "G = 3 (+ one NULL entry): every slot delivered -> true, settle never called");
        EXPECT(s[0].w[2] == 10 && s[1].w[2] == 20 && s[3].w[2] == 30 && s[1].w[0] == 5, "the triples are visible after the poll");
        EXPECT(clk::now() - t0 < std::chrono::seconds(2), "... and without waiting for the deadline");
    }
    {
        // the case ADVICE r03 asks for: G = 3, one slot is never written -> settle() runs exactly once (the streams are
        // drained) and the call reports failure
        Slot s[3];
        for (auto &x : s) arm(x);
        const volatile double *slots[3] = {s[0].w, s[1].w, s[2].w};
        deliver(s[0], 1, 1, 1);
        deliver(s[2], 1, 1, 1);
        int settled = 0;
        const auto t0 = clk::now();
        const bool ok = poll_slots(slots, 3, t0, std::chrono::milliseconds(30), [&] { ++settled; });
        const auto took = clk::now() - t0;
        EXPECT(!ok && settled == 1, "one slot never written: false, settle called exactly once");
        EXPECT(took >= std::chrono::milliseconds(30) && took < std::chrono::seconds(2), "... after the spin deadline, not before and not much later");
    }
    {
        // a slot that is written only while settle() waits in the runtime (a slow device): delivered after all
        Slot s[2];
        for (auto &x : s) arm(x);
        const volatile double *slots[2] = {s[0].w, s[1].w};
        deliver(s[0], 1, 1, 1);
        int settled = 0;
        const bool ok = poll_slots(slots, 2, clk::now(), std::chrono::milliseconds(10), [&] { ++settled; deliver(s[1], 7, 8, 9); });
        EXPECT(ok && settled == 1 && s[1].w[2] == 9, "a slot written during settle() still counts as delivered");
    }
    // ---- LaunchCrew ------------------------------------------------------------------------------------------------
    {
        struct Ctx { std::atomic<int> calls{0}; std::thread::id ids[8]; int inited[8]; } ctx;
        memset(ctx.inited, 0, sizeof ctx.inited);
        // steal_after = 10 s: the caller never takes a job over in this block, so "each on its own thread" can be asserted
        // linger 2 s: no worker goes to sleep before the first call, however slowly the eight threads start on a loaded box
        LaunchCrew crew(8, std::chrono::seconds(2), [](void *c, int g) { static_cast<Ctx *>(c)->inited[g] = g + 1; }, &ctx,
                        std::chrono::seconds(10));
        int rc[8];
        int64_t at[8];
        const auto job = [](void *c, int g) -> int {
            Ctx &x = *static_cast<Ctx *>(c);
            x.ids[g] = std::this_thread::get_id();
            x.calls.fetch_add(1);
            return 100 + g;
        };
        crew.run_all(job, &ctx, rc, clk::now(), at);
        bool each = ctx.calls.load() == 8, distinct = true, inited = true;
        for (int g = 0; g < 8; ++g) {
            each = each && rc[g] == 100 + g && at[g] >= 0;
            inited = inited && ctx.inited[g] == g + 1;
            for (int h = 0; h < g; ++h)
                distinct = distinct && ctx.ids[g] != ctx.ids[h];
            distinct = distinct && ctx.ids[g] != std::this_thread::get_id();
        }
        EXPECT(each, "8 workers: every job ran once and its status came back in its own slot");
        EXPECT(distinct && inited, "... each on its own thread (not the caller's), after that thread's init hook");
        // many calls back to back (workers spinning), then after the linger time (workers parked): nothing lost either way
        for (int i = 0; i < 20000; ++i)
            crew.run_all(job, &ctx, rc);
        EXPECT(ctx.calls.load() == 8 * 20001, "20 000 back-to-back calls: no hand-off lost while the workers spin");
        {
            LaunchCrew naps(8, std::chrono::milliseconds(2), nullptr, nullptr, std::chrono::seconds(10));
            for (int i = 0; i < 30; ++i) {
                std::this_thread::sleep_for(std::chrono::milliseconds(i % 3 == 0 ? 6 : 1));   // around the 2 ms linger time
                naps.run_all(job, &ctx, rc);
                naps.call_ended();
            }
            printf("   %llu of 240 jobs served by the caller for sleeping workers, %llu wake-ups\n", (unsigned long long)naps.stats().served_parked,
                   (unsigned long long)naps.stats().wakeups);
        }
        EXPECT(ctx.calls.load() == 8 * 20031, "30 calls with pauses around the linger time: sleeping workers are served or woken, none is lost");
        // hand-off latency while the workers spin: call entry -> the last job returned
        double worst = 0, sum = 0;
        for (int i = 0; i < 2000; ++i) {
            crew.run_all(job, &ctx, rc, clk::now(), at);
            int64_t last = 0;
            for (int g = 0; g < 8; ++g) last = at[g] > last ? at[g] : last;
            sum += last * 1e-3, worst = last * 1e-3 > worst ? last * 1e-3 : worst;
        }
        printf("   hand-off to 8 spinning workers (%d CPUs allowed, %s), entry -> last job returned: mean %.2f us, worst %.1f us (2000 calls)\n",
               cpus_allowed(), crew.yields() ? "yielding spin" : "pause spin", sum / 2000, worst);
        EXPECT(sum / 2000 < 200.0, "mean hand-off latency below 200 us (loose: shared CI cores)");
    }   // the destructor joins 8 workers, some parked, some spinning
    EXPECT(true, "crew destroyed (spinning and parked workers joined)");
    {
        LaunchCrew crew(3, std::chrono::nanoseconds(0));   // linger 0: workers park at once
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        int rc[3] = {-1, -1, -1};
        crew.run_all([](void *, int g) { return g; }, nullptr, rc);
        EXPECT(rc[0] == 0 && rc[1] == 1 && rc[2] == 2, "linger 0: a call on parked workers completes");
    }
    // ---- CPUs the process can keep busy: affinity mask capped by the cgroup quota (round 4 counted the mask alone) ------------
    {
        char tmpl[] = "/tmp/mc_cgroup_XXXXXX";
        const std::string root = mkdtemp(tmpl);
        const auto put = [&](const std::string &rel, const char *text) {
            FILE *f = fopen((root + "/" + rel).c_str(), "w");
            fputs(text, f);
            fclose(f);
        };
        EXPECT(cgroup_cpu_quota(root.c_str()) == 0, "no cpu.max and no cfs files: no quota");
        put("cpu.max", "max 100000\n");
        EXPECT(cgroup_cpu_quota(root.c_str()) == 0 && cpus_allowed(root.c_str()) == cpus_in_affinity_mask(), "cgroup v2 'max 100000': no quota, the mask counts");
        put("cpu.max", "1600000 100000\n");
        EXPECT(cgroup_cpu_quota(root.c_str()) == 16, "cgroup v2 '1600000 100000' (the GPU boxes): 16 CPUs");
        put("cpu.max", "150000 100000\n");
        EXPECT(cgroup_cpu_quota(root.c_str()) == 2, "a fractional grant rounds up: 1.5 -> 2");
        put("cpu.max", "100000 100000\n");
        EXPECT(cpus_allowed(root.c_str()) == 1, "quota of one CPU under a wider mask: cpus_allowed() = 1");
        unlink((root + "/cpu.max").c_str());
        mkdir((root + "/cpu").c_str(), 0700);
        put("cpu/cpu.cfs_quota_us", "400000\n");
        put("cpu/cpu.cfs_period_us", "50000\n");
        EXPECT(cgroup_cpu_quota(root.c_str()) == 8, "cgroup v1 cfs quota 400000 / period 50000: 8 CPUs");
        put("cpu/cpu.cfs_quota_us", "-1\n");
        EXPECT(cgroup_cpu_quota(root.c_str()) == 0, "cgroup v1 quota -1: no limit");
        unlink((root + "/cpu/cpu.cfs_quota_us").c_str()), unlink((root + "/cpu/cpu.cfs_period_us").c_str());
        rmdir((root + "/cpu").c_str()), rmdir(root.c_str());
        // what the crew makes of it: 8 workers + the caller need 9 CPUs to spin with `pause`; with fewer every spin yields
        LaunchCrew scarce(8, std::chrono::nanoseconds(0), nullptr, nullptr, std::chrono::microseconds(15), 8);
        LaunchCrew plenty(8, std::chrono::nanoseconds(0), nullptr, nullptr, std::chrono::microseconds(15), 16);
        EXPECT(scarce.yields() && !plenty.yields(), "8 launcher threads: yielding spin on a grant of 8 CPUs, pause spin on 16");
    }
    // ---- sleeping workers: the caller serves them at once; they are woken only when calls come in quick succession ---------
    {
        struct Ctx { std::atomic<int> calls{0}; } ctx;
        const auto job = [](void *c, int g) -> int { static_cast<Ctx *>(c)->calls.fetch_add(1); return g; };
        int rc[4];
        int64_t at[4], seen[4];
        LaunchCrew crew(4, std::chrono::milliseconds(100), nullptr, nullptr, std::chrono::seconds(10));   // steal_after 10 s: no late take-over here
        crew.run_all(job, &ctx, rc, clk::now(), at, seen);
        crew.call_ended();
        EXPECT(crew.stats().by_worker == 4 && crew.stats().served_parked == 0, "fresh crew (spinning): the four jobs ran on their workers");
        std::this_thread::sleep_for(std::chrono::milliseconds(250));                                       // > linger: all four park
        const auto t0 = clk::now();
        crew.run_all(job, &ctx, rc, t0, at, seen);
        crew.call_ended();
        bool all_parked = true;
        for (int g = 0; g < 4; ++g) all_parked = all_parked && seen[g] == -2 && rc[g] == g;
        EXPECT(all_parked && crew.stats().served_parked == 4, "a lone call after a long pause: all four jobs served by the caller (seen = -2)");
        EXPECT(crew.stats().wakeups == 0, "... and nobody was woken up for it");
        std::this_thread::sleep_for(std::chrono::milliseconds(5));                                         // < linger since the last call ended
        crew.run_all(job, &ctx, rc, clk::now(), at, seen);
        crew.call_ended();
        EXPECT(crew.stats().served_parked == 8 && crew.stats().wakeups == 4, "a second call soon after: served by the caller again, then the four sleepers are woken");
        std::this_thread::sleep_for(std::chrono::milliseconds(20));                                        // time to get up; they spin now
        crew.run_all(job, &ctx, rc, clk::now(), at, seen);
        crew.call_ended();
        bool by_workers = true;
        for (int g = 0; g < 4; ++g) by_workers = by_workers && seen[g] >= 0;
        EXPECT(by_workers && crew.stats().by_worker == 8, "the third call finds them spinning: the jobs run on the workers again");
        EXPECT(ctx.calls.load() == 16 && crew.stats().calls == 4 && crew.stats().stolen == 0, "16 jobs, each exactly once; none counted as a late take-over");
        // calls that each outlast the linger time, back to back: the workers sleep through all of them (waking them would buy 100 ms of
        // spinning and another sleep per call), the caller serves every job
        std::this_thread::sleep_for(std::chrono::milliseconds(250));
        const uint64_t wake0 = crew.stats().wakeups, served0 = crew.stats().served_parked;
        for (int i = 0; i < 3; ++i) {
            crew.run_all(job, &ctx, rc, clk::now(), at, seen);
            std::this_thread::sleep_for(std::chrono::milliseconds(150));      // the "kernel": longer than the linger time
            crew.call_ended();
        }
        EXPECT(crew.stats().wakeups == wake0 && crew.stats().served_parked == served0 + 12, "three back-to-back calls of 150 ms each: nobody is woken, 12 jobs served by the caller");
    }
    // ---- late workers: the caller claims their jobs (every job still runs exactly once) -------------------------------------
    {
        struct Ctx { std::atomic<int> calls{0}; std::atomic<int> per[4]; std::atomic<int> on_caller{0}; std::thread::id caller; } ctx;
        for (auto &x : ctx.per) x.store(0);
        ctx.caller = std::this_thread::get_id();
        const auto job = [](void *c, int g) -> int {
            Ctx &x = *static_cast<Ctx *>(c);
            x.calls.fetch_add(1), x.per[g].fetch_add(1);
            if (std::this_thread::get_id() == x.caller) x.on_caller.fetch_add(1);
            return 7 * g;
        };
        int rc[4];
        {
            // parked workers (linger 0) and steal_after 0: a worker needs a futex wake-up, the caller needs one compare-exchange
            LaunchCrew crew(4, std::chrono::nanoseconds(0), nullptr, nullptr, std::chrono::nanoseconds(0));
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            bool statuses = true;
            for (int i = 0; i < 200; ++i) {
                crew.run_all(job, &ctx, rc);
                for (int g = 0; g < 4; ++g) statuses = statuses && rc[g] == 7 * g;
                if (i % 20 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
            bool once = ctx.calls.load() == 4 * 200;
            for (auto &x : ctx.per) once = once && x.load() == 200;
            EXPECT(once && statuses, "parked workers, steal_after 0: 200 calls x 4 jobs, every job ran exactly once, statuses in their slots");
            EXPECT(crew.stolen() > 0 && (int)crew.stolen() == ctx.on_caller.load(), "... some of them on the calling thread, and stolen() counts exactly those");
            printf("   %d of 800 jobs ran on the caller\n", ctx.on_caller.load());
        }
        {
            // spinning workers and a steal_after of about one hand-off: worker and caller race for every job
            for (auto &x : ctx.per) x.store(0);
            ctx.calls.store(0), ctx.on_caller.store(0);
            LaunchCrew crew(4, std::chrono::milliseconds(50), nullptr, nullptr, std::chrono::nanoseconds(400));
            for (int i = 0; i < 20000; ++i)
                crew.run_all(job, &ctx, rc);
            bool once = ctx.calls.load() == 4 * 20000;
            for (auto &x : ctx.per) once = once && x.load() == 20000;
            EXPECT(once, "spinning workers racing the caller for 20 000 x 4 jobs: every job ran exactly once");
            printf("   %d of 80000 jobs ran on the caller\n", ctx.on_caller.load());
        }
    }
    // ---- the bounded wait for a CLAIMED job (ADVICE r04 / VERDICT r05 #5) --------------------------------------------------
    {
        // worker 2 claims its job and blocks inside it (a launch call stuck in the runtime): run_all gives up on THAT device after
        // the deadline, still collects the others, and the crew is broken from then on; retiring it frees nothing the stuck thread
        // can touch, so the job may finish whenever it likes
        struct Ctx {
            std::atomic<int> gate{0}, ran{0};
        } ctx;
        const auto job = [](void *c, int g) -> int {
            Ctx &x = *static_cast<Ctx *>(c);
            if (g == 2)
                while (!x.gate.load(std::memory_order_acquire))
                    std::this_thread::sleep_for(std::chrono::milliseconds(1));
            x.ran.fetch_add(1);
            return 10 + g;
        };
        std::unique_ptr<LaunchCrew> crew(new LaunchCrew(4, std::chrono::milliseconds(100), nullptr, nullptr, std::chrono::seconds(10), 64,
                                                        std::chrono::milliseconds(150)));
        int rc[4] = {0, 0, 0, 0};
        int64_t at[4], seen[4];
        const auto t0 = clk::now();
        const int lost = crew->run_all(job, &ctx, rc, t0, at, seen);
        const auto took = clk::now() - t0;
        EXPECT(lost == 1 && rc[2] == LaunchCrew::TIMED_OUT && rc[0] == 10 && rc[1] == 11 && rc[3] == 13,
               "a job blocked inside its worker: run_all returns 1, TIMED_OUT in that slot, the other statuses in theirs");
        EXPECT(took >= std::chrono::milliseconds(150) && took < std::chrono::seconds(5), "... after the deadline, not for ever");
        EXPECT(seen[2] == -4 && at[2] == -1, "... and the trace marks the device (seen = -4)");
        EXPECT(crew->broken() && crew->stats().timed_out == 1, "the crew is broken and counts the job");
        int rc2[4] = {0, 0, 0, 0};
        const auto t1 = clk::now();
        const int refused = crew->run_all(job, &ctx, rc2);
        EXPECT(refused == 4 && rc2[0] == LaunchCrew::TIMED_OUT && rc2[3] == LaunchCrew::TIMED_OUT && clk::now() - t1 < std::chrono::milliseconds(100) &&
                   ctx.ran.load() == 3,
               "a broken crew refuses the next call at once and runs nothing");
        LaunchCrew::retire(crew);
        EXPECT(!crew, "retire() takes the broken crew away without joining (it is leaked on purpose)");
        ctx.gate.store(1, std::memory_order_release);
        for (int i = 0; i < 2000 && ctx.ran.load() != 4; ++i)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        EXPECT(ctx.ran.load() == 4, "the stuck job finishes on its own thread afterwards: nothing it touches was freed");
        std::this_thread::sleep_for(std::chrono::milliseconds(20));   // let that thread see the quit flag and leave before ctx goes
        std::unique_ptr<LaunchCrew> healthy(new LaunchCrew(2, std::chrono::milliseconds(1)));
        LaunchCrew::retire(healthy);
        EXPECT(!healthy, "retire() of a healthy crew is its destructor (threads joined)");
    }
    if (failures) {
        printf("%d check(s) FAILED\n", failures);
        return 1;
    }
    printf("all checks passed\n");
    return 0;
}
