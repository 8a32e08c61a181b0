// rccl_mock.hip -- TEST DOUBLE for the six RCCL entry points libmc_multi.so calls (ncclCommInitAll, ncclCommDestroy,
// ncclGroupStart, ncclGroupEnd, ncclAllReduce, ncclGetErrorString), for one-GPU boxes.
//
// Why: RCCL refuses a communicator that lists a device twice, so on a box with ONE GPU the grouped all-reduce of
// csrc/mc_multi.cpp (run_sharded: G calls of ncclAllReduce between ncclGroupStart / ncclGroupEnd, one per device, each on
// that device's stream with that device's send / receive buffers and communicator; the publish of device 0's reduced triple
// behind it) has only ever run with G = 1.  Preloaded (LD_PRELOAD) in front of librccl.so, this file gives the SAME call
// sequence the semantics of an all-reduce among ranks that all live on device 0, so that the control flow and the indexing
// of a G > 1 call can be executed: tests/test_gpu_multi.py runs tests/c/multi_check.c over {0, 0, 0} this way.
// It is NOT RCCL and proves nothing about RCCL, xGMI or a second GPU -- only that libmc_multi hands every rank's buffers,
// stream and communicator to the collective correctly and reads the result back from the right place.
//
// Semantics implemented: inside a group the calls are recorded; ncclGroupEnd records an event on every call's stream (behind
// what that stream already holds: the rank's simulation kernel), makes every stream wait for all of them, and enqueues on
// each stream a one-workgroup kernel that adds the ranks' send buffers IN RANK ORDER into that rank's receive buffer -- the
// sum every rank of a real all-reduce would end with (RCCL's ring adds in another order: the library's own cross-check
// against the host sum allows 1e-12 relative for that reason).  Checked and refused with ncclInvalidArgument: a count or
// datatype other than the library's (3, ncclDouble, ncclSum), a communicator used twice in a group, a group that does not
// hold exactly one call per rank of the communicator set, a NULL buffer or communicator.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tests/cpp/rccl_mock.hip -o librccl_mock.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

struct MockSet {            // the communicators of one ncclCommInitAll
    int nranks = 0;
    int alive = 0;
};
struct MockComm {
    unsigned magic = 0x4d4f434b;   // "MOCK"
    MockSet *set = nullptr;
    int rank = -1;
    int device = -1;
};
struct Op {
    const double *send;
    double *recv;
    MockComm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
long g_allreduces = 0, g_groups = 0;

constexpr int MAX_RANKS = 64;
struct Sources { const double *p[MAX_RANKS]; int n; };

__global__ void mock_allreduce_sum3(Sources src, double *dst)
{
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int r = 0; r < src.n; ++r)
            s += src.p[r][threadIdx.x];
        dst[threadIdx.x] = s;
    }
}

MockComm *as_mock(ncclComm_t c)
{
    MockComm *m = reinterpret_cast<MockComm *>(c);
    return (m && m->magic == 0x4d4f434b) ? m : nullptr;
}

ncclResult_t run_group(std::vector<Op> &ops)
{
    if (ops.empty())
        return ncclSuccess;
    MockSet *set = ops[0].comm->set;
    const int n = (int)ops.size();
    if (n != set->nranks || n > MAX_RANKS) {
        fprintf(stderr, "rccl_mock: a group of %d all-reduce calls for a communicator set of %d ranks\n", n, set->nranks);
        return ncclInvalidArgument;
    }
    Sources src;
    src.n = n;
    std::vector<char> seen((size_t)n, 0);
    for (const Op &o : ops) {
        if (o.comm->set != set || seen[(size_t)o.comm->rank]) {
            fprintf(stderr, "rccl_mock: rank %d appears twice in a group, or communicators of two sets are mixed\n", o.comm->rank);
            return ncclInvalidArgument;
        }
        seen[(size_t)o.comm->rank] = 1;
        src.p[o.comm->rank] = o.send;      // rank order, whatever order the calls were made in
    }
    std::vector<hipEvent_t> ready((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(ops[(size_t)i].comm->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&ready[(size_t)i], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(ready[(size_t)i], ops[(size_t)i].stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (int i = 0; i < n; ++i) {
        const Op &o = ops[(size_t)i];
        if (hipSetDevice(o.comm->device) != hipSuccess) return ncclUnhandledCudaError;
        for (int j = 0; j < n; ++j)
            if (hipStreamWaitEvent(o.stream, ready[(size_t)j], 0) != hipSuccess) return ncclUnhandledCudaError;
        mock_allreduce_sum3<<<1, 64, 0, o.stream>>>(src, o.recv);
        if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    }
    for (hipEvent_t e : ready)
        (void)hipEventDestroy(e);   // destruction is deferred by the runtime until the recorded work has passed
    ++g_groups;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist)
{
    if (!comm || ndev < 1 || ndev > MAX_RANKS)
        return ncclInvalidArgument;
    MockSet *set = new MockSet;
    set->nranks = set->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        MockComm *c = new MockComm;
        c->set = set, c->rank = r, c->device = devlist ? devlist[r] : r;
        comm[r] = reinterpret_cast<ncclComm_t>(c);
    }
    if (getenv("RCCL_MOCK_VERBOSE"))
        fprintf(stderr, "rccl_mock: communicator set of %d ranks (a TEST DOUBLE, not RCCL)\n", ndev);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    MockComm *c = as_mock(comm);
    if (!c)
        return ncclInvalidArgument;
    if (--c->set->alive == 0) {
        if (getenv("RCCL_MOCK_VERBOSE"))
            fprintf(stderr, "rccl_mock: %ld grouped all-reduces of %ld calls in all\n", g_groups, g_allreduces);
        delete c->set;
    }
    c->magic = 0;
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0)
        return ncclInvalidUsage;
    if (--g_depth > 0)
        return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_group(ops);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    MockComm *c = as_mock(comm);
    if (!c || !sendbuff || !recvbuff || count != 3 || datatype != ncclDouble || op != ncclSum) {
        fprintf(stderr, "rccl_mock: ncclAllReduce(count %zu, type %d, op %d): only libmc_multi's (3, ncclDouble, ncclSum) on a mock communicator\n",
                count, (int)datatype, (int)op);
        return ncclInvalidArgument;
    }
    ++g_allreduces;
    g_ops.push_back({static_cast<const double *>(sendbuff), static_cast<double *>(recvbuff), c, stream});
    if (g_depth == 0) {   // an ungrouped call: complete only for a communicator of one
        std::vector<Op> ops;
        ops.swap(g_ops);
        return run_group(ops);
    }
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error (rccl_mock)";
    case ncclUnhandledCudaError: return "unhandled HIP error (rccl_mock)";
    case ncclInvalidArgument: return "invalid argument (rccl_mock)";
    case ncclInvalidUsage: return "invalid usage (rccl_mock)";
    default: return "error (rccl_mock)";
    }
}

}  // extern "C"
