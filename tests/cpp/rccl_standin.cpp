// rccl_standin.cpp -- TEST INFRASTRUCTURE, not a collective library.
//
// A five-symbol stand-in for librccl that lets the RCCL branch of rd_node_batch_* (rawdev.hip: ncclCommInitAll, the
// grouped in-place ncclAllReduce loop, the "every device holds the sum" read-back) execute with n > 1 ranks on a ONE-GPU
// box, where real RCCL refuses two ranks on one device.  It proves the call sequence, the buffer / stream / communicator
// wiring and the dealing of rd_node_batch; it says NOTHING about xGMI, RCCL's algorithms or their performance.
// The reduction is done through host memory when the group ends: every queued buffer is read back after its stream has
// drained, summed element-wise (u64 SUM only -- the one collective librawdev issues), and the sum is written to every
// rank's receive buffer.  librawdev recognises the stand-in by the exported marker `rawdev_rccl_standin` and only then
// accepts a device listed twice (include/rawdev.h, rd_node_batch_create).
//
// Build (tests/test_gpu_node_batch.py does it): hipcc -O1 -fPIC -shared -o tests/cpp/librccl_standin.so tests/cpp/rccl_standin.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace {
struct group;
struct comm {
    group *grp;
    int rank, device;
};
struct group {
    int n;
    std::vector<comm *> members;
    int alive;
};
struct pending {
    const void *send;
    void *recv;
    size_t count;
    comm *c;
    hipStream_t stream;
};
std::mutex g_mu;
int g_depth = 0;
std::vector<pending> g_queue;
unsigned long long g_calls = 0, g_groups = 0, g_max_ranks = 0;
char g_why[256] = "stand-in: HIP call failed";

int hip_fail(const char *what, hipError_t e)
{
    snprintf(g_why, sizeof g_why, "stand-in: %s failed: %s", what, hipGetErrorString(e));
    return 1;
}
#define SI_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(#call, e_); } while (0)

int flush()
{
    if (g_queue.empty()) return 0;
    const size_t count = g_queue[0].count;
    group *grp = g_queue[0].c->grp;
    if ((int)g_queue.size() != grp->n) return 5;                 // ncclInvalidUsage: every rank of the communicator must call
    for (const pending &p : g_queue)
        if (p.count != count || p.c->grp != grp) return 5;
    std::vector<uint64_t> sum(count, 0), part(count);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (const pending &p : g_queue) {
        SI_HIP(hipSetDevice(p.c->device));
        SI_HIP(hipStreamSynchronize(p.stream));                  // what was enqueued before the collective has run
        SI_HIP(hipMemcpy(part.data(), p.send, count * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < count; ++i) sum[i] += part[i];
    }
    for (const pending &p : g_queue) {
        SI_HIP(hipSetDevice(p.c->device));
        SI_HIP(hipMemcpy(p.recv, sum.data(), count * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    (void)hipSetDevice(prev);
    if (g_queue.size() > g_max_ranks) g_max_ranks = g_queue.size();
    ++g_groups;
    g_queue.clear();
    return 0;
}
}  // namespace

extern "C" {
int rawdev_rccl_standin = 1;                                     // the marker librawdev looks for

// what the test reads back: all-reduce calls seen, groups completed, most ranks in one group
void rawdev_rccl_standin_stats(unsigned long long *calls, unsigned long long *groups, unsigned long long *max_ranks)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (calls) *calls = g_calls;
    if (groups) *groups = g_groups;
    if (max_ranks) *max_ranks = g_max_ranks;
}

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return 4;                            // ncclInvalidArgument
    group *g = new group{ ndev, {}, ndev };
    for (int i = 0; i < ndev; ++i) {
        comm *c = new comm{ g, i, devlist ? devlist[i] : i };
        g->members.push_back(c);
        comms[i] = c;
    }
    return 0;
}

int ncclCommDestroy(void *cv)
{
    comm *c = (comm *)cv;
    if (!c) return 4;
    std::lock_guard<std::mutex> l(g_mu);
    group *g = c->grp;
    delete c;
    if (--g->alive == 0) delete g;
    return 0;
}

int ncclGroupStart()
{
    std::lock_guard<std::mutex> l(g_mu);
    ++g_depth;
    return 0;
}

int ncclGroupEnd()
{
    std::lock_guard<std::mutex> l(g_mu);
    if (g_depth <= 0) return 5;
    if (--g_depth == 0) return flush();
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *cv, hipStream_t stream)
{
    if (!send || !recv || !cv) return 4;
    if (dtype != 5 /* ncclUint64 */ || op != 0 /* ncclSum */) return 4;
    std::lock_guard<std::mutex> l(g_mu);
    ++g_calls;
    g_queue.push_back(pending{ send, recv, count, (comm *)cv, stream });
    if (g_depth == 0) return flush();                            // ungrouped: legal only for a one-rank communicator
    return 0;
}

const char *ncclGetErrorString(int r)
{
    switch (r) {
    case 0: return "no error";
    case 1: return g_why;
    case 4: return "stand-in: invalid argument (only u64 SUM all-reduce exists here)";
    case 5: return "stand-in: invalid usage (every rank of the communicator must join the group)";
    default: return "stand-in: unknown error";
    }
}
}
