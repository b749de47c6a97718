// mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that lets THREADS of one process be the ranks of a communicator on ONE GPU, so that
// nrf_allgather_tiles and nrf_allreduce_grads (comm.hip) -- group start / end, the all-gather of equal tiles, the grouped broadcasts of unequal ones, ranks that own no
// rows, the bucketed gradient all-reduce and its overflow agreement -- run at world
// sizes > 1 on a one-GPU box (RCCL itself refuses two ranks on a device).  It implements only the entry points comm.hip resolves, with RCCL's signatures and stream
// semantics: a collective is ENQUEUED on the caller's stream (copies between the ranks' buffers, ordered by events), never waited for on the host.
// Built by tests/helpers/build_mock_rccl.sh with SONAME librccl.so.1; the test binary links it, so comm.hip's dlopen(RTLD_NOLOAD) finds this copy.
// Reference: none (the reference is single-process); the interface is rccl/rccl.h's.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Op { int kind; const void *send; void *recv; size_t bytes; int root; hipStream_t st; };      // kind 0 all-gather, 1 broadcast, 2 all-reduce (fp32; root: 0 sum, 1 max)

// all-reduce: every rank first copies its contribution into a staging buffer of its own (in-place reductions overwrite the send buffer), then reduces the world's staged
// copies in RANK ORDER into its receive buffer -- one fixed order on every rank, so all ranks end with the same bits (RCCL's rings give that too; its order is its own)
struct Staged { const float *p[8]; };
__global__ void k_mock_allreduce(int world, Staged s, size_t n, int is_max, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = s.p[0][i];
    for (int q = 1; q < world; q++) a = is_max ? fmaxf(a, s.p[q][i]) : a + s.p[q][i];
    out[i] = a;
}

struct Group {
    int world = 0, joined = 0;
    std::mutex m; std::condition_variable cv;
    // one collective round: every rank deposits its op list + a "my inputs are ready" event, waits for all, enqueues its copies, deposits a "my copies are issued" event
    int arrived = 0, left = 0; long round = 0;
    std::vector<std::vector<Op>> ops;
    std::vector<hipEvent_t> ready, done;
    // all-reduce staging: stage[rank][op index] -> device buffer of at least stage_bytes[rank][op index]
    std::vector<std::vector<void *>> stage;
    std::vector<std::vector<size_t>> stage_bytes;
};

struct Comm { Group *g; int rank; };

std::mutex g_reg_mu;
std::map<unsigned long long, Group *> g_reg;
unsigned long long g_next_id = 1;
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
thread_local Comm *t_comm = nullptr;

unsigned long long id_key(const ncclUniqueId &id) { unsigned long long k; std::memcpy(&k, id.internal, sizeof k); return k; }

// all ranks meet; returns when every rank of the group has called it for this round
void barrier(Group *g, int &counter)
{
    std::unique_lock<std::mutex> lk(g->m);
    const long my_round = g->round;
    if (++counter == g->world) { counter = 0; g->round++; g->cv.notify_all(); }
    else g->cv.wait(lk, [&] { return g->round != my_round; });
}

ncclResult_t run_group(Comm *c, std::vector<Op> &ops)
{
    Group *g = c->g;
    if (ops.empty()) return ncclSuccess;
    hipStream_t st = ops[0].st;
    if ((int)ops.size() > 0) {
        auto &sb = g->stage[c->rank]; auto &sz = g->stage_bytes[c->rank];
        if (sb.size() < ops.size()) { sb.resize(ops.size(), nullptr); sz.resize(ops.size(), 0); }
        for (size_t i = 0; i < ops.size(); i++) {
            if (ops[i].kind != 2 || ops[i].bytes == 0) continue;
            if (g->world > 8) return ncclInvalidUsage;
            if (sz[i] < ops[i].bytes) {
                if (sb[i]) { if (hipStreamSynchronize(st) != hipSuccess || hipFree(sb[i]) != hipSuccess) return ncclUnhandledCudaError; }
                if (hipMalloc(&sb[i], ops[i].bytes) != hipSuccess) return ncclUnhandledCudaError;
                sz[i] = ops[i].bytes;
            }
            if (hipMemcpyAsync(sb[i], ops[i].send, ops[i].bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    if (hipEventRecord(g->ready[c->rank], st) != hipSuccess) return ncclUnhandledCudaError;        // everything this rank wrote before the collective (and its staged contributions)
    { std::lock_guard<std::mutex> lk(g->m); g->ops[c->rank] = ops; }
    barrier(g, g->arrived);                                                                      // every rank's op list and ready event are in place
    for (int q = 0; q < g->world; q++) if (hipStreamWaitEvent(st, g->ready[q], 0) != hipSuccess) return ncclUnhandledCudaError;
    const std::vector<Op> &mine = g->ops[c->rank];
    for (size_t i = 0; i < mine.size(); i++) {
        const Op &o = mine[i];
        if (o.kind == 0) {
            for (int q = 0; q < g->world; q++) {
                if (g->ops[q].size() != mine.size() || g->ops[q][i].kind != 0 || g->ops[q][i].bytes != o.bytes) return ncclInvalidUsage;       // mismatched collectives: RCCL would hang
                if (o.bytes && hipMemcpyAsync(static_cast<char *>(o.recv) + (size_t)q * o.bytes, g->ops[q][i].send, o.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
            }
        } else if (o.kind == 2) {
            Staged sp{};
            for (int q = 0; q < g->world; q++) {
                if (g->ops[q].size() != mine.size() || g->ops[q][i].kind != 2 || g->ops[q][i].root != o.root || g->ops[q][i].bytes != o.bytes) return ncclInvalidUsage;
                sp.p[q] = static_cast<const float *>(g->stage[q][i]);
            }
            const size_t n = o.bytes / 4;
            if (n) hipLaunchKernelGGL(k_mock_allreduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g->world, sp, n, o.root, static_cast<float *>(o.recv));
            if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
        } else {
            for (int q = 0; q < g->world; q++) if (g->ops[q].size() != mine.size() || g->ops[q][i].kind != 1 || g->ops[q][i].root != o.root || g->ops[q][i].bytes != o.bytes) return ncclInvalidUsage;
            const void *src = g->ops[o.root][i].send;
            if (o.bytes && src != o.recv && hipMemcpyAsync(o.recv, src, o.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    if (hipEventRecord(g->done[c->rank], st) != hipSuccess) return ncclUnhandledCudaError;
    barrier(g, g->left);                                                                         // every rank's copies are enqueued and its done event recorded
    for (int q = 0; q < g->world; q++) if (hipStreamWaitEvent(st, g->done[q], 0) != hipSuccess) return ncclUnhandledCudaError;   // nobody overwrites a buffer a peer still reads
    barrier(g, g->arrived);                                                                      // the op lists may be replaced
    return ncclSuccess;
}

ncclResult_t submit(Comm *c, const Op &o)
{
    if (t_depth > 0) { if (t_comm && t_comm != c) return ncclInvalidUsage; t_comm = c; t_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_group(c, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    std::memset(id, 0, sizeof *id);
    const unsigned long long k = g_next_id++;
    std::memcpy(id->internal, &k, sizeof k);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Group *g;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        Group *&slot = g_reg[id_key(id)];
        if (!slot) {
            slot = new Group(); slot->world = nranks; slot->ops.resize(nranks); slot->ready.resize(nranks); slot->done.resize(nranks);
            slot->stage.resize(nranks); slot->stage_bytes.resize(nranks);
            for (int q = 0; q < nranks; q++) {
                if (hipEventCreateWithFlags(&slot->ready[q], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&slot->done[q], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            }
        }
        g = slot;
        if (g->world != nranks) return ncclInvalidArgument;
    }
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->joined++;
        g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->joined >= g->world; });                                    // the rendezvous: blocks until every rank has arrived, as RCCL's does
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{g, rank});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm *>(comm); return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { if (!comm || !count) return ncclInvalidArgument; *count = reinterpret_cast<Comm *>(comm)->g->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { if (!comm || !rank) return ncclInvalidArgument; *rank = reinterpret_cast<Comm *>(comm)->rank; return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t *e) { if (e) *e = ncclSuccess; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (mock: the ranks' collectives do not match)" : "error (mock RCCL)"; }

ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    Comm *c = t_comm; t_comm = nullptr;
    return c ? run_group(c, ops) : ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclFloat) return ncclInvalidArgument;
    return submit(reinterpret_cast<Comm *>(comm), Op{0, sendbuff, recvbuff, sendcount * 4, 0, stream});
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclFloat || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    return submit(reinterpret_cast<Comm *>(comm), Op{2, sendbuff, recvbuff, count * 4, op == ncclMax ? 1 : 0, stream});
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclFloat) return ncclInvalidArgument;
    return submit(reinterpret_cast<Comm *>(comm), Op{1, sendbuff, recvbuff, count * 4, root, stream});
}

}  // extern "C"
