// comm.hip -- multi-GPU sharding of the render path behind the C ABI (SURVEY 8e / 8b `nrf_allgather_tiles`).
//
// Whole-image render batches are partitioned by ray into contiguous ROW TILES (rank r of R renders rows [row0_r, row0_r + rows_r) of the
// frame; the row-major pixel <-> ray mapping of RayUtils.h:5-21 is untouched), the model is replicated read-only, and ONE collective per
// step returns the per-tile pixels to every rank: ncclAllGather (RCCL over xGMI) straight into the frame when the tiles are equal, a
// grouped set of ncclBroadcast (the all-gather-v idiom, still one fused launch) when H does not divide by the world size.  Payload:
// 800 x 800 x 3 fp32 = 7.7 MB per frame over all ranks -- latency-bound on the 7 x 153 GB/s point-to-point links.
// The reference is single-process / single-GPU: no counterpart.
//
// RCCL is resolved at first use with dlopen (the copy already mapped into the process -- LibTorch's -- if there is one, so that a
// communicator handed over by the host and the calls made here belong to the same library), so libnerfpp_hip.so itself carries no RCCL
// dependency and loads on a box without it.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

struct nrf_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
    // nrf_allreduce_grads: the overflow word the ranks agree on before any gradient is exchanged (device float + pinned mirror; created on first use)
    mutable float *d_word = nullptr, *h_word = nullptr;
};

namespace nrf {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // optional: the state of a non-blocking communicator a host may hand over (nrf_comm_wrap)
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;
static char g_rccl_err[256] = "";

static void rccl_load()
{
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    // NRF_RCCL_LIBRARY: the host names the RCCL copy to use (a process that maps several; tests: the threads-as-ranks stand-in beside torch's own RCCL)
    if (const char *forced = getenv("NRF_RCCL_LIBRARY")) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) { snprintf(g_rccl_err, sizeof(g_rccl_err), "NRF_RCCL_LIBRARY=%s: %s", forced, dlerror()); return; }
    }
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy the host process already uses
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { snprintf(g_rccl_err, sizeof(g_rccl_err), "RCCL not found: %s", dlerror()); return; }
    Rccl r;
    r.handle = h;
    bool ok = true;
#define NRF_SYM(field, name) do { r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { ok = false; snprintf(g_rccl_err, sizeof(g_rccl_err), "RCCL lacks %s", name); } } while (0)
    NRF_SYM(GetUniqueId, "ncclGetUniqueId"); NRF_SYM(CommInitRank, "ncclCommInitRank"); NRF_SYM(CommDestroy, "ncclCommDestroy");
    NRF_SYM(CommCount, "ncclCommCount"); NRF_SYM(CommUserRank, "ncclCommUserRank"); NRF_SYM(AllGather, "ncclAllGather");
    NRF_SYM(Broadcast, "ncclBroadcast"); NRF_SYM(GroupStart, "ncclGroupStart"); NRF_SYM(GroupEnd, "ncclGroupEnd");
    NRF_SYM(GetErrorString, "ncclGetErrorString"); NRF_SYM(AllReduce, "ncclAllReduce");
#undef NRF_SYM
    r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
    if (ok) g_rccl = r;
}

static const Rccl *rccl()
{
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.handle) { set_error("%s", g_rccl_err); return nullptr; }
    return &g_rccl;
}

#define NRF_NCCL(R, call)                                                                                \
    do {                                                                                                 \
        ncclResult_t e_ = (call);                                                                        \
        if (e_ != ncclSuccess) {                                                                         \
            ::nrf::set_error("%s failed: %s (%s:%d)", #call, (R)->GetErrorString(e_), __FILE__, __LINE__); \
            return NRF_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)

// ncclGroupEnd's answer, waited out when a NON-BLOCKING communicator (handed over through nrf_comm_wrap) says ncclInProgress: the group is enqueued on RCCL's helper
// thread there, and nothing may be ordered behind the call on the stream (an event, the caller's next kernel) before the enqueue has happened -- bounded wait
static ncclResult_t settle_group_end(const Rccl *R, const nrf_comm *c, ncclResult_t end)
{
    if (end == ncclInProgress && R->CommGetAsyncError) {
        const auto t0 = std::chrono::steady_clock::now();
        ncclResult_t q = R->CommGetAsyncError(c->comm, &end);
        while (q == ncclSuccess && end == ncclInProgress && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 120.0) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            q = R->CommGetAsyncError(c->comm, &end);
        }
        if (q != ncclSuccess) end = q;
    }
    return end;
}

__global__ void k_scale_inplace(int64_t n, float f, float *__restrict__ p)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * f;
}

static void partition(int h, int world, int rank, int *row0, int *rows)
{
    const int base = h / world, rem = h % world;            // the first `rem` ranks take one row more
    *rows = base + (rank < rem ? 1 : 0);
    *row0 = rank * base + (rank < rem ? rank : rem);
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_tile_partition(int h, int world, int rank, int *row0, int *rows)
{
    NRF_CHECK_ARG(row0 && rows, "nrf_tile_partition: null pointer");
    NRF_CHECK_ARG(h >= 0 && world >= 1 && rank >= 0 && rank < world, "nrf_tile_partition: need h >= 0 and 0 <= rank < world (h %d, world %d, rank %d)", h, world, rank);
    partition(h, world, rank, row0, rows);
    return NRF_OK;
}

int nrf_comm_unique_id(void *id_out)
{
    NRF_CHECK_ARG(id_out, "nrf_comm_unique_id: null pointer");
    const Rccl *R = rccl();
    if (!R) return NRF_ERR_UNSUPPORTED;
    ncclUniqueId id;
    NRF_NCCL(R, R->GetUniqueId(&id));
    static_assert(sizeof(id) == NRF_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return NRF_OK;
}

int nrf_comm_create_timeout(const void *id, int world, int rank, double timeout_s, nrf_comm **out)
{
    NRF_CHECK_ARG(id && out, "nrf_comm_create: null pointer");
    NRF_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "nrf_comm_create: need 0 <= rank < world (world %d, rank %d)", world, rank);
    const Rccl *R = rccl();
    if (!R) return NRF_ERR_UNSUPPORTED;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    if (timeout_s > 0.0) {
        // Bounded rendezvous with a BLOCKING communicator: the plain ncclCommInitRank runs on a helper thread (bound to the caller's device) and this thread waits for it
        // with a deadline -- a peer that never starts, or one that was handed a stale id, would otherwise park this rank in the rendezvous for ever.  The communicator
        // itself stays an ordinary blocking one, so ncclGroupEnd / ncclCommDestroy return only once the work is enqueued (a non-blocking communicator answers
        // ncclInProgress there at world > 1 and enqueues on a helper thread: events recorded after the call would then run ahead of the collective).
        // On timeout the helper is left behind in its rendezvous (it owns its state; there is no handle to abort yet) and the caller is expected to exit non-zero.
        struct InitJob {
            std::mutex m; std::condition_variable cv;
            bool done = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr;
            ncclUniqueId uid; int world = 1, rank = 0, dev = 0;
        };
        auto job = std::make_shared<InitJob>();
        job->uid = uid; job->world = world; job->rank = rank;
        if (hipGetDevice(&job->dev) != hipSuccess) { set_error("nrf_comm_create: no current HIP device"); return NRF_ERR_HIP; }
        const auto init = R->CommInitRank;
        std::thread([job, init]() {
            ncclComm_t cc = nullptr;
            ncclResult_t e = hipSetDevice(job->dev) == hipSuccess ? init(&cc, job->world, job->uid, job->rank) : ncclUnhandledCudaError;
            { std::lock_guard<std::mutex> g(job->m); job->res = e; job->comm = cc; job->done = true; }
            job->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(job->m);
        if (!job->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return job->done; })) {
            set_error("nrf_comm_create: rank %d of %d waited %.0f s for its peers (a rank that never started, or a communicator id of another launch); giving up",
                      rank, world, timeout_s);
            return NRF_ERR_HIP;
        }
        if (job->res != ncclSuccess) { set_error("ncclCommInitRank failed: %s", R->GetErrorString(job->res)); return NRF_ERR_HIP; }
        c = job->comm;
    } else {
        NRF_NCCL(R, R->CommInitRank(&c, world, uid, rank));     // on the calling thread's current HIP device; blocks until all ranks arrive
    }
    nrf_comm *cm = new nrf_comm();
    cm->comm = c; cm->world = world; cm->rank = rank; cm->owned = true;
    *out = cm;
    return NRF_OK;
}

int nrf_comm_create(const void *id, int world, int rank, nrf_comm **out)
{
    // default bound: NRF_COMM_TIMEOUT_S seconds (300).  Only a well-formed number is taken; only an explicit 0 selects "wait for ever" (the plain blocking
    // ncclCommInitRank on the calling thread) -- an empty or non-numeric value keeps the default instead of silently disabling the bound
    double t = 300.0;
    if (const char *e = getenv("NRF_COMM_TIMEOUT_S")) {
        char *end = nullptr;
        const double v = strtod(e, &end);
        while (end && (*end == ' ' || *end == '\t')) end++;
        if (end != e && end && *end == '\0' && v >= 0.0 && v == v) t = v;
    }
    return nrf_comm_create_timeout(id, world, rank, t, out);
}

int nrf_comm_wrap(void *nccl_comm, nrf_comm **out)
{
    NRF_CHECK_ARG(nccl_comm && out, "nrf_comm_wrap: null pointer");
    const Rccl *R = rccl();
    if (!R) return NRF_ERR_UNSUPPORTED;
    nrf_comm *cm = new nrf_comm();
    cm->comm = static_cast<ncclComm_t>(nccl_comm);
    ncclResult_t e1 = R->CommCount(cm->comm, &cm->world), e2 = R->CommUserRank(cm->comm, &cm->rank);
    if (e1 != ncclSuccess || e2 != ncclSuccess) { delete cm; set_error("nrf_comm_wrap: not a live ncclComm_t of the RCCL mapped into this process"); return NRF_ERR_INVALID_ARG; }
    *out = cm;
    return NRF_OK;
}

void nrf_comm_destroy(nrf_comm *c)
{
    if (!c) return;
    if (c->d_word) (void)hipFree(c->d_word);
    if (c->h_word) (void)hipHostFree(c->h_word);
    if (c->owned && c->comm && g_rccl.handle) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int nrf_comm_world(const nrf_comm *c) { return c ? c->world : 0; }
int nrf_comm_rank(const nrf_comm *c) { return c ? c->rank : -1; }

int nrf_allgather_tiles(const nrf_comm *c, const float *d_tiles, int frames, int h, int w, int ch, float *d_frames, void *stream)
{
    NRF_CHECK_ARG(c && c->comm, "nrf_allgather_tiles: null communicator");
    NRF_CHECK_ARG(frames >= 0 && h >= 0 && w >= 0 && ch >= 1, "nrf_allgather_tiles: bad sizes");
    if (frames == 0 || h == 0 || w == 0) return NRF_OK;
    NRF_CHECK_ARG(d_frames, "nrf_allgather_tiles: null pointer");
    const Rccl *R = rccl();
    if (!R) return NRF_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    int row0, rows;
    partition(h, c->world, c->rank, &row0, &rows);
    // a rank that owns no rows (h < world) has nothing to send: its d_tiles may be NULL -- it still takes part in the collective, or its peers would hang
    NRF_CHECK_ARG(d_tiles || rows == 0, "nrf_allgather_tiles: null tile buffer on a rank that owns %d rows", rows);
    const size_t px = (size_t)w * ch, tile = (size_t)rows * px, frame = (size_t)h * px;
    const bool even = (h % c->world) == 0;
    // one fused launch for all frames of the step (ncclGroupStart / End aggregates the operations).  The group is closed on EVERY path: the first failure is
    // remembered, the remaining operations are skipped, GroupEnd runs, and only then is the error returned -- a dangling group would swallow every later
    // collective of this thread.
    NRF_NCCL(R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (int f = 0; f < frames && first == ncclSuccess; f++) {
        const float *src = d_tiles ? d_tiles + (size_t)f * tile : d_frames;      // never dereferenced when rows == 0 (this rank is the root of no broadcast)
        float *dst = d_frames + (size_t)f * frame;
        if (even) {
            first = R->AllGather(src, dst, tile, ncclFloat, c->comm, st); what = "ncclAllGather";
        } else {
            for (int r = 0; r < c->world && first == ncclSuccess; r++) {
                int r0, rr;
                partition(h, c->world, r, &r0, &rr);
                if (rr == 0) continue;
                first = R->Broadcast(src, dst + (size_t)r0 * px, (size_t)rr * px, ncclFloat, r, c->comm, st); what = "ncclBroadcast";   // sendbuff is read on the root only
            }
        }
    }
    const ncclResult_t end = settle_group_end(R, c, R->GroupEnd());
    if (first != ncclSuccess) { set_error("nrf_allgather_tiles: %s failed: %s", what, R->GetErrorString(first)); return NRF_ERR_HIP; }
    if (end != ncclSuccess) { set_error("nrf_allgather_tiles: ncclGroupEnd failed: %s", R->GetErrorString(end)); return NRF_ERR_HIP; }
    return NRF_OK;
}

// The data-parallel training step's exchange ("the only real collective"; 5, last row: 64 MiB of table gradient per step at L16 T2^19 F2).
// Every rank has rendered and back-propagated ITS ray batch against the replicated model; the gradients become their mean over the ranks IN PLACE: one ncclAllReduce(sum)
// per `bucket_bytes` slice (xGMI is point-to-point: a ring moves 2 (N - 1) / N of the payload over every link, several slices in flight keep the links busy while the
// previous slice is reduced), all of one group launch, then one multiply by 1 / world per buffer -- on `stream`, nothing waits.
// overflow >= 0 asks for the agreement FIRST: the fp16 backward's overflow report (nrf_mlp_backward_f16_flags) is per rank, the decision to skip the optimizer step must not
// be -- a replica that steps while another skips, or that sums a peer's inf, ends with different parameters, moments and step counts.  The ranks all-reduce (max) one word
// and read it back (one host synchronisation, as a loss scaler costs); *skip_out = 1 on EVERY rank iff any rank passed overflow != 0, and the gradients are then left untouched.
int nrf_allreduce_grads(const nrf_comm *c, float *const *d_grads, const int64_t *counts, int n_grads, int64_t bucket_bytes, int overflow, int *skip_out, void *stream)
{
    NRF_CHECK_ARG(c && c->comm, "nrf_allreduce_grads: null communicator");
    NRF_CHECK_ARG(n_grads >= 0 && (n_grads == 0 || (d_grads && counts)) && bucket_bytes >= 0, "nrf_allreduce_grads: bad argument");
    NRF_CHECK_ARG(overflow < 0 || skip_out, "nrf_allreduce_grads: the overflow agreement needs skip_out");
    for (int i = 0; i < n_grads; i++) NRF_CHECK_ARG(counts[i] >= 0 && (counts[i] == 0 || d_grads[i]), "nrf_allreduce_grads: gradient %d: null buffer / negative count", i);
    if (skip_out) *skip_out = 0;
    if (c->world == 1) { if (skip_out) *skip_out = overflow > 0; return NRF_OK; }          // the mean over one rank
    const Rccl *R = rccl();
    if (!R) return NRF_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (overflow >= 0) {
        if (!c->d_word) {
            NRF_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_word), sizeof(float)));
            NRF_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_word), sizeof(float), hipHostMallocDefault));
        }
        *c->h_word = overflow ? 1.0f : 0.0f;
        NRF_HIP(hipMemcpyAsync(c->d_word, c->h_word, sizeof(float), hipMemcpyHostToDevice, st));
        NRF_NCCL(R, R->GroupStart());
        const ncclResult_t e1 = R->AllReduce(c->d_word, c->d_word, 1, ncclFloat, ncclMax, c->comm, st);
        const ncclResult_t e2 = settle_group_end(R, c, R->GroupEnd());
        if (e1 != ncclSuccess || e2 != ncclSuccess) { set_error("nrf_allreduce_grads: the overflow agreement failed: %s", R->GetErrorString(e1 != ncclSuccess ? e1 : e2)); return NRF_ERR_HIP; }
        NRF_HIP(hipMemcpyAsync(c->h_word, c->d_word, sizeof(float), hipMemcpyDeviceToHost, st));
        NRF_HIP(hipStreamSynchronize(st));
        if (*c->h_word > 0.0f) { *skip_out = 1; return NRF_OK; }
    }
    const int64_t bucket = bucket_bytes > 0 ? (bucket_bytes / 4 > 0 ? bucket_bytes / 4 : 1) : ((int64_t)32 << 20) / 4;          // elements; default 32 MiB
    NRF_NCCL(R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n_grads && first == ncclSuccess; i++)
        for (int64_t off = 0; off < counts[i] && first == ncclSuccess; off += bucket) {
            const int64_t m = counts[i] - off < bucket ? counts[i] - off : bucket;
            first = R->AllReduce(d_grads[i] + off, d_grads[i] + off, (size_t)m, ncclFloat, ncclSum, c->comm, st);
        }
    const ncclResult_t end = settle_group_end(R, c, R->GroupEnd());          // closed on every path (a dangling group would swallow every later collective of this thread)
    if (first != ncclSuccess) { set_error("nrf_allreduce_grads: ncclAllReduce failed: %s", R->GetErrorString(first)); return NRF_ERR_HIP; }
    if (end != ncclSuccess) { set_error("nrf_allreduce_grads: ncclGroupEnd failed: %s", R->GetErrorString(end)); return NRF_ERR_HIP; }
    const float inv = 1.0f / (float)c->world;
    for (int i = 0; i < n_grads; i++) {
        if (counts[i] == 0) continue;
        hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)ceil_div(counts[i], 256)), dim3(256), 0, st, counts[i], inv, d_grads[i]);
        NRF_LAUNCH_CHECK();
    }
    return NRF_OK;
}

}  // extern "C"
