// render.hip -- the renderer behind NeRFRenderer<TEmbedder,TEmbedDirs,TNeRF> (NeRFRenderer.h:88-159):
//   RunNetwork  :164-194     RenderRays  :366-459
// plus library-wide bookkeeping (error text, profiling events).
//
// The deterministic render path (Perturb = 0, RawNoiseStd = 0, ThinRay: what FillRenderParams configures for test-time
// rendering, NeRFExecutor.h:379-415) forms sample points inside the encoders and never materialises them.  The stochastic
// branches (jitter, cone rays / TangentScatter, sigma noise, preconditioning) add one point-generation launch per pass
// (stoch.hip) whose draws are counter-based, and then run the same network kernels on explicit points.  One network
// serves both passes and the fine pass re-evaluates all S + N_importance depths (NeRFRenderer.h:422,447).
#include "encode.h"
#include "hash_fast.h"
#include "mlp.h"
#include "stoch.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

constexpr int NRF_MAX_LANES = 4;

struct nrf_renderer {
    nrf_renderer_desc desc;
    int in_ch = 0, in_views = 0;
    // the lanes of the Chunk loop (nrf_batchify_rays): auxiliary streams and the fork / join events, created on first use on the device that is current then and
    // re-created when a later call comes on another device.  A renderer serves one device and one caller at a time (include/nerfpp_hip.h, nrf_batchify_rays).
    mutable std::mutex lane_mu;
    mutable hipStream_t lane[NRF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    mutable hipEvent_t lane_fork = nullptr, lane_done[NRF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    mutable int lane_device = -1;
    int lanes = 0;                // this renderer's lane count; 0: the process-wide default (nrf_set_render_lanes / NRF_RENDER_LANES)
    // non-finite words of the matrix-core precisions (nrf_render_params.overflow_policy): one per chunk of the current call on the device, a pinned host mirror for the
    // synchronous policies and one for the deferred copy, and the event that says the deferred copy has landed
    mutable uint32_t *d_flags = nullptr, *h_flags = nullptr, *h_deferred = nullptr;      // h_deferred: NRF_DEFERRED_RING x NRF_FLAG_SLOTS words
    mutable hipEvent_t deferred_ev[32] = {};
    mutable int deferred_slots[32] = {};               // > 0: a deferred copy of that many words is pending in ring entry i
    mutable int deferred_head = 0;                     // ring entry the next deferred copy goes to (entries are filled and looked at in order)
    mutable int64_t flagged_chunks = 0, rerendered_chunks = 0;
    // where the most recent SINGLE-chunk render of the feature-reusing fast path (CuHashEmbedder mode) left its hash features in the caller's workspace: the level-major
    // table, its column count, the keep mask by column and the merge map [n, sf] (nrf_renderer_last_features: the training backward reads them instead of encoding the
    // fine points again).  Invalidated by every render call on entry.
    mutable struct { const void *feats = nullptr; int64_t cols = 0; const uint8_t *keep = nullptr; const int32_t *src = nullptr; int64_t n = 0; int sf = 0; bool valid = false; } last_view;
    mutable uint64_t chunk_serial = 0;                 // chunks rendered so far: a caller that saw serial k and still sees k knows that no render has touched the view since
    void drop_lanes() const
    {
        for (auto &st : lane) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
        for (auto &e : lane_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (lane_fork) { (void)hipEventDestroy(lane_fork); lane_fork = nullptr; }
        lane_device = -1;
    }
    ~nrf_renderer()
    {
        drop_lanes();
        for (auto &e : deferred_ev) if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
        if (d_flags) (void)hipFree(d_flags);
        if (h_flags) (void)hipHostFree(h_flags);
        if (h_deferred) (void)hipHostFree(h_deferred);
    }
};
constexpr int NRF_FLAG_SLOTS = 4096;        // chunks of one call that have a word of their own (further chunks share them modulo this)
constexpr int NRF_DEFERRED_RING = 32;       // NRF_OVERFLOW_DEFERRED: calls whose words may be in flight to the host at once (the host may run that many calls ahead before a
                                            // call waits for the oldest one's copy; 512 KB of pinned memory)

namespace nrf {

// ---------------------------------------------------------------------------------------------------
// error text
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------------
// profiling: HIP events around the dominant kernels, recorded on the caller's stream
// ---------------------------------------------------------------------------------------------------
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
struct ProfRec { int slot; hipEvent_t e0, e1; };
static std::vector<ProfRec> g_prof_pending;
static std::vector<hipEvent_t> g_prof_pool;          // timing events are created once and recycled by nrf_profile_read: no hipEventCreate on a launch path
static double g_prof_ms[NRF_PROF_COUNT];
static int64_t g_prof_n[NRF_PROF_COUNT];

static hipEvent_t prof_event()
{
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

ProfScope::ProfScope(int slot_, hipStream_t stream_) : slot(slot_), stream(stream_), active(g_prof_on.load(std::memory_order_relaxed))
{
    if (!active) return;
    e0 = prof_event(); e1 = prof_event();
    if (!e0 || !e1) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (e0) g_prof_pool.push_back(e0);
        if (e1) g_prof_pool.push_back(e1);
        active = false;
        return;
    }
    (void)hipEventRecord(e0, stream);
}

ProfScope::~ProfScope()
{
    if (!active) return;
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back({slot, e0, e1});
}

// raw[i][3] = 0 where the embedder's keep_mask is false (NeRFRenderer.h:187-188)
__global__ void k_mask_sigma(int64_t p, int c, const uint8_t *__restrict__ keep, float *__restrict__ raw)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p && !keep[i]) raw[i * c + (c - 1)] = 0.0f;
}

// matrix-core precisions on HashEmbedder rows: a point outside the box gets zero features (see k_hash_ngp_lm: extrapolated features leave the fp16 range, sigma is masked anyway)
__global__ void k_zero_unkept_rows(int64_t p, int width, int stride, const uint8_t *__restrict__ keep, float *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p * width) return;
    const int64_t row = i / width;
    if (!keep[row]) x[row * stride + (i - row * width)] = 0.0f;
}

// raw_f[i] = the network output of fine depth i: computed in the coarse pass (columns [0, n_coarse) of the merge map) or in the fine pass's evaluation of the new samples
__global__ void k_gather_raw(int64_t p, const int32_t *__restrict__ src, const float4 *__restrict__ raw_coarse, const float4 *__restrict__ raw_new, int64_t n_coarse,
                             float4 *__restrict__ raw_f)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    const int64_t c = src[i];
    raw_f[i] = c < n_coarse ? raw_coarse[c] : raw_new[c - n_coarse];
}

struct Bump {
    char *base;
    size_t off = 0, cap;
    Bump(void *b, size_t c) : base(static_cast<char *>(b)), cap(c) {}
    template <class T> T *take(size_t count)
    {
        off = align_up(off, 256);
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
};

static size_t network_ws_bytes(const nrf_renderer *r, int64_t p, int prec)
{
    size_t b = 0;
    b += align_up((size_t)p * (r->in_ch + r->in_views) * sizeof(float), 256);    // concatenated MLP input
    b += align_up((size_t)p, 256);                                               // keep mask
    b += align_up((size_t)p * 3 * sizeof(float), 256);                           // explicit points (PE path)
    b += align_up(mlp_workspace_bytes(r->desc.mlp, p, prec), 256) + 1024;
    return b;
}

// The fast path (matrix-core precisions, a hash grid of either encoder with F = 2 and 16 levels, SH directions, NeRFSmall in the built
// matrix-core family): level-major fp16 features -> MFMA MLP, no concatenated input.  CuHashEmbedder features are exact fp16 numbers;
// HashEmbedder (LibTorch, fp32) features travel as hi + lo planes in the split-precision mode.
// The split-precision image of a NeRFSmall network is scaled for the magnitude of its inputs (mlp.h, SMALL_MAX_GROUPS): a renderer on a hash grid points the network at
// the grid's table RMS and has the scales re-derived whenever the table was uploaded since (a training loop: every step).  Called on the caller's stream BEFORE any
// lane forks: the images are rewritten in that stream's order.  The handles are shared mutable state of one caller (nerfpp_hip.h, nrf_batchify_rays).
static int ensure_scales(const nrf_renderer *r, hipStream_t st)
{
    const nrf_hash *h = r->desc.hash;
    nrf_mlp *m = const_cast<nrf_mlp *>(r->desc.mlp);
    if (!h || !m || m->family != MLP_SMALL || !m->d_group || !h->d_table_rms) return NRF_OK;
    if (m->d_in_rms_src == h->d_table_rms && m->in_rms_seen == h->table_version) return NRF_OK;
    NRF_TRY(hash_table_rms_update(const_cast<nrf_hash *>(h), st));
    m->d_in_rms_src = h->d_table_rms; m->in_rms_seen = h->table_version;
    return mlp_small_rescale(m, st);
}

static bool fast_path(const nrf_renderer *r, int prec)
{
    return (prec == NRF_PREC_F16_MFMA || prec == NRF_PREC_F16_SPLIT) && r->desc.hash && hash_fast_supported(r->desc.hash) && r->in_ch == 32 &&
           (r->desc.dirs_encoder == NRF_DIRS_SH_CUDA || r->desc.dirs_encoder == NRF_DIRS_SH_LIBTORCH) && (r->in_views == 16 || r->in_views == 64) &&
           mlp_small_mfma_available(r->desc.mlp);
}

static size_t fast_ws_bytes(const nrf_renderer *r, int64_t n, int64_t p)
{
    return align_up((size_t)p * 16 * sizeof(__half2), 256) * 2 + align_up((size_t)p, 256) + align_up((size_t)n * r->in_views * sizeof(__half), 256) + 1024;
}

// dirs_f16: per-ray direction features [n, V] prepared once per chunk (nullptr: computed here from `viewdirs`)
static int run_network_fast(const nrf_renderer *r, const PointSource &ps, const __half *dirs_f16, const __half *dirs_lo, int64_t n, int s, float *raw, void *ws,
                            size_t ws_bytes, hipStream_t st)
{
    const int64_t p = n * s;
    if (p == 0) return NRF_OK;
    Bump bump(ws, ws_bytes);
    const bool ngp = r->desc.hash->desc.mode == NRF_HASH_NGP;
    const bool want_lo = ngp && dirs_lo != nullptr;                       // split precision on fp32-valued features
    __half2 *feats = bump.take<__half2>((size_t)p * 16 * (want_lo ? 2 : 1));
    uint8_t *keep = bump.take<uint8_t>((size_t)p);
    if (bump.off > ws_bytes) { set_error("run_network_fast: workspace too small"); return NRF_ERR_WORKSPACE; }
    if (ngp) NRF_TRY(launch_hash_ngp_lm(r->desc.hash, ps, p, feats, p, want_lo ? p * 16 : 0, keep, st));
    else NRF_TRY(launch_hash_lm(r->desc.hash, ps, p, feats, p, keep, HASH_LM_DEFAULT_VARIANT, st));
    return mlp_small_forward_mfma_lm(r->desc.mlp, feats, want_lo ? feats + p * 16 : nullptr, p, dirs_f16, dirs_lo, s, keep, p, raw, st);
}

// Coarse pass of a hierarchical render on the fast path: sigma only, exact fp32 on the matrix cores (sigma_small_f32.hip).  The features are the
// fast path's own (CuHashEmbedder: level-major fp16, exact; HashEmbedder: one level-major fp32 plane), so sigma equals NRF_PREC_F32's bit for bit.
static int run_sigma_fast(const nrf_renderer *r, const PointSource &ps, int64_t n, int s, float *sigma, void *ws, size_t ws_bytes, hipStream_t st)
{
    const int64_t p = n * s;
    if (p == 0) return NRF_OK;
    Bump bump(ws, ws_bytes);
    const bool ngp = r->desc.hash->desc.mode == NRF_HASH_NGP;
    __half2 *feats = bump.take<__half2>((size_t)p * 16 * (ngp ? 2 : 1));
    uint8_t *keep = bump.take<uint8_t>((size_t)p);
    if (bump.off > ws_bytes) { set_error("run_sigma_fast: workspace too small"); return NRF_ERR_WORKSPACE; }
    if (ngp) NRF_TRY(launch_hash_ngp_lm(r->desc.hash, ps, p, feats, p, 0, keep, st, true));
    else NRF_TRY(launch_hash_lm(r->desc.hash, ps, p, feats, p, keep, HASH_LM_DEFAULT_VARIANT, st));
    return mlp_small_sigma_f32_lm(r->desc.mlp, feats, ngp ? 1 : 0, p, keep, p, sigma, st);
}

// Feature reuse across the two passes of a hierarchical render (both hash encoders' fast paths, deterministic sample points): the fine pass evaluates the network at
// all S + N_importance depths (NeRFRenderer.h:431,447), S of which ARE the coarse pass's sample points -- same o + d z, same hash features, bit for bit.  The
// feature table of a chunk therefore keeps the coarse pass's columns [0, n S), the hash encode of the fine pass runs on the N_importance NEW samples only
// (columns [n S, n (S + N_importance))), and the MLP reads column src[i] for point i (k_fine_depths emits the map while it merges the two sorted runs).
// A third of the fine pass's gathers (2.9 of 11.7 ms per 800x800 frame) is not issued; results are unchanged.
struct ReuseWs {
    __half2 *feats;      // [16][cols] level-major
    __half2 *feats_lo;   // HashEmbedder (fp32-valued features) in split precision: the plane of the rounding residuals, else null
    float2 *f32;         // HashEmbedder, sigma-only coarse pass: the fp32 features of the coarse columns [16][n S] (sigma_small_f32.hip reads these), else null
    uint8_t *keep;       // [cols]
    int32_t *src;        // [n, sf]
    float *z_new;        // [n, ni]
    int64_t cols;
};

static int reuse_layout(void *ws, size_t ws_bytes, int64_t n, int s, int ni, bool want_lo, bool want_f32, ReuseWs &w)
{
    Bump bump(ws, ws_bytes);
    w.cols = n * (int64_t)(s + ni);
    w.feats = bump.take<__half2>((size_t)w.cols * 16);
    w.feats_lo = want_lo ? bump.take<__half2>((size_t)w.cols * 16) : nullptr;
    w.f32 = want_f32 ? bump.take<float2>((size_t)n * s * 16) : nullptr;
    w.keep = bump.take<uint8_t>((size_t)w.cols);
    w.src = bump.take<int32_t>((size_t)w.cols);
    w.z_new = bump.take<float>((size_t)n * ni);
    if (bump.off > ws_bytes) { set_error("nrf_render_rays: workspace too small for the feature-reuse layout"); return NRF_ERR_WORKSPACE; }
    return NRF_OK;
}

// RunNetwork over p = n*s points given either explicit points or (rays, z).
static int run_network(const nrf_renderer *r, const PointSource &ps, const float *viewdirs, int vd_stride, int64_t n, int s, int prec,
                       float *raw, void *ws, size_t ws_bytes, hipStream_t st)
{
    const int64_t p = n * s;
    if (p == 0) return NRF_OK;
    if (ws_bytes < network_ws_bytes(r, p, prec)) { set_error("run_network: workspace %zu < %zu bytes", ws_bytes, network_ws_bytes(r, p, prec)); return NRF_ERR_WORKSPACE; }
    Bump bump(ws, ws_bytes);
    const int xd = r->in_ch + r->in_views;
    float *x = bump.take<float>((size_t)p * xd);
    uint8_t *keep = bump.take<uint8_t>((size_t)p);
    float *pts = bump.take<float>((size_t)p * 3);
    void *mws = bump.take<char>(0);
    const size_t mws_bytes = ws_bytes - bump.off;
    // embed_fn->forward(inputs_flat)                                            (NeRFRenderer.h:175)
    if (r->desc.hash) {
        NRF_TRY(launch_hash(r->desc.hash, ps, p, x, xd, keep, st));
        if (prec != NRF_PREC_F32 && r->desc.hash->desc.mode == NRF_HASH_NGP) {
            hipLaunchKernelGGL(k_zero_unkept_rows, dim3((unsigned)ceil_div(p * r->in_ch, 256)), dim3(256), 0, st, p, r->in_ch, xd, keep, x);
            NRF_LAUNCH_CHECK();
        }
    } else {
        const float *px = ps.pts;
        if (!px) {
            NRF_TRY(nrf_points(ps.rays, ps.ray_stride, ps.z, n, s, pts, st));
            px = pts;
        }
        NRF_TRY(launch_pe(px, 3, p, r->desc.pe_freqs, 1, x, xd, st));
    }
    // embeddirs_fn(view_dirs expanded per sample)                               (NeRFRenderer.h:177-183)
    if (r->in_views > 0) {
        if (!viewdirs) { set_error("run_network: the renderer was built with a direction encoder but no view directions were given"); return NRF_ERR_INVALID_ARG; }
        if (r->desc.dirs_encoder == NRF_DIRS_PE) NRF_TRY(launch_pe(viewdirs, vd_stride, p, r->desc.dirs_param, s, x + r->in_ch, xd, st));
        else NRF_TRY(launch_sh(viewdirs, vd_stride, p, r->desc.dirs_param, r->desc.dirs_encoder == NRF_DIRS_SH_CUDA ? NRF_SH_CUDA : NRF_SH_LIBTORCH, s, x + r->in_ch, xd, st));
    }
    // fn->forward(embedded)                                                     (NeRFRenderer.h:184)
    NRF_TRY(mlp_forward(r->desc.mlp, x, xd, p, prec, raw, r->desc.mlp->out_dims, mws, mws_bytes, st));
    // outputs_flat[~keep_mask, -1] = 0                                          (NeRFRenderer.h:187-188)
    if (r->desc.hash) {
        hipLaunchKernelGGL(k_mask_sigma, dim3((unsigned)ceil_div(p, 256)), dim3(256), 0, st, p, r->desc.mlp->out_dims, keep, raw);
        NRF_LAUNCH_CHECK();
    }
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_version(void) { return 100; }
const char *nrf_last_error(void) { return g_err; }

const char *nrf_status_string(int status)
{
    switch (status) {
        case NRF_OK: return "ok";
        case NRF_ERR_INVALID_ARG: return "invalid argument";
        case NRF_ERR_HIP: return "HIP runtime error";
        case NRF_ERR_UNSUPPORTED: return "unsupported configuration";
        case NRF_ERR_WORKSPACE: return "workspace too small";
        case NRF_ERR_NONFINITE: return "non-finite network outputs in a matrix-core precision";
        default: return "unknown status";
    }
}

int nrf_profile_enable(int on)
{
    g_prof_on.store(on != 0, std::memory_order_relaxed);
    return NRF_OK;
}

int nrf_profile_is_enabled(void) { return g_prof_on.load(std::memory_order_relaxed) ? 1 : 0; }

int nrf_profile_read(double *ms, int64_t *launches, int reset)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &rec : g_prof_pending) {
        float t = 0.0f;
        if (hipEventSynchronize(rec.e1) == hipSuccess && hipEventElapsedTime(&t, rec.e0, rec.e1) == hipSuccess) {
            g_prof_ms[rec.slot] += (double)t;
            g_prof_n[rec.slot] += 1;
        }
        g_prof_pool.push_back(rec.e0);
        g_prof_pool.push_back(rec.e1);
    }
    g_prof_pending.clear();
    for (int i = 0; i < NRF_PROF_COUNT; i++) {
        if (ms) ms[i] = g_prof_ms[i];
        if (launches) launches[i] = g_prof_n[i];
        if (reset) { g_prof_ms[i] = 0.0; g_prof_n[i] = 0; }
    }
    return NRF_OK;
}

int nrf_renderer_create(const nrf_renderer_desc *desc, nrf_renderer **out)
{
    NRF_CHECK_ARG(desc && out && desc->mlp, "nrf_renderer_create: null pointer");
    nrf_renderer *r = new nrf_renderer();
    r->desc = *desc;
    r->in_ch = desc->hash ? nrf_hash_output_dims(desc->hash) : 3 + 6 * desc->pe_freqs;
    switch (desc->dirs_encoder) {
        case NRF_DIRS_NONE: r->in_views = 0; break;
        case NRF_DIRS_PE: r->in_views = 3 + 6 * desc->dirs_param; break;
        case NRF_DIRS_SH_LIBTORCH:
        case NRF_DIRS_SH_CUDA: r->in_views = desc->dirs_param * desc->dirs_param; break;
        default: delete r; set_error("nrf_renderer_create: unknown direction encoder %d", desc->dirs_encoder); return NRF_ERR_INVALID_ARG;
    }
    const int max_deg = desc->dirs_encoder == NRF_DIRS_SH_LIBTORCH ? 5 : 8;
    if ((desc->dirs_encoder == NRF_DIRS_SH_LIBTORCH || desc->dirs_encoder == NRF_DIRS_SH_CUDA) && (desc->dirs_param < 1 || desc->dirs_param > max_deg)) {
        delete r; set_error("nrf_renderer_create: SH degree %d outside [1,%d]", desc->dirs_param, max_deg); return NRF_ERR_INVALID_ARG;
    }
    if (!desc->hash && (desc->pe_freqs < 1 || desc->pe_freqs > 32)) { delete r; set_error("nrf_renderer_create: pe_freqs %d outside [1,32]", desc->pe_freqs); return NRF_ERR_INVALID_ARG; }
    if (r->in_ch + r->in_views != desc->mlp->in_dims) {
        set_error("nrf_renderer_create: encoders produce %d + %d features but the MLP expects %d", r->in_ch, r->in_views, desc->mlp->in_dims);
        delete r;
        return NRF_ERR_INVALID_ARG;
    }
    if (desc->mlp->out_dims < 4) { delete r; set_error("nrf_renderer_create: the MLP must output at least rgb + sigma"); return NRF_ERR_INVALID_ARG; }
    *out = r;
    return NRF_OK;
}

void nrf_renderer_destroy(nrf_renderer *r) { delete r; }

size_t nrf_run_network_workspace_bytes(const nrf_renderer *r, int64_t n, int s)
{
    return r ? network_ws_bytes(r, n * s, NRF_PREC_F32) : 0;
}

int nrf_run_network(const nrf_renderer *r, const float *d_pts, const float *d_viewdirs, int64_t n, int s, int precision,
                    float *d_raw, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && d_pts && d_raw && n >= 0 && s >= 1, "nrf_run_network: bad argument");
    PointSource ps{d_pts, nullptr, nullptr, 0, s};
    return run_network(r, ps, d_viewdirs, 3, n, s, precision, d_raw, d_workspace, workspace_bytes, as_stream(stream));   // generic boundary: row-major path
}

size_t nrf_render_rays_workspace_bytes(const nrf_renderer *r, int64_t n, const nrf_render_params *p)
{
    if (!r || !p) return 0;
    const int s = p->n_samples, sf = p->n_samples + p->n_importance;
    const int c = r->desc.mlp->out_dims;
    size_t b = 0;
    b += align_up((size_t)n * s * 4, 256) * 2;            // z_coarse, weights_coarse
    b += align_up((size_t)n * s * c * 4, 256);            // raw_coarse
    b += align_up((size_t)n * sf * 4, 256);               // z_fine
    b += align_up((size_t)n * sf * c * 4, 256);           // raw_fine
    // the network's scratch: the generic row-major path of all S + N_importance depths, or the feature-reuse layout of the hash fast paths (reuse_layout), whichever is
    // larger -- with few importance samples the HashEmbedder layout (two feature planes of every column + the fp32 plane of the S coarse ones) passes the first
    const size_t reuse = align_up((size_t)n * sf * 16 * sizeof(__half2), 256) * 2 + align_up((size_t)n * s * 16 * sizeof(float2), 256) + align_up((size_t)n * sf, 256) +
                         align_up((size_t)n * sf * 4, 256) + align_up((size_t)n * (sf - s) * 4, 256) + 2048;
    const size_t generic = network_ws_bytes(r, n * sf, p->precision);
    b += (generic > reuse ? generic : reuse) + 4096;
    b += align_up((size_t)n * 64 * sizeof(__half), 256) * 2;  // per-ray direction features of the fast path (hi, lo)
    b += align_up((size_t)n * sf * 4, 256) + align_up((size_t)n * (sf - s) * 4, 256) + 1024;   // feature reuse: merge map + new-sample depths
    b += align_up((size_t)n * sf * 4, 256) + align_up((size_t)n * (sf - s) * 4, 256) + align_up((size_t)n * (sf - s) * 16, 256);   // raw reuse: map, depths, outputs of the new samples
    b += align_up((size_t)n * s * 64, 256) + align_up((size_t)n * sf * 16, 256);                 // geo hand-over: operand fragments of the coarse columns, outputs by column
    if (p->perturb > 0.0f) b += align_up((size_t)n * s * 4, 256);                             // un-jittered depths
    if (p->has_cone || p->precond_alpha > 0.0f) b += align_up((size_t)n * sf * 12, 256);     // explicit sample points
    return b;
}

}  // extern "C"

namespace nrf {

static int flag_buffers(const nrf_renderer *r)
{
    if (r->d_flags) return NRF_OK;
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&r->d_flags), NRF_FLAG_SLOTS * sizeof(uint32_t)));
    NRF_HIP(hipHostMalloc(reinterpret_cast<void **>(&r->h_flags), NRF_FLAG_SLOTS * sizeof(uint32_t), hipHostMallocDefault));
    NRF_HIP(hipHostMalloc(reinterpret_cast<void **>(&r->h_deferred), (size_t)NRF_DEFERRED_RING * NRF_FLAG_SLOTS * sizeof(uint32_t), hipHostMallocDefault));
    for (auto &e : r->deferred_ev) NRF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return NRF_OK;
}

// A pending NRF_OVERFLOW_DEFERRED copy: looked at when it has landed (wait = false: only then), reported ONCE
// (oldest first; only_entry >= 0: that ring entry alone, which the caller is about to reuse)
static int take_deferred(const nrf_renderer *r, bool wait, const char *who, int only_entry = -1)
{
    int bad = 0;
    for (int k = 0; k < NRF_DEFERRED_RING; k++) {
        const int e = only_entry >= 0 ? only_entry : (r->deferred_head + k) % NRF_DEFERRED_RING;      // head is the oldest entry once the ring has wrapped
        if (r->deferred_slots[e] > 0) {
            if (wait) NRF_HIP(hipEventSynchronize(r->deferred_ev[e]));
            else if (hipEventQuery(r->deferred_ev[e]) != hipSuccess) break;                            // later entries were recorded later
            for (int i = 0; i < r->deferred_slots[e]; i++) bad += r->h_deferred[(size_t)e * NRF_FLAG_SLOTS + i] != 0;
            r->deferred_slots[e] = 0;
        }
        if (only_entry >= 0) break;
    }
    if (!bad) return NRF_OK;
    r->flagged_chunks += bad;
    set_error("%s: an EARLIER render call on this renderer (NRF_OVERFLOW_DEFERRED) produced non-finite network outputs in %d chunk(s): an fp16 operand of the matrix-core "
              "precision left its range; render with NRF_OVERFLOW_RERENDER or NRF_PREC_F32", who, bad);
    return NRF_ERR_NONFINITE;
}

static inline int policy_of(const nrf_render_params *p) { return p->overflow_policy == NRF_OVERFLOW_AUTO ? NRF_OVERFLOW_RERENDER : p->overflow_policy; }
static inline bool detects(const nrf_render_params *p) { return p->precision != NRF_PREC_F32 && policy_of(p) != NRF_OVERFLOW_IGNORE; }

static int render_rays_impl(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p,
                            const float *d_t, const float *d_u, const nrf_render_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream, uint32_t *d_flag);

// After `slots` chunk words were written on `st`: what the policy says.  chunk_of(i, &first, &count) names the rays of chunk i.
template <class ChunkOf, class Rerender>
static int settle_flags(const nrf_renderer *r, const nrf_render_params *p, int slots, hipStream_t st, const char *who, ChunkOf chunk_of, Rerender rerender)
{
    const int pol = policy_of(p);
    if (pol == NRF_OVERFLOW_DEFERRED) {
        const int e = r->deferred_head;
        if (r->deferred_slots[e]) NRF_TRY(take_deferred(r, true, who, e));      // the ring is full: the copy of NRF_DEFERRED_RING calls ago (the host waits only when it runs that far ahead)
        NRF_HIP(hipMemcpyAsync(r->h_deferred + (size_t)e * NRF_FLAG_SLOTS, r->d_flags, (size_t)slots * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        NRF_HIP(hipEventRecord(r->deferred_ev[e], st));
        r->deferred_slots[e] = slots;
        r->deferred_head = (e + 1) % NRF_DEFERRED_RING;
        return NRF_OK;
    }
    NRF_HIP(hipMemcpyAsync(r->h_flags, r->d_flags, (size_t)slots * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NRF_HIP(hipStreamSynchronize(st));
    int bad = 0;
    for (int i = 0; i < slots; i++) bad += r->h_flags[i] != 0;
    if (!bad) return NRF_OK;
    r->flagged_chunks += bad;
    if (pol == NRF_OVERFLOW_ERROR) {
        set_error("%s: non-finite network outputs in %d of %d chunk(s): an fp16 operand of the matrix-core precision left its range (weights / features of unusual magnitude); "
                  "render with NRF_OVERFLOW_RERENDER or NRF_PREC_F32", who, bad, slots);
        return NRF_ERR_NONFINITE;
    }
    (void)chunk_of;
    for (int i = 0; i < slots; i++) if (r->h_flags[i]) { NRF_TRY(rerender(i)); }
    return NRF_OK;
}

// one chunk again in NRF_PREC_F32 into the same outputs (workspace: the caller's when it is large enough, else stream-ordered scratch)
static int rerender_f32(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p, const float *d_t, const float *d_u,
                        const nrf_render_outputs *out, void *d_workspace, size_t workspace_bytes, hipStream_t st)
{
    nrf_render_params q = *p;
    q.precision = NRF_PREC_F32;
    const size_t need = nrf_render_rays_workspace_bytes(r, n, &q);
    void *ws = d_workspace; size_t wsb = workspace_bytes; void *tmp = nullptr;
    if (need > workspace_bytes) { NRF_HIP(scratch_take(&tmp, need, st)); ws = tmp; wsb = need; }
    const int rc = render_rays_impl(r, d_rays, ray_stride, n, &q, d_t, d_u, out, ws, wsb, st, nullptr);
    if (tmp) (void)scratch_give(tmp, st);
    if (rc == NRF_OK) r->rerendered_chunks++;
    return rc;
}

}  // namespace nrf

extern "C" {

int nrf_renderer_nonfinite(const nrf_renderer *r, int64_t *flagged_chunks, int64_t *rerendered_chunks)
{
    NRF_CHECK_ARG(r, "nrf_renderer_nonfinite: null pointer");
    const int rc = take_deferred(r, true, "nrf_renderer_nonfinite");
    if (flagged_chunks) *flagged_chunks = r->flagged_chunks;
    if (rerendered_chunks) *rerendered_chunks = r->rerendered_chunks;
    return rc == NRF_ERR_NONFINITE ? NRF_OK : rc;          // the counters ARE the report here
}

int nrf_render_rays(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p,
                    const float *d_t, const float *d_u, const nrf_render_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && p && out, "nrf_render_rays: null pointer");
    NRF_CHECK_ARG(p->overflow_policy >= NRF_OVERFLOW_AUTO && p->overflow_policy <= NRF_OVERFLOW_IGNORE, "nrf_render_rays: overflow_policy %d is not an NRF_OVERFLOW_* value", p->overflow_policy);
    if (n == 0 || !detects(p)) return render_rays_impl(r, d_rays, ray_stride, n, p, d_t, d_u, out, d_workspace, workspace_bytes, stream, nullptr);
    hipStream_t st = as_stream(stream);
    NRF_TRY(flag_buffers(r));
    NRF_TRY(take_deferred(r, false, "nrf_render_rays"));
    NRF_HIP(hipMemsetAsync(r->d_flags, 0, sizeof(uint32_t), st));
    NRF_TRY(render_rays_impl(r, d_rays, ray_stride, n, p, d_t, d_u, out, d_workspace, workspace_bytes, stream, r->d_flags));
    return settle_flags(r, p, 1, st, "nrf_render_rays", [](int) {}, [&](int) { return rerender_f32(r, d_rays, ray_stride, n, p, d_t, d_u, out, d_workspace, workspace_bytes, st); });
}

}  // extern "C"

namespace nrf {

static int render_rays_impl(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p,
                            const float *d_t, const float *d_u, const nrf_render_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream, uint32_t *d_flag)
{
    r->last_view.valid = false;          // (set again below by the feature-reusing fast path)
    r->chunk_serial++;
    NRF_CHECK_ARG(r && p && out, "nrf_render_rays: null pointer");
    if (n == 0) return NRF_OK;
    NRF_CHECK_ARG(d_rays && d_t, "nrf_render_rays: null pointer");
    NRF_CHECK_ARG(n >= 0 && (ray_stride == 8 || ray_stride == 11), "nrf_render_rays: ray_stride must be 8 or 11 (NeRFRenderer.h:580-583)");
    NRF_CHECK_ARG(p->n_samples >= 1 && p->n_importance >= 0, "nrf_render_rays: bad sample counts");
    NRF_CHECK_ARG(p->n_importance == 0 || d_u || p->perturb > 0.0f, "nrf_render_rays: n_importance > 0 needs the u table");
    NRF_CHECK_ARG(r->in_views == 0 || ray_stride == 11, "nrf_render_rays: the renderer encodes view directions but the ray batch has none");
    NRF_CHECK_ARG(p->perturb >= 0.0f && p->raw_noise_std >= 0.0f && p->precond_alpha >= 0.0f, "nrf_render_rays: negative perturb / raw_noise_std / precond_alpha");
    NRF_CHECK_ARG(p->precond_alpha == 0.0f || p->has_bbox, "nrf_render_rays: stochastic preconditioning reflects at the bounding box (NeRFRenderer.h:436-442): bbox required");
    NRF_CHECK_ARG(p->perturb == 0.0f || p->n_samples >= 2, "nrf_render_rays: Perturb > 0 needs n_samples >= 2");
    NRF_CHECK_ARG(p->coarse_mode >= NRF_COARSE_AUTO && p->coarse_mode <= NRF_COARSE_SIGMA_F32, "nrf_render_rays: coarse_mode %d is not an NRF_COARSE_* value", p->coarse_mode);
    if (workspace_bytes < nrf_render_rays_workspace_bytes(r, n, p)) {
        set_error("nrf_render_rays: workspace %zu < %zu bytes", workspace_bytes, nrf_render_rays_workspace_bytes(r, n, p));
        return NRF_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    if (p->precision == NRF_PREC_F16_SPLIT) NRF_TRY(ensure_scales(r, st));          // (a no-op inside a Chunk loop: nrf_batchify_rays did it before its lanes forked)
    const int s = p->n_samples, ni = p->n_importance, sf = s + ni;
    const int c = r->desc.mlp->out_dims;
    // the FINAL compositing of the matrix-core precisions uses hardware exp / log / rcp and fp32 scans (composite.hip); NRF_PREC_F32, and the coarse pass whose
    // weights feed SamplePDF (a moved CDF bin is a visibly different sample set), keep the oracle's arithmetic
    const bool fastc = p->precision != NRF_PREC_F32;
    Bump bump(d_workspace, workspace_bytes);
    float *z_c = out->d_z_coarse ? out->d_z_coarse : bump.take<float>((size_t)n * s);
    float *w_c = out->d_weights_coarse ? out->d_weights_coarse : bump.take<float>((size_t)n * s);
    float *raw_c = out->d_raw_coarse ? out->d_raw_coarse : ((ni == 0 && out->d_raw) ? out->d_raw : bump.take<float>((size_t)n * s * c));
    float *z_f = nullptr, *raw_f = nullptr;
    if (ni > 0) {
        z_f = out->d_z_fine ? out->d_z_fine : bump.take<float>((size_t)n * sf);
        raw_f = out->d_raw ? out->d_raw : bump.take<float>((size_t)n * sf * c);
    }
    const bool fast = fast_path(r, p->precision);
    // the coarse pass only supplies SamplePDF's weights: sigma net alone, in the parity arithmetic (see nrf_render_params.coarse_mode)
    const bool sigma_only_hash = ni > 0 && fast && !out->d_raw_coarse && mlp_small_sigma_f32_available(r->desc.mlp) &&
                                 (p->coarse_mode == NRF_COARSE_SIGMA_F32 || (p->coarse_mode == NRF_COARSE_AUTO && p->precision == NRF_PREC_F16_SPLIT));
    // classic NeRF fast path: PE(10) positions / PE(4) directions + the 8x256 matrix-core kernel with the PE fused in
    const bool classic_split = p->precision == NRF_PREC_F16_SPLIT;
    const bool fast_classic = (p->precision == NRF_PREC_F16_MFMA || classic_split) && !r->desc.hash && r->desc.pe_freqs == 10 && r->desc.dirs_encoder == NRF_DIRS_PE &&
                              r->desc.dirs_param == 4 && (classic_split ? mlp_nerf_split_available(r->desc.mlp) : mlp_nerf_mfma_available(r->desc.mlp));
    // ... and its coarse pass: the density branch (eight 256-wide layers + alpha_linear) in exact fp32 on the matrix cores, followed in the same kernel by the colour
    // branch on the exact h8 in split precision (sigma_nerf_f32.hip).  Sigma -- hence the coarse weights and the fine sample set -- equals NRF_PREC_F32's bit for bit, and
    // the coarse pass still leaves whole (rgb, sigma) rows for the fine pass to reuse at its S coarse depths.  2 x the time of NRF_COARSE_FULL.
    const bool exact_classic = ni > 0 && fast_classic && mlp_nerf_sigma_f32_available(r->desc.mlp) && c == 4 &&
                               (p->coarse_mode == NRF_COARSE_SIGMA_F32 || (p->coarse_mode == NRF_COARSE_AUTO && classic_split));
    const bool sigma_only = sigma_only_hash;
    __half *dirs16 = nullptr;
    __half *dirs_lo = nullptr;
    if (fast) dirs16 = bump.take<__half>((size_t)n * r->in_views);
    if (fast && p->precision == NRF_PREC_F16_SPLIT) dirs_lo = bump.take<__half>((size_t)n * r->in_views);
    if (fast_classic) dirs16 = bump.take<__half>((size_t)n * 32);
    if (fast_classic && classic_split) dirs_lo = bump.take<__half>((size_t)n * 32);
    // merge map, new-sample depths and their network outputs of the raw-reuse fine pass (see reuse_raw below)
    int32_t *rr_src = ni > 0 ? bump.take<int32_t>((size_t)n * sf) : nullptr;
    float *rr_znew = ni > 0 ? bump.take<float>((size_t)n * ni) : nullptr;
    float *rr_rawnew = ni > 0 ? bump.take<float>((size_t)n * ni * 4) : nullptr;
    void *geo_planes = ni > 0 ? static_cast<void *>(bump.take<char>((size_t)n * s * 64)) : nullptr;      // see geo_reuse below
    float *raw_cols = ni > 0 ? bump.take<float>((size_t)n * sf * 4) : nullptr;
    float *z_plain = p->perturb > 0.0f ? bump.take<float>((size_t)n * s) : nullptr;
    float *bump_pts = (p->has_cone || p->precond_alpha > 0.0f) ? bump.take<float>((size_t)n * sf * 3) : nullptr;
    void *nws = bump.take<char>(0);
    const size_t nws_bytes = workspace_bytes - bump.off;
    const float *viewdirs = r->in_views > 0 ? d_rays + 8 : nullptr;
    if (fast) NRF_TRY(launch_dirs_f16(d_rays, ray_stride, n, r->desc.dirs_param, r->desc.dirs_encoder == NRF_DIRS_SH_CUDA ? NRF_SH_CUDA : NRF_SH_LIBTORCH, dirs16, dirs_lo, st));
    if (fast_classic && classic_split) NRF_TRY(launch_dirs_pe_split(d_rays, ray_stride, n, dirs16, dirs_lo, st));
    else if (fast_classic) NRF_TRY(launch_dirs_pe_f16(d_rays, ray_stride, n, dirs16, st));
    auto network = [&](const PointSource &src, int ns_, float *raw_out) -> int {
        if (fast) return run_network_fast(r, src, dirs16, dirs_lo, n, ns_, raw_out, nws, nws_bytes, st);
        if (fast_classic && classic_split) return mlp_nerf_forward_split_fused(r->desc.mlp, src.pts, src.rays, src.ray_stride, src.z, ns_, dirs16, dirs_lo, n * (int64_t)ns_, raw_out, st);
        if (fast_classic) return mlp_nerf_forward_mfma_fused(r->desc.mlp, src.pts, src.rays, src.ray_stride, src.z, ns_, dirs16, n * (int64_t)ns_, raw_out, st);
        return run_network(r, src, viewdirs, ray_stride, n, ns_, p->precision, raw_out, nws, nws_bytes, st);
    };
    // ---- stochastic branches: all off on the render path ----
    const RngRef rng{p->seed, p->ray_base};
    const bool jitter = p->perturb > 0.0f, cone = p->has_cone != 0, precond = p->precond_alpha > 0.0f;
    float *pts = (cone || precond) ? bump_pts : nullptr;
    SigmaNoise nz{};
    nz.on = p->raw_noise_std > 0.0f; nz.std = p->raw_noise_std; nz.g = rng;
    StochPoints sp{};
    sp.cone = cone; sp.cone_angle = p->cone_angle; sp.clamp = cone && p->has_bbox; sp.alpha = p->precond_alpha;
    for (int a = 0; a < 3; a++) { sp.box.mn[a] = p->bbox[a]; sp.box.mx[a] = p->bbox[3 + a]; }

    // z_vals; pts = o + d*z formed inside the encoder                           (NeRFRenderer.h:393-419)
    if (jitter) {
        NRF_TRY(nrf_z_vals(d_rays, ray_stride, n, d_t, s, p->lindisp, z_plain, st));
        NRF_TRY(launch_jitter_z(z_plain, nullptr, rng, n, s, z_c, st));                                             // :404-417
    } else NRF_TRY(nrf_z_vals(d_rays, ray_stride, n, d_t, s, p->lindisp, z_c, st));
    PointSource ps{nullptr, d_rays, z_c, ray_stride, s};
    if (cone) {                                                                                                    // :420
        sp.precond = 0; sp.stream_r = NRF_RNG_R_COARSE; sp.stream_theta = NRF_RNG_THETA_COARSE;
        NRF_TRY(launch_stoch_points(nullptr, d_rays, ray_stride, z_c, n, s, sp, rng, pts, st));
        ps.pts = pts;
    }
    // Where the coarse pass runs the WHOLE network in the fine pass's own arithmetic (the classic NeRF fast path in either matrix-core precision; HashNeRF with
    // NRF_COARSE_FULL), the fine pass's S coarse depths need no evaluation at all: their outputs exist.  The network then runs on the N_importance new samples and
    // k_gather_raw assembles raw_f through the merge map: a quarter of the frame's network evaluations (64 of 256 per ray) is not done; same kernel on the same
    // inputs, so the result is unchanged bit for bit.
    const bool reuse_raw = (fast || fast_classic) && ni > 0 && !sigma_only && !cone && !precond && c == 4 && n * (int64_t)sf < ((int64_t)1 << 31);
    // otherwise (the default split mode: coarse pass = sigma net alone) the coarse hash features are kept for the fine pass (see ReuseWs)
    const bool reuse = !reuse_raw && fast && ni > 0 && !cone && !precond && n * (int64_t)sf < ((int64_t)1 << 31);
    const bool ngp = fast && r->desc.hash->desc.mode == NRF_HASH_NGP;       // HashEmbedder: fp32-valued features, (hi, lo) planes in split precision
    // The default HashNeRF mode (split precision, exact sigma-only coarse pass): the coarse kernel also leaves the sigma net's whole output (sigma, geo_feat) as the colour
    // net's operand fragment, so the fine pass runs the colour net alone at its S coarse depths -- 44 of the 116 matrix instructions per 32 points and 144 of the 336
    // conversions per point are not repeated there, and what it uses is the EXACT sigma-net output instead of a split-precision repeat.  Both network launches of the
    // fine pass then walk feature COLUMNS in order (coarse columns, new columns: coalesced loads, no merge-map indirection) and write by column; the compositing
    // kernel reads through the merge map (one ray's samples lie in two contiguous runs).
    const bool geo_reuse = reuse && sigma_only && dirs_lo != nullptr && r->desc.mlp->small.geo_feat_dim <= 15;
    ReuseWs rw{};
    if (reuse) {
        NRF_TRY(reuse_layout(nws, nws_bytes, n, s, ni, ngp && dirs_lo, ngp && sigma_only, rw));
        // HashEmbedder mode: the exact coarse kernel reads the unrounded fp32 features; the (hi, lo) planes of the coarse columns are read only by a fine pass that
        // runs the whole network there -- not when the coarse kernel hands (sigma, geo_feat) over
        if (ngp && geo_reuse) NRF_TRY(launch_hash_ngp_lm(r->desc.hash, ps, n * (int64_t)s, reinterpret_cast<__half2 *>(rw.f32), n * (int64_t)s, 0, rw.keep, st, true, nullptr, 0));
        else if (ngp) NRF_TRY(launch_hash_ngp_lm(r->desc.hash, ps, n * (int64_t)s, rw.feats, rw.cols, rw.feats_lo ? rw.feats_lo - rw.feats : 0, rw.keep, st, false, rw.f32, n * (int64_t)s));
        else NRF_TRY(launch_hash_lm(r->desc.hash, ps, n * (int64_t)s, rw.feats, rw.cols, rw.keep, HASH_LM_DEFAULT_VARIANT, st));
        if (sigma_only) NRF_TRY(mlp_small_sigma_f32_lm(r->desc.mlp, ngp ? static_cast<const void *>(rw.f32) : rw.feats, ngp ? 1 : 0, ngp ? n * (int64_t)s : rw.cols, rw.keep, n * (int64_t)s, raw_c, st,
                                                       geo_reuse ? geo_planes : nullptr, n * (int64_t)s));
        else NRF_TRY(mlp_small_forward_mfma_lm(r->desc.mlp, rw.feats, rw.feats_lo, rw.cols, dirs16, dirs_lo, s, rw.keep, n * (int64_t)s, raw_c, st));
    } else if (exact_classic) NRF_TRY(mlp_nerf_exact_coarse(r->desc.mlp, ps.pts, ps.rays, ps.ray_stride, ps.z, s, dirs16, dirs_lo, n * (int64_t)s, raw_c, st));
    else if (sigma_only) NRF_TRY(run_sigma_fast(r, ps, n, s, raw_c, nws, nws_bytes, st));                         // raw_c holds sigma [n,s] only
    else NRF_TRY(network(ps, s, raw_c));                                                                           // :422
    nz.stream = NRF_RNG_NOISE_COARSE;
    if (ni == 0) {
        // a caller that asked for both the coarse intermediates and Raw gets the same rows in both (raw_c is the coarse buffer then: Raw was left unwritten -- a training
        // step with N_importance = 0 differentiated garbage; found by tools/scratch/train_fuzz.py)
        if (out->d_raw && raw_c != out->d_raw) NRF_HIP(hipMemcpyAsync(out->d_raw, raw_c, (size_t)n * s * c * sizeof(float), hipMemcpyDeviceToDevice, st));
        // the reference leaves result.Outputs UNDEFINED in this case (:423 vs :448); the coarse maps are what a caller wants
        return launch_raw2outputs(raw_c, z_c, d_rays + 3, ray_stride, n, s, c, 3, p->white_bkgr, out->d_rgb, out->d_disp, out->d_acc,
                                  out->d_weights ? out->d_weights : w_c, out->d_depth, nz, st, fastc, nullptr, nullptr, 0, d_flag);
    }
    NRF_TRY(launch_raw2outputs(raw_c, z_c, d_rays + 3, ray_stride, n, s, sigma_only ? 1 : c, sigma_only ? 0 : 3, p->white_bkgr, nullptr, nullptr, nullptr, w_c, nullptr, nz,
                               st, false));   // :423  always the exact arithmetic: these weights choose the fine samples
    NRF_TRY(launch_fine_depths(z_c, w_c, n, s, jitter ? nullptr : d_u, 0, rng, ni, p->sum_vec, z_f, st, reuse_raw ? rr_src : (reuse ? rw.src : nullptr),
                               reuse_raw ? rr_znew : (reuse ? rw.z_new : nullptr)));                                // :427-431 (det = perturb == 0)
    PointSource psf{nullptr, d_rays, z_f, ray_stride, sf};
    const float *raw_final = raw_f;
    const int32_t *src_final = nullptr;          // compositing reads sample i's network output at row src_final[i] (NULL: i) of raw_final | raw2_final (launch_raw2outputs)
    const float *raw2_final = nullptr;
    int64_t split_final = 0;
    if (cone || precond) {                                                                                         // :433-445
        sp.precond = precond; sp.stream_r = NRF_RNG_R_FINE; sp.stream_theta = NRF_RNG_THETA_FINE;
        NRF_TRY(launch_stoch_points(nullptr, d_rays, ray_stride, z_f, n, sf, sp, rng, pts, st));
        psf.pts = pts;
    }
    if (reuse_raw) {
        // the network on the N_importance new samples only; the S coarse depths take the coarse pass's outputs (same kernel, same inputs: same bits)
        PointSource psn{nullptr, d_rays, rr_znew, ray_stride, ni};
        NRF_TRY(network(psn, ni, rr_rawnew));
        if (out->d_raw) {          // the caller wants raw in depth order
            hipLaunchKernelGGL(k_gather_raw, dim3((unsigned)ceil_div(n * (int64_t)sf, 256)), dim3(256), 0, st, n * (int64_t)sf, rr_src, reinterpret_cast<const float4 *>(raw_c),
                               reinterpret_cast<const float4 *>(rr_rawnew), n * (int64_t)s, reinterpret_cast<float4 *>(raw_f));
            NRF_LAUNCH_CHECK();
        } else { raw_final = raw_c; raw2_final = rr_rawnew; split_final = n * (int64_t)s; src_final = rr_src; }          // the compositing kernel reads through the merge map
    } else if (reuse) {
        // the hash encode of the N_importance new samples only; the MLP gathers every depth's column through the merge map
        PointSource psn{nullptr, d_rays, rw.z_new, ray_stride, ni};
        if (ngp) NRF_TRY(launch_hash_ngp_lm(r->desc.hash, psn, n * (int64_t)ni, rw.feats + n * (int64_t)s, rw.cols, rw.feats_lo ? rw.feats_lo - rw.feats : 0, rw.keep + n * (int64_t)s, st));
        else NRF_TRY(launch_hash_lm(r->desc.hash, psn, n * (int64_t)ni, rw.feats + n * (int64_t)s, rw.cols, rw.keep + n * (int64_t)s, HASH_LM_DEFAULT_VARIANT, st));
        if (!ngp) { r->last_view.feats = rw.feats; r->last_view.cols = rw.cols; r->last_view.keep = rw.keep; r->last_view.src = rw.src; r->last_view.n = n; r->last_view.sf = sf; r->last_view.valid = true; }
        if (geo_reuse) {
            const int64_t nc = n * (int64_t)s;
            NRF_TRY(mlp_small_color_from_geo_lm(r->desc.mlp, geo_planes, nc, raw_c, dirs16, dirs_lo, s, rw.keep, nc, raw_cols, st));
            NRF_TRY(mlp_small_forward_mfma_lm(r->desc.mlp, rw.feats + nc, rw.feats_lo ? rw.feats_lo + nc : nullptr, rw.cols, dirs16, dirs_lo, ni, rw.keep + nc, n * (int64_t)ni,
                                              raw_cols + nc * 4, st));
            if (out->d_raw) {          // the caller wants raw in depth order
                hipLaunchKernelGGL(k_gather_raw, dim3((unsigned)ceil_div(n * (int64_t)sf, 256)), dim3(256), 0, st, n * (int64_t)sf, rw.src, reinterpret_cast<const float4 *>(raw_cols),
                                   reinterpret_cast<const float4 *>(raw_cols) + nc, nc, reinterpret_cast<float4 *>(raw_f));
                NRF_LAUNCH_CHECK();
            } else { raw_final = raw_cols; src_final = rw.src; }
        } else NRF_TRY(mlp_small_forward_mfma_lm(r->desc.mlp, rw.feats, rw.feats_lo, rw.cols, dirs16, dirs_lo, sf, rw.keep, n * (int64_t)sf, raw_f, st, rw.src));
    } else NRF_TRY(network(psf, sf, raw_f));                                                                       // :447
    nz.stream = NRF_RNG_NOISE_FINE;
    return launch_raw2outputs(raw_final, z_f, d_rays + 3, ray_stride, n, sf, c, 3, p->white_bkgr, out->d_rgb, out->d_disp, out->d_acc,
                              out->d_weights, out->d_depth, nz, st, fastc, src_final, raw2_final, split_final, d_flag);     // :448
}

}  // namespace nrf

extern "C" {


// ---- BatchifyRays (NeRFRenderer.h:465-525) and the pose branch of Render (:530-605) as single calls ----
static nrf_render_outputs slice_outputs(const nrf_render_outputs &o, int64_t i, int s, int so, int sf, int c)
{
    nrf_render_outputs q = o;
    if (q.d_rgb) q.d_rgb += i * 3;
    if (q.d_disp) q.d_disp += i;
    if (q.d_acc) q.d_acc += i;
    if (q.d_depth) q.d_depth += i;
    if (q.d_weights) q.d_weights += i * so;
    if (q.d_raw) q.d_raw += i * so * c;
    if (q.d_z_coarse) q.d_z_coarse += i * s;
    if (q.d_raw_coarse) q.d_raw_coarse += i * s * c;
    if (q.d_weights_coarse) q.d_weights_coarse += i * s;
    if (q.d_z_fine) q.d_z_fine += i * sf;
    return q;
}

// ---- the Chunk loop on lanes ----
// Consecutive chunks are independent, and their kernels are bound by different things: the hash encode's fine levels by gather latency, the network kernels by the
// matrix / vector issue of the SIMDs.  Issued on L streams, the chunks' kernels share the CUs (the NeRFSmall kernels hold 206-218 of a SIMD's 512 registers per wave,
// two waves per SIMD: a wave of the 44-66-register encode fits beside them).  Chunk i runs on the lane that has been given the fewest rays so far, in its own slice of
// the workspace; the lanes fork from the caller's stream and join it again, so the call is as asynchronous and as ordered as before, and the results do not depend on
// it (same kernels on the same slices).  A batch that is one chunk is cut into L.  NRF_RENDER_LANES=1 (or nrf_set_render_lanes(1)) restores the single-stream loop.
static std::atomic<int> g_render_lanes{0};          // 0: not decided yet (environment, default 2)
static int render_lanes()
{
    int v = g_render_lanes.load(std::memory_order_relaxed);
    if (v == 0) {
        const char *e = getenv("NRF_RENDER_LANES");
        v = e ? atoi(e) : 2;
        if (v < 1 || v > NRF_MAX_LANES) v = 2;
        g_render_lanes.store(v, std::memory_order_relaxed);
    }
    return v;
}
#ifndef NRF_LANE_STAGGER
#define NRF_LANE_STAGGER 1               // lane k's first chunk is (L - k) / L of a chunk: the lanes run out of phase (0: full chunks from the start)
#endif
#ifndef NRF_LANE_BALANCE
#define NRF_LANE_BALANCE 1               // the tail of the batch is cut so that all lanes end together (0: full chunks to the end)
#endif
constexpr int64_t LANE_MIN_RAYS = 32768;           // below this a batch stays on the caller's stream (a 16 384-ray training batch: 9.39 ms per step on one stream, 9.49 cut in two)

// rays per chunk of the L-lane loop; 0: single-stream loop
static int64_t lane_chunk(int64_t n, int chunk, int lanes)
{
    if (lanes < 2 || n < LANE_MIN_RAYS) return 0;
    if (n > chunk) return chunk;
    return ((n + lanes - 1) / lanes + 63) / 64 * 64;            // one chunk: L parts
}

static int lanes_for(const nrf_renderer *r) { return r->lanes > 0 ? r->lanes : render_lanes(); }

int nrf_renderer_set_lanes(nrf_renderer *r, int lanes)
{
    NRF_CHECK_ARG(r && lanes >= 0 && lanes <= NRF_MAX_LANES, "nrf_renderer_set_lanes: 0 (the process-wide default) .. %d lanes", NRF_MAX_LANES);
    r->lanes = lanes;
    return NRF_OK;
}

size_t nrf_batchify_rays_workspace_bytes(const nrf_renderer *r, int64_t n, int chunk, const nrf_render_params *p)
{
    if (!r || !p || chunk <= 0) return 0;
    const int lanes = lanes_for(r);
    const int64_t lc = lane_chunk(n, chunk, lanes);
    if (lc > 0 && lc < n) return (size_t)lanes * align_up(nrf_render_rays_workspace_bytes(r, lc, p), 256);
    return nrf_render_rays_workspace_bytes(r, n < chunk ? n : (int64_t)chunk, p);
}

int nrf_get_render_lanes(void) { return render_lanes(); }

int nrf_set_render_lanes(int lanes)
{
    NRF_CHECK_ARG(lanes >= 1 && lanes <= NRF_MAX_LANES, "nrf_set_render_lanes: 1 (single stream) .. %d", NRF_MAX_LANES);
    g_render_lanes.store(lanes, std::memory_order_relaxed);
    return NRF_OK;
}

static int lanes_of(const nrf_renderer *r, int lanes, hipStream_t *st, hipEvent_t *fork, hipEvent_t *done)
{
    std::lock_guard<std::mutex> lk(r->lane_mu);
    int dev = 0;
    NRF_HIP(hipGetDevice(&dev));
    if (r->lane_device >= 0 && r->lane_device != dev) {
        // the renderer is now used on another device: its lanes move with it (the old ones are drained and destroyed on their own device)
        int cur = dev;
        (void)hipSetDevice(r->lane_device);
        r->drop_lanes();
        NRF_HIP(hipSetDevice(cur));
    }
    for (int i = 0; i < lanes; i++) {
        if (!r->lane[i]) {
            // NRF_LANE_CU_MASK=1 (experiment, docs/history/profiles/round4/r4z_*): lane i of L on its own 256 / L compute units (hipExtStreamCreateWithCUMask; a contiguous bit range)
            static const int masked = [] { const char *e = getenv("NRF_LANE_CU_MASK"); return e ? atoi(e) : 0; }();
            if (masked && lanes > 1) {
                uint32_t bits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const int per = 256 / lanes;
                for (int b = i * per; b < (i + 1) * per; b++) bits[b >> 5] |= 1u << (b & 31);
                NRF_HIP(hipExtStreamCreateWithCUMask(&r->lane[i], 8, bits));
            } else NRF_HIP(hipStreamCreateWithFlags(&r->lane[i], hipStreamNonBlocking));
        }
        if (!r->lane_done[i]) NRF_HIP(hipEventCreateWithFlags(&r->lane_done[i], hipEventDisableTiming));
        st[i] = r->lane[i]; done[i] = r->lane_done[i];
    }
    if (!r->lane_fork) NRF_HIP(hipEventCreateWithFlags(&r->lane_fork, hipEventDisableTiming));
    *fork = r->lane_fork;
    r->lane_device = dev;
    return NRF_OK;
}

int nrf_batchify_rays(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, int chunk, const nrf_render_params *p, const float *d_t,
                      const float *d_u, const nrf_render_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && p && out, "nrf_batchify_rays: null pointer");
    NRF_CHECK_ARG(chunk > 0 && n >= 0, "nrf_batchify_rays: Chunk must be positive");
    NRF_CHECK_ARG(p->overflow_policy >= NRF_OVERFLOW_AUTO && p->overflow_policy <= NRF_OVERFLOW_IGNORE, "nrf_batchify_rays: overflow_policy %d is not an NRF_OVERFLOW_* value", p->overflow_policy);
    const int s = p->n_samples, sf = p->n_samples + p->n_importance, so = p->n_importance > 0 ? sf : s;
    const int c = r->desc.mlp->out_dims;
    nrf_render_params q = *p;
    // before any lane forks, on the caller's stream: the split image's range scales for the table as it is now, and the chunk words of this call cleared
    if (p->precision == NRF_PREC_F16_SPLIT) NRF_TRY(ensure_scales(r, as_stream(stream)));
    const bool det = n > 0 && detects(p);
    if (det) {
        NRF_TRY(flag_buffers(r));
        NRF_TRY(take_deferred(r, false, "nrf_batchify_rays"));
        NRF_HIP(hipMemsetAsync(r->d_flags, 0, NRF_FLAG_SLOTS * sizeof(uint32_t), as_stream(stream)));
    }
    struct ChunkRec { int64_t first, count; };
    std::vector<ChunkRec> done_chunks;
    auto flag_of = [&](size_t idx) -> uint32_t * { return det ? r->d_flags + (idx % NRF_FLAG_SLOTS) : nullptr; };
    // the policy, once every chunk has been issued and the lanes have joined `stream` (nrf_render_params.overflow_policy)
    auto settle = [&]() -> int {
        if (!det || done_chunks.empty()) return NRF_OK;
        const int slots = (int)(done_chunks.size() < (size_t)NRF_FLAG_SLOTS ? done_chunks.size() : (size_t)NRF_FLAG_SLOTS);
        return settle_flags(r, p, slots, as_stream(stream), "nrf_batchify_rays", [](int) {}, [&](int slot) -> int {
            for (size_t j = (size_t)slot; j < done_chunks.size(); j += NRF_FLAG_SLOTS) {
                nrf_render_params q2 = *p;
                q2.ray_base = p->ray_base + done_chunks[j].first;
                const nrf_render_outputs o2 = slice_outputs(*out, done_chunks[j].first, s, so, sf, c);
                NRF_TRY(rerender_f32(r, d_rays + done_chunks[j].first * ray_stride, ray_stride, done_chunks[j].count, &q2, d_t, d_u, &o2, d_workspace, workspace_bytes, as_stream(stream)));
            }
            return NRF_OK;
        });
    };
    const int L = lanes_for(r);
    const int64_t lc = lane_chunk(n, chunk, L);
    const size_t part = lc > 0 && lc < n ? align_up(nrf_render_rays_workspace_bytes(r, lc, p), 256) : 0;
    if (part > 0 && (size_t)L * part <= workspace_bytes && d_workspace) {
        // fork from the caller's stream, each chunk on the least-loaded lane in that lane's slice of the workspace, join
        hipStream_t st = as_stream(stream), lane[NRF_MAX_LANES];
        hipEvent_t fork = nullptr, done[NRF_MAX_LANES];
        NRF_TRY(lanes_of(r, L, lane, &fork, done));
        int rc = NRF_OK;
        if (hipEventRecord(fork, st) != hipSuccess) { set_error("nrf_batchify_rays: forking the lanes failed"); rc = NRF_ERR_HIP; }
        for (int j = 0; j < L && rc == NRF_OK; j++)
            if (hipStreamWaitEvent(lane[j], fork, 0) != hipSuccess) { set_error("nrf_batchify_rays: forking the lanes failed"); rc = NRF_ERR_HIP; }
        // Lane k starts with (L - k) / L of a chunk: the lanes then run out of phase (one in its encode while another is in its network) instead of doing the same
        // thing at the same time, which is what makes them share the CUs well.  The next chunk goes to the lane that has been given the fewest rays so far, and the
        // last < L chunks are cut so that all lanes end together.
        int64_t given[NRF_MAX_LANES] = {0, 0, 0, 0};
        bool first[NRF_MAX_LANES] = {true, true, true, true};
        for (int64_t i = 0; i < n && rc == NRF_OK;) {                                                             // :476
            int k = 0;
            for (int j = 1; j < L; j++) if (given[j] < given[k]) k = j;
            const int64_t rem = n - i;
            int64_t m = lc;
#if NRF_LANE_STAGGER
            if (first[k]) m = (lc * (L - k) / L + 63) / 64 * 64;
            if (m > lc) m = lc;                                                 // a Chunk that is no multiple of 64: the rounding must not pass the lane's slice of the workspace
#endif
            first[k] = false;
#if NRF_LANE_BALANCE
            if (rem < (int64_t)L * lc) {
                int64_t total = rem;
                for (int j = 0; j < L; j++) total += given[j];
                m = (total / L - given[k] + 63) / 64 * 64;                       // lane k's share of the rest that evens the lanes out
                if (m > lc) m = lc;
                if (m < lc / 8) m = lc / 8;                                     // a lane already past its share: no slivers (a chunk has ~12 launches whatever its size)
                if (rem - m < lc / 8 && rem <= lc) m = rem;                     // no crumbs
            }
#endif
            if (m < 1) m = rem < lc ? rem : lc;                                 // Chunk < 8 on the lane path: lc / 8 == 0 must not leave an empty chunk (the loop would never advance)
            if (m > rem) m = rem;
            q.ray_base = p->ray_base + i;
            const nrf_render_outputs o = slice_outputs(*out, i, s, so, sf, c);
            rc = render_rays_impl(r, d_rays + i * ray_stride, ray_stride, m, &q, d_t, d_u, &o, static_cast<char *>(d_workspace) + (size_t)k * part, part, lane[k], flag_of(done_chunks.size()));
            done_chunks.push_back({i, m});
            given[k] += m;
            i += m;
        }
        // join on every path: whatever was launched is ordered before the caller's next operation
        for (int j = 0; j < L; j++) {
            if (hipEventRecord(done[j], lane[j]) != hipSuccess || hipStreamWaitEvent(st, done[j], 0) != hipSuccess) {
                if (rc == NRF_OK) { set_error("nrf_batchify_rays: joining the lanes failed"); rc = NRF_ERR_HIP; }
                (void)hipStreamSynchronize(lane[j]);
            }
        }
        if (done_chunks.size() != 1) r->last_view.valid = false;          // the feature view describes ONE chunk's workspace: only a single-chunk call keeps it
        return rc == NRF_OK ? settle() : rc;
    }
    for (int64_t i = 0; i < n; i += chunk) {                                                                      // :476
        const int64_t m = n - i < chunk ? n - i : (int64_t)chunk;
        q.ray_base = p->ray_base + i;
        const nrf_render_outputs o = slice_outputs(*out, i, s, so, sf, c);
        NRF_TRY(render_rays_impl(r, d_rays + i * ray_stride, ray_stride, m, &q, d_t, d_u, &o, d_workspace, workspace_bytes, stream, flag_of(done_chunks.size())));
        done_chunks.push_back({i, m});
    }
    if (done_chunks.size() != 1) r->last_view.valid = false;
    return settle();
}

// The hash features the most recent render left in its workspace (a single-chunk render of the feature-reusing fast path, CuHashEmbedder grid): level-major fp16
// [16][cols] half2, the keep mask by column, the merge map [n, sf] (sample i of the sorted depths -> column).  Valid until the next render call on this renderer or
// any other use of that workspace.  NRF_ERR_UNSUPPORTED when the last call left none.
extern "C" NRF_API int nrf_renderer_last_features(const nrf_renderer *r, const void **d_feats_lm, int64_t *cols, const uint8_t **d_keep_cols, const int32_t **d_src, int64_t *n, int *sf,
                                                  uint64_t *serial)
{
    NRF_CHECK_ARG(r && d_feats_lm && cols && d_keep_cols && d_src && n && sf, "nrf_renderer_last_features: null pointer");
    if (serial) *serial = r->chunk_serial;
    if (!r->last_view.valid) { set_error("nrf_renderer_last_features: the last render call left no feature view (several chunks, another path, or none yet)"); return NRF_ERR_UNSUPPORTED; }
    *d_feats_lm = r->last_view.feats; *cols = r->last_view.cols; *d_keep_cols = r->last_view.keep; *d_src = r->last_view.src; *n = r->last_view.n; *sf = r->last_view.sf;
    return NRF_OK;
}

extern "C" int nrf_view_check(const nrf_view *v, const char *who);

size_t nrf_render_rows_workspace_bytes(const nrf_renderer *r, const nrf_view *v, const nrf_render_params *p)
{
    if (!r || !v || !p || v->chunk <= 0 || v->rows < 0 || v->w <= 0) return 0;
    const int64_t n = (int64_t)v->rows * v->w;
    return align_up((size_t)n * (v->use_viewdirs ? 11 : 8) * sizeof(float), 256) + 256 + nrf_batchify_rays_workspace_bytes(r, n, v->chunk, p);
}

int nrf_render_rows(const nrf_renderer *r, const nrf_view *v, const nrf_render_params *p, const float *d_t, const float *d_u, const nrf_render_outputs *out,
                    float *d_rays_out, float *d_near_far, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && p && out, "nrf_render_rows: null pointer");
    NRF_TRY(nrf_view_check(v, "nrf_render_rows"));
    // Ndc with cone rays (ThinRay = false): NDCRays multiplies cone_angle by |d_ndc| / |rays_d| AFTER rays_d has been replaced by d_ndc (RayUtils.h:73-81) -- the
    // quotient of a finite non-zero number by itself, exactly 1.0 -- so every ray keeps the camera's cone_angle bit for bit, as a [.., 1] tensor there, as the scalar
    // of p->cone_angle here
    const int64_t n = (int64_t)v->rows * v->w;
    if (n == 0) return nrf_view_rays(v, nullptr, d_near_far, stream);        // an empty tile: only Near / Far (= +inf / -inf) are defined
    const int stride = v->use_viewdirs ? 11 : 8;
    if (workspace_bytes < nrf_render_rows_workspace_bytes(r, v, p)) {
        set_error("nrf_render_rows: workspace %zu < %zu bytes", workspace_bytes, nrf_render_rows_workspace_bytes(r, v, p));
        return NRF_ERR_WORKSPACE;
    }
    Bump bump(d_workspace, workspace_bytes);
    float *rays = d_rays_out ? d_rays_out : bump.take<float>((size_t)n * stride);
    void *ws = bump.take<char>(0);
    NRF_TRY(nrf_view_rays(v, rays, d_near_far, stream));                                                          // :541-583, :602-603
    nrf_render_params q = *p;
    q.ray_base = p->ray_base + (int64_t)v->row0 * v->w;
    return nrf_batchify_rays(r, rays, stride, n, v->chunk, &q, d_t, d_u, out, ws, workspace_bytes - bump.off, stream);   // :586-590
}

}  // extern "C"
