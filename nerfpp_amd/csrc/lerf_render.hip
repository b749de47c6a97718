// lerf_render.hip -- the LeRF language-embedding render pass as library calls:
//
//   LeRFRenderer::Render       LeRFRenderer.cpp:265-330   pose -> rays -> BatchifyRays -> reshape           nrf_lerf_render_rows
//   LeRFRenderer::BatchifyRays LeRFRenderer.cpp:189-263   host loop over Chunk-sized slices + torch::cat    nrf_lerf_batchify_rays (slices written in place, lanes inside)
//   LeRFRenderer::RenderRays   LeRFRenderer.cpp:85-187    z_vals -> RunLENetwork -> RawToLEOutputs -> SamplePDF -> sort -> RunLENetwork -> RawToLEOutputs
//                                                                                                            nrf_lerf_render_rays
//   Relevancy                  LeRFRenderer.cpp:79        rendered embedding vs the prompt embeddings         nrf_lerf_relevancy   (PARITY UNPINNED, see below)
//   relevancy image            NeRFExecutor.h:713-719     rel[..., 0] * 255 -> u8 -> COLORMAP_JET             nrf_relevancy_image  (PARITY UNPINNED)
//
// One chunk on the device (split precision, the default; every sample point is encoded once and its density net evaluated once):
//   k_z_vals, k_points                         coarse depths and sample points
//   hash encode (CuHash L16 F8, level-major)   coarse points -> feature columns [0, 64 n)
//   sigma_le in EXACT fp32 (matrix cores)      -> sigma_le, (sigma, geo32) operand planes          (sigma_lerf_f32.hip; the fp32 stage path's fine sample set bit for bit)
//   raw2weights                                coarse weights                                      (RawToLEOutputs, weights only)
//   fine_depths_merge                          SamplePDF + stable merge -> z_fine, new depths, merge map
//   k_points, hash encode                      the 128 NEW samples -> columns [64 n, 192 n)
//   sigma net (split fp16) on the new columns  -> sigma_le, geo planes
//   raw2weights through the merge map          weights / depth / disp / acc in depth order
//   embedding passes (LE net from the geo planes, Gram norm, 256 -> 768 once per ray) + normalise   (RenderCLIPEmbedding, LeRFRenderer.h:45-54)
//   relevancy                                  when prompts are set                                (LeRFRenderer.cpp:79)
//
// `Relevancy` is defined in the external module DeliriumV01D/RuCLIP (RuCLIPProcessor.h; relative path ../RuCLIP/src, no pinned version, absent from the reference tree)
// and cv::applyColorMap is OpenCV's: both are restated from their published algorithms (oracle/nerf_oracle.c, orc_relevancy / orc_colormap_jet_lut, which say what exactly)
// and anchored on the reference's call sites; no reference run or fixture pins them.
#include "common.h"
#include "mlp.h"
#include "stoch.h"

#include <cstdlib>
#include <mutex>
#include <vector>

constexpr int NRF_LERF_MAX_LANES = 4;

struct nrf_lerf_renderer {
    nrf_lerf_renderer_desc desc;
    int embed_dim = 0;
    // prompt embeddings on the device (SetLeRFPrompts, LeRFRenderer.h:86): [P, E] positives, [Q, E] negatives
    float *d_pos = nullptr, *d_neg = nullptr;
    int n_pos = 0, n_neg = 0;
    // lanes of this renderer's Chunk loop.  ONE by default: the LeRF kernels gain nothing from sharing the CUs (800x800 frame, same call: 1 lane 140-143 ms, 2 lanes
    // 146-147 ms at Chunk 32768; docs/history/profiles/round4/r4g_lerf_lane_chunk_sweep.log) -- their sum is matrix-bound and the F = 8 encode is at the HBM roofline by itself
    int lanes = 1;
    // lanes of the Chunk loop (as nrf_renderer's): created on first use, bound to one device and one caller at a time
    mutable std::mutex lane_mu;
    mutable hipStream_t lane[NRF_LERF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    mutable hipEvent_t lane_fork = nullptr, lane_done[NRF_LERF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    mutable int lane_device = -1;
    // the pass's non-finite word (nrf_render_params.overflow_policy): ONE per call -- the pass has no fp32 single-call twin to render a flagged chunk again with, so a
    // flagged call is an error under every detecting policy; a pinned mirror and the event of a deferred copy
    mutable uint32_t *d_flag = nullptr, *h_flag = nullptr;
    mutable hipEvent_t flag_ev = nullptr;
    mutable bool flag_pending = false;
    mutable int64_t flagged_calls = 0;
    // where the last chunk left the language features of its fine depths (nrf_lerf_renderer_last_features; as nrf_renderer's last_view): the level-major fp16 table
    // [16][cols][8], the keep mask by column, the merge map [n, sf] -- in the caller's workspace; valid after a call that rendered exactly ONE chunk
    mutable struct { const void *feats = nullptr; int64_t cols = 0; const uint8_t *keep = nullptr; const int32_t *src = nullptr; int64_t n = 0; int sf = 0; bool valid = false; } last_view;
    mutable uint64_t chunk_serial = 0;
    void drop_lanes() const
    {
        for (auto &st : lane) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
        for (auto &e : lane_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (lane_fork) { (void)hipEventDestroy(lane_fork); lane_fork = nullptr; }
        lane_device = -1;
    }
    ~nrf_lerf_renderer()
    {
        drop_lanes();
        if (flag_ev) { (void)hipEventSynchronize(flag_ev); (void)hipEventDestroy(flag_ev); }
        if (d_flag) (void)hipFree(d_flag);
        if (h_flag) (void)hipHostFree(h_flag);
        if (d_pos) (void)hipFree(d_pos);
        if (d_neg) (void)hipFree(d_neg);
    }
};

namespace nrf {

struct LBump {
    char *base;
    size_t off = 0;
    explicit LBump(void *b) : base(static_cast<char *>(b)) {}
    template <class T> T *take(size_t count)
    {
        off = align_up(off, 256);
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
};

// ---------------------------------------------------------------------------------------------------
// Relevancy: one wave per ray.  logits = e . phrase (fp32, 768 terms: 12 per lane then a wave reduction), pairwise softmax at temperature 10 against each
// negative, the pair whose positive probability is smallest (first on ties, as torch::argmin).  The phrases (1 + Q rows of E floats) sit in LDS.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int REL_WAVES = 4;

__global__ void __launch_bounds__(64 * REL_WAVES) k_lerf_relevancy(const float *__restrict__ emb, int64_t n, int e, const float *__restrict__ pos, const float *__restrict__ neg,
                                                                  int q, float *__restrict__ out)
{
    extern __shared__ float ph[];          // [1 + q][e]
    for (int i = threadIdx.x; i < e; i += blockDim.x) ph[i] = pos[i];
    for (int i = threadIdx.x; i < q * e; i += blockDim.x) ph[e + i] = neg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t ray = (int64_t)blockIdx.x * REL_WAVES + wave; ray < n; ray += (int64_t)gridDim.x * REL_WAVES) {
        const float *x = emb + ray * (int64_t)e;
        float lp = 0.0f;
        for (int k = lane; k < e; k += 64) lp = __builtin_fmaf(x[k], ph[k], lp);
        lp = wave_sum(lp);
        float best0 = 0.0f, best1 = 0.0f;
        for (int j = 0; j < q; j++) {
            float ln = 0.0f;
            for (int k = lane; k < e; k += 64) ln = __builtin_fmaf(x[k], ph[(j + 1) * e + k], ln);
            ln = wave_sum(ln);
            const float a = 10.0f * lp, b = 10.0f * ln, m = fmaxf(a, b);
            const float ea = expf(a - m), eb = expf(b - m), sum = ea + eb;
            const float s0 = ea / sum, s1 = eb / sum;
            if (j == 0 || s0 < best0) { best0 = s0; best1 = s1; }
        }
        if (lane == 0) { out[ray * 2 + 0] = best0; out[ray * 2 + 1] = best1; }
    }
}

// rel[i * stride] * 255 -> u8 (truncation, saturated: torch's .mul(255).to(kU8) on values in [0, 1]) -> lut -> [n, 3] B, G, R
__global__ void k_relevancy_image(const float *__restrict__ rel, int64_t n, int stride, const uint8_t *__restrict__ lut, uint8_t *__restrict__ bgr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = rel[i * stride] * 255.0f;
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    const int b = (int)(uint8_t)v;
    bgr[i * 3 + 0] = lut[b * 3 + 0]; bgr[i * 3 + 1] = lut[b * 3 + 1]; bgr[i * 3 + 2] = lut[b * 3 + 2];
}

__global__ void k_lut_u8(const uint8_t *__restrict__ idx, int64_t n, const uint8_t *__restrict__ lut, uint8_t *__restrict__ bgr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = idx[i];
    bgr[i * 3 + 0] = lut[b * 3 + 0]; bgr[i * 3 + 1] = lut[b * 3 + 1]; bgr[i * 3 + 2] = lut[b * 3 + 2];
}

// OpenCV's COLORMAP_JET table restated (see oracle/nerf_oracle.c, orc_colormap_jet_lut): Octave's jet(256) at x = k / 255, entry * 255.f rounded to nearest even; B, G, R
static void jet_lut_host(uint8_t *lut)
{
    for (int k = 0; k < 256; k++) {
        const double x = (double)k / 255.0;
        const double c[3] = {fmin(4.0 * x - 1.5, -4.0 * x + 4.5), fmin(4.0 * x - 0.5, -4.0 * x + 3.5), fmin(4.0 * x + 0.5, -4.0 * x + 2.5)};
        for (int ch = 0; ch < 3; ch++) {
            const float entry = (float)(c[ch] < 0.0 ? 0.0 : (c[ch] > 1.0 ? 1.0 : c[ch]));
            const long v = lrintf(entry * 255.0f);
            lut[k * 3 + (2 - ch)] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

static std::mutex g_lut_mu;
static uint8_t *g_lut_dev[64] = {};          // per device

static int jet_lut_device(const uint8_t **out, hipStream_t st)
{
    int dev = 0;
    NRF_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_lut_mu);
    uint8_t *&p = g_lut_dev[dev & 63];
    if (!p) {
        uint8_t host[256 * 3];
        jet_lut_host(host);
        NRF_HIP(hipMalloc(reinterpret_cast<void **>(&p), sizeof(host)));
        NRF_HIP(hipMemcpy(p, host, sizeof(host), hipMemcpyHostToDevice));          // 768 bytes, once per device
    }
    (void)st;
    *out = p;
    return NRF_OK;
}

static int lerf_lanes(const nrf_lerf_renderer *r) { return r->lanes < 1 ? 1 : (r->lanes > NRF_LERF_MAX_LANES ? NRF_LERF_MAX_LANES : r->lanes); }

struct LerfPlan {
    int s, ni, sf, E;
    bool split, exact, geo;
};

static int lerf_plan(const nrf_lerf_renderer *r, const nrf_render_params *p, LerfPlan *pl, const char *who)
{
    const nrf_mlp *m = r->desc.lerf;
    pl->s = p->n_samples; pl->ni = p->n_importance; pl->sf = p->n_samples + p->n_importance; pl->E = r->embed_dim;
    if (p->perturb > 0.0f || p->raw_noise_std > 0.0f || p->precond_alpha > 0.0f || p->has_cone) {
        set_error("%s: perturb / noise / preconditioning / TangentScatter are the RNG branches of training; the LeRF render pass is built for ThinRay = true, Perturb = 0", who);
        return NRF_ERR_UNSUPPORTED;
    }
    if (!nrf_lerf_mfma_available(m)) { set_error("%s: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768", who); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG(pl->s >= 32 && pl->s % 32 == 0 && pl->ni > 0 && pl->sf % 32 == 0, "%s: n_samples and n_samples + n_importance must be multiples of 32, n_importance > 0 (got %d + %d)", who, pl->s, pl->ni);
    pl->split = m->lerf_precision == NRF_PREC_F16_SPLIT;
    pl->geo = pl->split;                                                                   // the sigma pass hands (sigma, geo32) to the embedding pass
    pl->exact = pl->split && p->coarse_mode != NRF_COARSE_FULL && nrf_lerf_sigma_exact_available(m);
    return NRF_OK;
}

static size_t lerf_chunk_ws(const LerfPlan &pl, int64_t n)
{
    const size_t cols = (size_t)n * pl.sf;
    size_t b = 0;
    auto add = [&](size_t bytes) { b = align_up(b, 256) + bytes; };
    add((size_t)n * pl.s * 4);              // z
    add((size_t)n * pl.s * 12);             // pts
    add(cols * 256);                        // x: [16][cols][8] halfs
    add(cols);                              // keep
    add(cols * 4);                          // sigma_le by column
    if (pl.geo) add(nrf_lerf_geo_bytes((int64_t)cols));
    add((size_t)n * pl.s * 4);              // coarse weights
    add((size_t)n * 4 * 3);                 // coarse depth / disp / acc
    add((size_t)n * pl.sf * 4);             // z_fine
    add((size_t)n * pl.sf * 4);             // merge map
    add((size_t)n * pl.ni * 4);             // new depths
    add((size_t)n * pl.ni * 12);            // new points
    add((size_t)n * pl.sf * 4);             // fine weights (when the caller does not want them)
    add((size_t)n * 4 * 3);                 // fine depth / disp / acc (likewise)
    add((size_t)n * pl.E * 4);              // un-normalised embedding
    add((size_t)n * pl.E * 4);              // normalised embedding (when the caller wants the relevancy only)
    add((size_t)n * 4);                     // ones
    return align_up(b, 256);
}

static nrf_lerf_outputs slice(const nrf_lerf_outputs &o, int64_t i, int s, int sf, int E)
{
    nrf_lerf_outputs q = o;
    if (q.d_embedding) q.d_embedding += i * E;
    if (q.d_disp) q.d_disp += i;
    if (q.d_acc) q.d_acc += i;
    if (q.d_depth) q.d_depth += i;
    if (q.d_weights) q.d_weights += i * sf;
    if (q.d_relevancy) q.d_relevancy += i * 2;
    if (q.d_z_coarse) q.d_z_coarse += i * s;
    if (q.d_weights_coarse) q.d_weights_coarse += i * s;
    if (q.d_z_fine) q.d_z_fine += i * sf;
    return q;
}

static int lerf_lanes_of(const nrf_lerf_renderer *r, int lanes, hipStream_t *st, hipEvent_t *fork, hipEvent_t *done)
{
    std::lock_guard<std::mutex> lk(r->lane_mu);
    int dev = 0;
    NRF_HIP(hipGetDevice(&dev));
    if (r->lane_device >= 0 && r->lane_device != dev) {
        (void)hipSetDevice(r->lane_device);
        r->drop_lanes();
        NRF_HIP(hipSetDevice(dev));
    }
    for (int i = 0; i < lanes; i++) {
        if (!r->lane[i]) NRF_HIP(hipStreamCreateWithFlags(&r->lane[i], hipStreamNonBlocking));
        if (!r->lane_done[i]) NRF_HIP(hipEventCreateWithFlags(&r->lane_done[i], hipEventDisableTiming));
        st[i] = r->lane[i]; done[i] = r->lane_done[i];
    }
    if (!r->lane_fork) NRF_HIP(hipEventCreateWithFlags(&r->lane_fork, hipEventDisableTiming));
    *fork = r->lane_fork;
    r->lane_device = dev;
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" int nrf_view_check(const nrf_view *v, const char *who);

// the renderer's two modules, for lerf_train.hip (library-internal: not exported)
extern "C" const nrf_hash *nrf_lerf_renderer_lang_embed(const nrf_lerf_renderer *r) { return r ? r->desc.lang_embed : nullptr; }
extern "C" const nrf_mlp *nrf_lerf_renderer_head(const nrf_lerf_renderer *r) { return r ? r->desc.lerf : nullptr; }

extern "C" NRF_API int nrf_lerf_renderer_last_features(const nrf_lerf_renderer *r, const void **d_feats_lm, int64_t *cols, const uint8_t **d_keep_cols, const int32_t **d_src, int64_t *n, int *sf,
                                                       uint64_t *serial)
{
    NRF_CHECK_ARG(r && d_feats_lm && cols && d_keep_cols && d_src && n && sf, "nrf_lerf_renderer_last_features: null pointer");
    if (serial) *serial = r->chunk_serial;
    if (!r->last_view.valid) { set_error("nrf_lerf_renderer_last_features: the last render call left no feature view (several chunks, or none yet)"); return NRF_ERR_UNSUPPORTED; }
    *d_feats_lm = r->last_view.feats; *cols = r->last_view.cols; *d_keep_cols = r->last_view.keep; *d_src = r->last_view.src; *n = r->last_view.n; *sf = r->last_view.sf;
    return NRF_OK;
}

extern "C" {

int nrf_lerf_relevancy(const float *d_embeds, int64_t n, int embed_dim, const float *d_positives, int n_pos, const float *d_negatives, int n_neg, int positive_id,
                       float *d_out, void *stream)
{
    NRF_CHECK_ARG(n >= 0 && embed_dim >= 1 && n_pos >= 1 && n_neg >= 1 && positive_id >= 0 && positive_id < n_pos,
                  "nrf_lerf_relevancy: need n >= 0, embed_dim >= 1, at least one positive and one negative phrase, 0 <= positive_id < n_pos (got n %lld, E %d, P %d, Q %d, id %d)",
                  (long long)n, embed_dim, n_pos, n_neg, positive_id);
    if (n == 0) return NRF_OK;
    NRF_CHECK_ARG(d_embeds && d_positives && d_negatives && d_out, "nrf_lerf_relevancy: null pointer");
    const size_t lds = (size_t)(1 + n_neg) * embed_dim * sizeof(float);
    NRF_CHECK_ARG(lds <= 64 * 1024, "nrf_lerf_relevancy: %d negative phrases of %d floats exceed the 64 KB phrase buffer", n_neg, embed_dim);
    const int64_t blocks = ceil_div(n, (int64_t)REL_WAVES);
    hipLaunchKernelGGL(k_lerf_relevancy, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(64 * REL_WAVES), lds, as_stream(stream), d_embeds, n, embed_dim,
                       d_positives + (size_t)positive_id * embed_dim, d_negatives, n_neg, d_out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_colormap_jet_lut(uint8_t *lut_host)
{
    NRF_CHECK_ARG(lut_host, "nrf_colormap_jet_lut: null pointer");
    jet_lut_host(lut_host);
    return NRF_OK;
}

int nrf_colormap_jet_u8(const uint8_t *d_gray, int64_t n, uint8_t *d_bgr, void *stream)
{
    NRF_CHECK_ARG(n >= 0, "nrf_colormap_jet_u8: negative count");
    if (n == 0) return NRF_OK;
    NRF_CHECK_ARG(d_gray && d_bgr, "nrf_colormap_jet_u8: null pointer");
    const uint8_t *lut = nullptr;
    NRF_TRY(jet_lut_device(&lut, as_stream(stream)));
    hipLaunchKernelGGL(k_lut_u8, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), d_gray, n, lut, d_bgr);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_relevancy_image(const float *d_relevancy, int64_t n, int rel_stride, uint8_t *d_bgr, void *stream)
{
    NRF_CHECK_ARG(n >= 0 && rel_stride >= 1, "nrf_relevancy_image: bad sizes");
    if (n == 0) return NRF_OK;
    NRF_CHECK_ARG(d_relevancy && d_bgr, "nrf_relevancy_image: null pointer");
    const uint8_t *lut = nullptr;
    NRF_TRY(jet_lut_device(&lut, as_stream(stream)));
    hipLaunchKernelGGL(k_relevancy_image, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), d_relevancy, n, rel_stride, lut, d_bgr);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_lerf_renderer_create(const nrf_lerf_renderer_desc *desc, nrf_lerf_renderer **out)
{
    NRF_CHECK_ARG(desc && out && desc->lang_embed && desc->lerf, "nrf_lerf_renderer_create: null pointer");
    NRF_CHECK_ARG(desc->lerf->family == MLP_LERF, "nrf_lerf_renderer_create: the head must be a LeRF handle (nrf_mlp_lerf_create)");
    NRF_CHECK_ARG(nrf_hash_output_dims(desc->lang_embed) == desc->lerf->in_dims, "nrf_lerf_renderer_create: the embedder emits %d features, the LeRF head takes %d",
                  nrf_hash_output_dims(desc->lang_embed), desc->lerf->in_dims);
    nrf_lerf_renderer *r = new nrf_lerf_renderer();
    r->desc = *desc;
    r->embed_dim = desc->lerf->out_dims - 1;
    *out = r;
    return NRF_OK;
}

void nrf_lerf_renderer_destroy(nrf_lerf_renderer *r) { delete r; }

int nrf_lerf_renderer_set_lanes(nrf_lerf_renderer *r, int lanes)
{
    NRF_CHECK_ARG(r && lanes >= 1 && lanes <= NRF_LERF_MAX_LANES, "nrf_lerf_renderer_set_lanes: 1 .. %d lanes", NRF_LERF_MAX_LANES);
    r->lanes = lanes;
    return NRF_OK;
}

int nrf_lerf_set_prompts(nrf_lerf_renderer *r, const float *positives, int n_pos, const float *negatives, int n_neg, int on_device, void *stream)
{
    NRF_CHECK_ARG(r, "nrf_lerf_set_prompts: null renderer");
    NRF_CHECK_ARG(n_pos >= 0 && n_neg >= 0 && (n_pos == 0) == (n_neg == 0), "nrf_lerf_set_prompts: positives and negatives come together (got %d and %d); 0 and 0 clears them", n_pos, n_neg);
    hipStream_t st = as_stream(stream);
    NRF_HIP(hipStreamSynchronize(st));                      // a render in flight may still read the old prompts
    if (r->d_pos) { NRF_HIP(hipFree(r->d_pos)); r->d_pos = nullptr; }
    if (r->d_neg) { NRF_HIP(hipFree(r->d_neg)); r->d_neg = nullptr; }
    r->n_pos = r->n_neg = 0;
    if (n_pos == 0) return NRF_OK;
    NRF_CHECK_ARG(positives && negatives, "nrf_lerf_set_prompts: null pointer");
    const size_t row = (size_t)r->embed_dim * sizeof(float);
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&r->d_pos), n_pos * row));
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&r->d_neg), n_neg * row));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    NRF_HIP(hipMemcpyAsync(r->d_pos, positives, n_pos * row, kind, st));
    NRF_HIP(hipMemcpyAsync(r->d_neg, negatives, n_neg * row, kind, st));
    NRF_HIP(hipStreamSynchronize(st));
    r->n_pos = n_pos; r->n_neg = n_neg;
    return NRF_OK;
}

size_t nrf_lerf_render_rays_workspace_bytes(const nrf_lerf_renderer *r, int64_t n, const nrf_render_params *p)
{
    if (!r || !p || n <= 0) return 0;
    LerfPlan pl;
    if (lerf_plan(r, p, &pl, "nrf_lerf_render_rays_workspace_bytes") != NRF_OK) return 0;
    return lerf_chunk_ws(pl, n);
}

static int lerf_render_rays_impl(const nrf_lerf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p, const float *d_t, const float *d_u,
                                 const nrf_lerf_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream, uint32_t *d_flag);

static inline bool lerf_detects(const nrf_render_params *p) { return p->overflow_policy != NRF_OVERFLOW_IGNORE; }

static int lerf_flag_buffers(const nrf_lerf_renderer *r)
{
    if (r->d_flag) return NRF_OK;
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&r->d_flag), sizeof(uint32_t)));
    NRF_HIP(hipHostMalloc(reinterpret_cast<void **>(&r->h_flag), 2 * sizeof(uint32_t), hipHostMallocDefault));          // [0] synchronous read-back, [1] deferred copy
    NRF_HIP(hipEventCreateWithFlags(&r->flag_ev, hipEventDisableTiming));
    return NRF_OK;
}

static int lerf_take_deferred(const nrf_lerf_renderer *r, bool wait, const char *who)
{
    if (!r->flag_pending) return NRF_OK;
    if (wait) NRF_HIP(hipEventSynchronize(r->flag_ev));
    else if (hipEventQuery(r->flag_ev) != hipSuccess) return NRF_OK;
    r->flag_pending = false;
    if (!r->h_flag[1]) return NRF_OK;
    r->flagged_calls++;
    set_error("%s: an EARLIER LeRF render call (NRF_OVERFLOW_DEFERRED) produced non-finite language densities / embeddings: an fp16 operand of the fused passes left its range", who);
    return NRF_ERR_NONFINITE;
}

// before a call's chunks are issued (clears the word) and after they have joined `st`
static int lerf_flag_begin(const nrf_lerf_renderer *r, const nrf_render_params *p, hipStream_t st, const char *who)
{
    NRF_CHECK_ARG(p->overflow_policy >= NRF_OVERFLOW_AUTO && p->overflow_policy <= NRF_OVERFLOW_IGNORE, "%s: overflow_policy %d is not an NRF_OVERFLOW_* value", who, p->overflow_policy);
    if (!lerf_detects(p)) return NRF_OK;
    NRF_TRY(lerf_flag_buffers(r));
    NRF_TRY(lerf_take_deferred(r, p->overflow_policy == NRF_OVERFLOW_DEFERRED, who));
    NRF_HIP(hipMemsetAsync(r->d_flag, 0, sizeof(uint32_t), st));
    return NRF_OK;
}

static int lerf_flag_end(const nrf_lerf_renderer *r, const nrf_render_params *p, hipStream_t st, const char *who)
{
    if (!lerf_detects(p)) return NRF_OK;
    if (p->overflow_policy == NRF_OVERFLOW_DEFERRED) {
        NRF_HIP(hipMemcpyAsync(r->h_flag + 1, r->d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        NRF_HIP(hipEventRecord(r->flag_ev, st));
        r->flag_pending = true;
        return NRF_OK;
    }
    NRF_HIP(hipMemcpyAsync(r->h_flag, r->d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NRF_HIP(hipStreamSynchronize(st));
    if (!r->h_flag[0]) return NRF_OK;
    r->flagged_calls++;
    set_error("%s: non-finite language densities / embeddings: an fp16 operand of the fused LeRF passes left its range (weights / features of unusual magnitude).  The pass has "
              "no fp32 single-call twin: render through the stage calls (nrf_mlp_forward NRF_PREC_F32 + nrf_raw2weights + nrf_render_clip_embedding)", who);
    return NRF_ERR_NONFINITE;
}

int nrf_lerf_renderer_nonfinite(const nrf_lerf_renderer *r, int64_t *flagged_calls)
{
    NRF_CHECK_ARG(r, "nrf_lerf_renderer_nonfinite: null pointer");
    const int rc = lerf_take_deferred(r, true, "nrf_lerf_renderer_nonfinite");
    if (flagged_calls) *flagged_calls = r->flagged_calls;
    return rc == NRF_ERR_NONFINITE ? NRF_OK : rc;
}

int nrf_lerf_render_rays(const nrf_lerf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p, const float *d_t, const float *d_u,
                         const nrf_lerf_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && p && out, "nrf_lerf_render_rays: null pointer");
    if (n == 0) return lerf_render_rays_impl(r, d_rays, ray_stride, n, p, d_t, d_u, out, d_workspace, workspace_bytes, stream, nullptr);
    NRF_TRY(lerf_flag_begin(r, p, as_stream(stream), "nrf_lerf_render_rays"));
    NRF_TRY(lerf_render_rays_impl(r, d_rays, ray_stride, n, p, d_t, d_u, out, d_workspace, workspace_bytes, stream, lerf_detects(p) ? r->d_flag : nullptr));
    return lerf_flag_end(r, p, as_stream(stream), "nrf_lerf_render_rays");
}

static int lerf_render_rays_impl(const nrf_lerf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p, const float *d_t, const float *d_u,
                                 const nrf_lerf_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream, uint32_t *d_flag)
{
    NRF_CHECK_ARG(r && p && out, "nrf_lerf_render_rays: null pointer");
    NRF_CHECK_ARG(n >= 0 && (ray_stride == 8 || ray_stride == 11), "nrf_lerf_render_rays: packed rays are [n, 8 | 11]");
    LerfPlan pl;
    NRF_TRY(lerf_plan(r, p, &pl, "nrf_lerf_render_rays"));
    r->last_view.valid = false;          // (set again at the end of the chunk)
    r->chunk_serial++;
    if (n == 0) return NRF_OK;
    NRF_CHECK_ARG(d_rays && d_t && d_u && d_workspace, "nrf_lerf_render_rays: null pointer");
    NRF_CHECK_ARG(n * (int64_t)pl.sf < ((int64_t)1 << 31), "nrf_lerf_render_rays: %lld rays x %d samples exceed the 2^31 columns of one chunk; lower Chunk", (long long)n, pl.sf);
    NRF_CHECK_ARG(!out->d_relevancy || r->n_pos > 0, "nrf_lerf_render_rays: relevancy asked for, but no prompts are set (nrf_lerf_set_prompts)");
    const size_t need = lerf_chunk_ws(pl, n);
    if (workspace_bytes < need) { set_error("nrf_lerf_render_rays: workspace %zu < %zu bytes", workspace_bytes, need); return NRF_ERR_WORKSPACE; }
    const nrf_hash *h = r->desc.lang_embed;
    const nrf_mlp *m = r->desc.lerf;
    const int s = pl.s, ni = pl.ni, sf = pl.sf, E = pl.E;
    const int64_t cols = n * (int64_t)sf, nc = n * (int64_t)s, nn = n * (int64_t)ni;
    LBump b(d_workspace);
    float *z = out->d_z_coarse ? out->d_z_coarse : b.take<float>((size_t)nc);
    if (out->d_z_coarse) (void)b.take<float>((size_t)nc);
    float *pts = b.take<float>((size_t)nc * 3);
    char *x = b.take<char>((size_t)cols * 256);
    uint8_t *keep = b.take<uint8_t>((size_t)cols);
    float *sig = b.take<float>((size_t)cols);
    char *geo = pl.geo ? b.take<char>(nrf_lerf_geo_bytes(cols)) : nullptr;
    float *w_c_ws = b.take<float>((size_t)nc);
    float *w_c = out->d_weights_coarse ? out->d_weights_coarse : w_c_ws;
    float *dda_c = b.take<float>((size_t)n * 3);
    float *zf_ws = b.take<float>((size_t)cols);
    float *zf = out->d_z_fine ? out->d_z_fine : zf_ws;
    int32_t *src = b.take<int32_t>((size_t)cols);
    float *z_new = b.take<float>((size_t)nn);
    float *pts_new = b.take<float>((size_t)nn * 3);
    float *w_f_ws = b.take<float>((size_t)cols);
    float *w_f = out->d_weights ? out->d_weights : w_f_ws;
    float *dda_f = b.take<float>((size_t)n * 3);
    float *acc = b.take<float>((size_t)n * E);
    float *emb_ws = b.take<float>((size_t)n * E);
    float *ones = b.take<float>((size_t)n);
    const float *dirs = d_rays + 3;

    NRF_TRY(nrf_z_vals(d_rays, ray_stride, n, d_t, s, p->lindisp, z, stream));                                                 // :113-135
    NRF_TRY(nrf_points(d_rays, ray_stride, z, n, s, pts, stream));                                                            // :137
    NRF_TRY(nrf_hash_encode_lm_f16_strided(h, pts, nc, x, cols, keep, stream));                                               // RunLENetwork (:5-25), coarse points
    if (pl.exact) NRF_TRY(nrf_lerf_sigma_exact_lm_strided(m, x, cols, keep, nc, sig, geo, cols, stream));
    else if (pl.geo) NRF_TRY(nrf_lerf_sigma_geo_lm_strided(m, x, cols, keep, nc, sig, geo, cols, stream));
    else NRF_TRY(nrf_lerf_sigma_lm_strided(m, x, cols, keep, nc, sig, stream));
    NRF_TRY(nrf_raw2weights(sig, 1, 0, z, dirs, ray_stride, n, s, w_c, dda_c, dda_c + n, dda_c + 2 * n, stream));            // RawToLEOutputs (:27-77), coarse weights
    NRF_TRY(nrf_fine_depths_merge(z, w_c, n, s, d_u, ni, p->sum_vec, zf, src, z_new, stream));                                // :144-150
    NRF_TRY(nrf_points(d_rays, ray_stride, z_new, n, ni, pts_new, stream));
    char *x_new = x + (size_t)nc * 16;                                                                                       // column n*s of level 0 (8 halfs per column)
    NRF_TRY(nrf_hash_encode_lm_f16_strided(h, pts_new, nn, x_new, cols, keep + nc, stream));                                  // the fine pass encodes nothing twice
    if (pl.geo) NRF_TRY(nrf_lerf_sigma_geo_lm_strided(m, x_new, cols, keep + nc, nn, sig + nc, geo + (size_t)nc * 32, cols, stream));
    else NRF_TRY(nrf_lerf_sigma_lm_strided(m, x_new, cols, keep + nc, nn, sig + nc, stream));
    float *depth = out->d_depth ? out->d_depth : dda_f, *disp = out->d_disp ? out->d_disp : dda_f + n, *accm = out->d_acc ? out->d_acc : dda_f + 2 * n;
    // RawToLEOutputs, fine pass, through the merge map (nrf_raw2weights_gather + the pass's non-finite word: every sample's sigma_le is looked at here)
    NRF_TRY(nrf::launch_raw2outputs(sig, zf, dirs, ray_stride, n, sf, 1, 0, 0, nullptr, disp, accm, w_f, depth, nrf::SigmaNoise{}, as_stream(stream), false, src, nullptr, 0, d_flag));
    if (out->d_embedding || out->d_relevancy) {
        if (pl.geo) NRF_TRY(nrf_lerf_render_embedding_lm_geo(m, x, cols, src, geo, cols, w_f, n, sf, acc, stream));
        else NRF_TRY(nrf_lerf_render_embedding_lm_gather(m, x, cols, src, w_f, n, sf, acc, stream));
        // the final normalise of RenderCLIPEmbedding (LeRFRenderer.h:53): one "sample" of weight 1 per ray
        NRF_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ones), 0x3f800000, (size_t)n, as_stream(stream)));
        float *emb = out->d_embedding ? out->d_embedding : emb_ws;
        NRF_TRY(nrf::launch_clip_embedding(acc, E, E, ones, n, 1, emb, as_stream(stream), d_flag));
        if (out->d_relevancy)
            NRF_TRY(nrf_lerf_relevancy(emb, n, E, r->d_pos, r->n_pos, r->d_neg, r->n_neg, 0, out->d_relevancy, stream));     // LeRFRenderer.cpp:79 (one positive phrase)
    }
    r->last_view.feats = x; r->last_view.cols = cols; r->last_view.keep = keep; r->last_view.src = src; r->last_view.n = n; r->last_view.sf = sf; r->last_view.valid = true;
    return NRF_OK;
}

static int64_t lerf_lane_chunk(int64_t n, int chunk, int lanes)
{
    if (lanes < 2 || n <= chunk) return 0;          // a batch of one chunk stays on the caller's stream
    return chunk;
}

size_t nrf_lerf_batchify_rays_workspace_bytes(const nrf_lerf_renderer *r, int64_t n, int chunk, const nrf_render_params *p)
{
    if (!r || !p || chunk <= 0 || n <= 0) return 0;
    const int lanes = lerf_lanes(r);
    const int64_t lc = lerf_lane_chunk(n, chunk, lanes);
    const size_t one = nrf_lerf_render_rays_workspace_bytes(r, n < chunk ? n : (int64_t)chunk, p);
    return lc > 0 ? (size_t)lanes * one : one;
}

int nrf_lerf_batchify_rays(const nrf_lerf_renderer *r, const float *d_rays, int ray_stride, int64_t n, int chunk, const nrf_render_params *p, const float *d_t,
                           const float *d_u, const nrf_lerf_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && p && out, "nrf_lerf_batchify_rays: null pointer");
    NRF_CHECK_ARG(chunk > 0 && n >= 0, "nrf_lerf_batchify_rays: Chunk must be positive");
    LerfPlan pl;
    NRF_TRY(lerf_plan(r, p, &pl, "nrf_lerf_batchify_rays"));
    if (n > 0) NRF_TRY(lerf_flag_begin(r, p, as_stream(stream), "nrf_lerf_batchify_rays"));
    uint32_t *const d_flag = (n > 0 && lerf_detects(p)) ? r->d_flag : nullptr;
    const int L = lerf_lanes(r);
    const int64_t lc = lerf_lane_chunk(n, chunk, L);
    const size_t part = lc > 0 ? nrf_lerf_render_rays_workspace_bytes(r, lc, p) : 0;
    if (part > 0 && (size_t)L * part <= workspace_bytes && d_workspace) {
        // the Chunk loop on lanes (as nrf_batchify_rays): consecutive chunks on L streams forked from and joined to the caller's, so that one chunk's gather-bound
        // F = 8 hash encode shares the CUs with another's matrix-bound passes.  Same kernels on the same slices: same results.
        hipStream_t st = as_stream(stream), lane[NRF_LERF_MAX_LANES];
        hipEvent_t fork = nullptr, done[NRF_LERF_MAX_LANES];
        NRF_TRY(lerf_lanes_of(r, L, lane, &fork, done));
        int rc = NRF_OK;
        if (hipEventRecord(fork, st) != hipSuccess) { set_error("nrf_lerf_batchify_rays: forking the lanes failed"); rc = NRF_ERR_HIP; }
        for (int j = 0; j < L && rc == NRF_OK; j++)
            if (hipStreamWaitEvent(lane[j], fork, 0) != hipSuccess) { set_error("nrf_lerf_batchify_rays: forking the lanes failed"); rc = NRF_ERR_HIP; }
        int k = 0;
        for (int64_t i = 0; i < n && rc == NRF_OK; i += lc, k = (k + 1) % L) {                                                // :206
            const int64_t mm = n - i < lc ? n - i : lc;
            const nrf_lerf_outputs o = slice(*out, i, pl.s, pl.sf, pl.E);
            rc = lerf_render_rays_impl(r, d_rays + i * ray_stride, ray_stride, mm, p, d_t, d_u, &o, static_cast<char *>(d_workspace) + (size_t)k * part, part, lane[k], d_flag);
        }
        for (int j = 0; j < L; j++) {
            if (hipEventRecord(done[j], lane[j]) != hipSuccess || hipStreamWaitEvent(st, done[j], 0) != hipSuccess) {
                if (rc == NRF_OK) { set_error("nrf_lerf_batchify_rays: joining the lanes failed"); rc = NRF_ERR_HIP; }
                (void)hipStreamSynchronize(lane[j]);
            }
        }
        if (n > lc) r->last_view.valid = false;          // the view describes ONE chunk's workspace
        return (rc == NRF_OK && n > 0) ? lerf_flag_end(r, p, st, "nrf_lerf_batchify_rays") : rc;
    }
    for (int64_t i = 0; i < n; i += chunk) {                                                                                  // :206
        const int64_t mm = n - i < chunk ? n - i : (int64_t)chunk;
        const nrf_lerf_outputs o = slice(*out, i, pl.s, pl.sf, pl.E);
        NRF_TRY(lerf_render_rays_impl(r, d_rays + i * ray_stride, ray_stride, mm, p, d_t, d_u, &o, d_workspace, workspace_bytes, stream, d_flag));
    }
    if (n > chunk) r->last_view.valid = false;          // the view describes ONE chunk's workspace
    return n > 0 ? lerf_flag_end(r, p, as_stream(stream), "nrf_lerf_batchify_rays") : NRF_OK;
}

size_t nrf_lerf_render_rows_workspace_bytes(const nrf_lerf_renderer *r, const nrf_view *v, const nrf_render_params *p)
{
    if (!r || !v || !p || v->chunk <= 0 || v->rows < 0 || v->w <= 0) return 0;
    const int64_t n = (int64_t)v->rows * v->w;
    return align_up((size_t)n * (v->use_viewdirs ? 11 : 8) * sizeof(float), 256) + 256 + nrf_lerf_batchify_rays_workspace_bytes(r, n, v->chunk, p);
}

int nrf_lerf_render_rows(const nrf_lerf_renderer *r, const nrf_view *v, const nrf_render_params *p, const float *d_t, const float *d_u, const nrf_lerf_outputs *out,
                         float *d_rays_out, float *d_near_far, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && p && out, "nrf_lerf_render_rows: null pointer");
    NRF_TRY(nrf_view_check(v, "nrf_lerf_render_rows"));
    const int64_t n = (int64_t)v->rows * v->w;
    if (n == 0) return nrf_view_rays(v, nullptr, d_near_far, stream);
    const int stride = v->use_viewdirs ? 11 : 8;
    const size_t need = nrf_lerf_render_rows_workspace_bytes(r, v, p);
    if (workspace_bytes < need) { set_error("nrf_lerf_render_rows: workspace %zu < %zu bytes", workspace_bytes, need); return NRF_ERR_WORKSPACE; }
    LBump b(d_workspace);
    float *rays = d_rays_out ? d_rays_out : b.take<float>((size_t)n * stride);
    void *ws = b.take<char>(0);
    NRF_TRY(nrf_view_rays(v, rays, d_near_far, stream));                                                                     // LeRFRenderer.cpp:275-305
    return nrf_lerf_batchify_rays(r, rays, stride, n, v->chunk, p, d_t, d_u, out, ws, workspace_bytes - b.off, stream);      // :308
}

}  // extern "C"
