// lerf_train.hip -- N1, the LeRF branch of the optimisation step (NeRFExecutor.h:955-982):
//   lerf_render_result = LeRFRenderer->Render(ray batch)                                   :958-961
//   lang_loss = huber_loss(RenderedLangEmbedding, target, reduction none, delta 1.25).sum(-1).nanmean()     :970-974        nrf_huber_rows_nanmean
//   lang_loss.backward()  into LeRFImpl's two bias-free MLPs and the F = 8 language grid        :981            nrf_lerf_head_backward / nrf_lerf_backward_points
// What carries gradient is the FINE pass (z_samples are detached, LeRFRenderer.cpp:150):
//   pts -> CuHashEmbedder (language grid) -> LeRFImpl::forward (LeRF.cpp:86-108: sigma net in -> H.. -> 1 + geo; LE net cat[geo, in] -> H.. -> E; normalize eps 1e-8)
//       -> sigma_le[~keep] = 0 (LeRFRenderer.cpp:37-38) -> RawToLEOutputs' weights (:38-66, TruncExp CustomOps.cpp:5-15) -> RenderCLIPEmbedding (LeRFRenderer.h:45-54).
// fp32 throughout, from the generic layer kernels of mlp.hip (forward = the oracle's FMA chains; dW = TN products with one atomic add per element and workgroup) or, by
// default, the library GEMMs of gemm_f32.hip: the forward is RECOMPUTED here chunk by chunk with every layer input kept -- the render pass (fused matrix-core kernels) never
// forms raw_le [n, S, 769].  The LAST layer (256 -> 768 at main.cpp sizes), its normalize and RenderCLIPEmbedding run in their Gram form (k_ltg_ray_u below): 136 -> 77 ms
// per step of 16 384 rays x 192 samples (docs/history/profiles/round5/r5K_*, r5O_*).
// Pinned by LibTorch autograd over the compiled LeRF.cpp / RawToOutputs weights / the reference's inline RenderCLIPEmbedding: goldens train_lerf*.
#include "encode.h"
#include "mlp.h"
#include "nrf_math.h"

#include <atomic>

struct nrf_lerf_renderer;
extern "C" const nrf_hash *nrf_lerf_renderer_lang_embed(const nrf_lerf_renderer *r);
extern "C" const nrf_mlp *nrf_lerf_renderer_head(const nrf_lerf_renderer *r);

namespace nrf {

// sample points per pass: 2 n_layers + 4 activation / gradient buffers of this many rows x the widest layer (main.cpp sizes: 8 x 400 MB).  With 2^15 a pass was 170
// rays = 170 workgroups of the per-ray kernel on 256 CUs, and every layer product a launch of a few tens of microseconds (docs/history/profiles/round5/r5o_*)
#ifndef NRF_LT_CHUNK_LOG2
#define NRF_LT_CHUNK_LOG2 17
#endif
constexpr int64_t LT_CHUNK_PTS = (int64_t)1 << NRF_LT_CHUNK_LOG2;
// With the last layer in its Gram form (below) nothing embedding-wide exists per sample: the activation rows are as wide as the widest OTHER layer (256 instead of 768 at
// main.cpp sizes) and a pass takes 2^NRF_LT_GRAM_CHUNK_LOG2 points (fewer, larger launches: the step is ~1 000 launches at 2^17; docs/history/profiles/round5/r5N_*, r5O_*)
#ifndef NRF_LT_GRAM_CHUNK_LOG2
#define NRF_LT_GRAM_CHUNK_LOG2 20
#endif
static bool lerf_train_gram(const nrf_mlp *m, int s);
static int lt_width(const nrf_mlp *m, int s)
{
    if (!lerf_train_gram(m, s)) return m->max_width;
    int w = 8;          // the per-sample scalars of the Gram form take 8 columns of a buffer
    for (size_t l = 0; l + 1 < m->layers.size(); l++) { w = w > m->layers[l].in ? w : m->layers[l].in; w = w > m->layers[l].out ? w : m->layers[l].out; }
    w = w > m->layers.back().in ? w : m->layers.back().in;
    return w;
}
static int64_t lt_chunk_pts(const nrf_mlp *m, int s) { return lerf_train_gram(m, s) ? ((int64_t)1 << NRF_LT_GRAM_CHUNK_LOG2) : LT_CHUNK_PTS; }
// rays per backward pass: n rays in passes of at most cpts / s rays, cut EVENLY (16 384 rays at 5 461 per pass are four passes of 4 096, not three full ones and a
// one-ray tail whose products fall back to the small-shape paths)
static int64_t lt_rays_per_pass(int64_t n, int64_t cpts, int s)
{
    const int64_t cap = cpts / s > 0 ? cpts / s : 1;
    if (n <= cap) return n < 1 ? 1 : n;
    const int64_t passes = (n + cap - 1) / cap;
    return (n + passes - 1) / passes;
}

__device__ __forceinline__ double lt_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide sum of one double per thread (blockDim.x = 256: four waves); every thread gets the result
__device__ __forceinline__ double lt_block_sum(double v, double *sh /*[4]*/)
{
    v = lt_wave_sum(v);
    __syncthreads();                                 // sh may still be read from the previous use
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// huber(reduction none, delta) summed over a row; row_loss[i] (NaN for a row that holds a NaN)
__global__ void k_huber_rows(int64_t n, int e, float delta, const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ row_loss)
{
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int k = lane; k < e; k += 64) {
        const float d = pred[i * e + k] - target[i * e + k];
        const float z = fabsf(d);
        acc += (z < delta) ? 0.5 * (double)z * (double)z : (double)delta * ((double)z - 0.5 * (double)delta);
    }
    acc = lt_wave_sum(acc);
    if (lane == 0) row_loss[i] = (float)acc;
}

// nanmean over the rows: out[0] = loss, out[1] = 1 / (count of non-NaN rows)
__global__ void k_nanmean(int64_t n, const float *__restrict__ row_loss, float *__restrict__ loss, float *__restrict__ inv_count)
{
    __shared__ double sh[4];
    double acc = 0.0, cnt = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = row_loss[i];
        if (v == v) { acc += (double)v; cnt += 1.0; }
    }
    acc = lt_block_sum(acc, sh);
    cnt = lt_block_sum(cnt, sh);
    if (threadIdx.x == 0) { *loss = (float)(acc / cnt); *inv_count = 1.0f / (float)cnt; }
}

// d loss / d pred: nansum hands a NaN row 0 instead of 1 / count; huber's own derivative multiplies it (NaN * 0 = NaN at a NaN element: what LibTorch leaves there)
__global__ void k_huber_rows_grad(int64_t n, int e, float delta, const float *__restrict__ pred, const float *__restrict__ target, const float *__restrict__ row_loss,
                                  const float *__restrict__ inv_count, float *__restrict__ grad)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * e) return;
    const int64_t i = idx / e;
    const float rl = row_loss[i];
    const float go = (rl == rl) ? inv_count[0] : 0.0f;
    const float d = pred[idx] - target[idx];
    grad[idx] = (d < -delta) ? -delta * go : (d > delta ? delta * go : d * go);
}

// h [p, E] (row stride hs) -> le = h / max(||h||, 1e-8) IN PLACE, nrm[p] = ||h||        (LeRF.cpp:104; one wave per point)
__global__ void k_lt_point_norm(int64_t p, int e, float *__restrict__ h, int hs, float *__restrict__ nrm)
{
    const int64_t pt = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (pt >= p) return;
    const int lane = threadIdx.x & 63;
    double ss = 0.0;
    for (int k = lane; k < e; k += 64) { const float v = h[pt * hs + k]; ss += (double)v * (double)v; }
    ss = lt_wave_sum(ss);
    const float nr = (float)sqrt(ss);
    const float c = fmaxf(nr, 1e-8f);
    for (int k = lane; k < e; k += 64) h[pt * hs + k] = h[pt * hs + k] / c;
    if (lane == 0) nrm[pt] = nr;
}

// One workgroup (256 threads) per ray: weights of the ray's s samples, RenderCLIPEmbedding forward and backward, the per-sample normalize backward (le -> g_h in place)
// and the weights' backward -> g_sigma.  Dynamic LDS: 7 s + 2 e floats.
//   sig33 [c, ss]: column 0 = sigma_le (unmasked: masked here with keep);   le / g_h [c, hs];   out: g33 [c, gs] column 0 = d loss / d sigma_le (0 where masked)
__global__ void __launch_bounds__(256) k_lt_ray(int s, int e, const float *__restrict__ sig33, int ss, const uint8_t *__restrict__ keep, const float *__restrict__ z,
                                                const float *__restrict__ dirs, int d_stride, const float *__restrict__ noise, float noise_std, float *__restrict__ le, int hs,
                                                const float *__restrict__ nrm, const float *__restrict__ g_rendered, float *__restrict__ g33, int gs,
                                                float *__restrict__ rendered, float *__restrict__ weights_out)
{
    extern __shared__ float lds[];
    float *alpha = lds, *trans = lds + s, *xx = lds + 2 * s, *lt = lds + 3 * s, *wgt = lds + 4 * s, *gw = lds + 5 * s, *sg = lds + 6 * s;
    float *v = lds + 7 * s, *gv = v + e;
    __shared__ double sh[4];
    const int64_t ray = blockIdx.x;
    const int64_t p0 = ray * s;
    const float *dv = dirs + ray * d_stride;
    const float dn = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
    for (int j = threadIdx.x; j < s; j += blockDim.x) {
        float sr = (keep && !keep[p0 + j]) ? 0.0f : sig33[(p0 + j) * ss];
        if (noise) sr = sr + noise[p0 + j] * noise_std;                    // RawNoiseStd > 0 (LeRFRenderer.cpp:50-51): the density that went through relu / alpha
        sg[j] = sr;
        float dist = (j + 1 < s) ? (z[p0 + j + 1] - z[p0 + j]) : 1e10f;
        dist = dist * dn;
        const float x = -(sr > 0.0f ? sr : 0.0f) * dist;
        xx[j] = x;
        alpha[j] = -nrf_expf(x) + 1.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                                // the log-space transmittance: a double running sum, each prefix rounded to fp32 (ATen's cumsum)
        double logt = 0.0; float tprev = 0.0f;
        for (int j = 0; j < s; j++) {
            lt[j] = tprev; trans[j] = nrf_expf(tprev);
            const float om = 1.0f - alpha[j];
            logt += (double)nrf_logf(om > 1e-10f ? om : 1e-10f);
            tprev = (float)logt;
            wgt[j] = alpha[j] * trans[j];
        }
    }
    __syncthreads();
    if (weights_out) for (int j = threadIdx.x; j < s; j += blockDim.x) weights_out[p0 + j] = wgt[j];
    // v = sum_j w_j le_j ; ||v|| ; g . v
    double vss = 0.0, gdot = 0.0;
    for (int k = threadIdx.x; k < e; k += blockDim.x) {
        double a = 0.0;
        for (int j = 0; j < s; j++) a += (double)(wgt[j] * le[(p0 + j) * hs + k]);
        const float vk = (float)a;
        v[k] = vk;
        vss += (double)vk * (double)vk;
        gdot += (double)g_rendered[ray * e + k] * (double)vk;
    }
    vss = lt_block_sum(vss, sh);
    gdot = lt_block_sum(gdot, sh);
    const float vn = (float)sqrt(vss), vc = fmaxf(vn, 1e-8f);
    const bool through_norm = vn >= 1e-8f && vn > 0.0f;                     // clamp_min passes the gradient to ||v|| only where it did not clamp
    for (int k = threadIdx.x; k < e; k += blockDim.x) {
        if (rendered) rendered[ray * e + k] = v[k] / vc;
        float g = g_rendered[ray * e + k] / vc;
        if (through_norm) g -= (float)(gdot / ((double)vc * (double)vc)) * (v[k] / vn);
        gv[k] = g;
    }
    __syncthreads();
    // per sample (a wave each): t = le . g_v = d loss / d w ; g_le = w g_v ; g_h = g_le / c - [||h|| >= eps] (g_le . h / c^2) h / ||h||, h = le c
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = wave; j < s; j += 4) {
        float *row = le + (p0 + j) * hs;
        double t = 0.0;
        for (int k = lane; k < e; k += 64) t += (double)row[k] * (double)gv[k];
        t = lt_wave_sum(t);
        const float nr = nrm[p0 + j], c = fmaxf(nr, 1e-8f), w = wgt[j];
        const bool thr = nr >= 1e-8f && nr > 0.0f;
        const float hdot_over_c2 = (float)((double)w * t * (double)c / ((double)c * (double)c));     // g_le . h / c^2 with h = le c
        for (int k = lane; k < e; k += 64) {
            float g = w * gv[k] / c;
            if (thr) g -= hdot_over_c2 * (row[k] * c / nr);
            row[k] = g;
        }
        if (lane == 0) gw[j] = (float)t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                                // weights backward (RawToLEOutputs :38-66, TruncExp::backward = grad * exp(clamp(x, -100, 5)))
        double suffix = 0.0;
        for (int j = s - 1; j >= 0; j--) {
            float g_alpha = gw[j] * trans[j];
            const float om = 1.0f - alpha[j];
            if (om >= 1e-10f) g_alpha -= (float)suffix / om;
            const float cl = lt[j] < -100.0f ? -100.0f : (lt[j] > 5.0f ? 5.0f : lt[j]);
            suffix += (double)(gw[j] * alpha[j] * nrf_expf(cl));
            const float cx = xx[j] < -100.0f ? -100.0f : (xx[j] > 5.0f ? 5.0f : xx[j]);
            const float g_x = -g_alpha * nrf_expf(cx);
            float dist = (j + 1 < s) ? (z[p0 + j + 1] - z[p0 + j]) : 1e10f;
            dist = dist * dn;
            float gsig = (sg[j] > 0.0f) ? -g_x * dist : 0.0f;
            if (keep && !keep[p0 + j]) gsig = 0.0f;                        // index_put_ of a constant: no gradient to the masked sigma
            g33[(p0 + j) * gs] = gsig;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// GRAM FORM of the head's last layer (round 5).  LE1 (W: E x H, bias-free, LeRF.cpp:21-24) followed by normalize and RenderCLIPEmbedding is linear up to per-sample
// scalars, so -- as in the render pass (mlp_lerf_mfma.hip) -- nothing E-wide has to exist per SAMPLE.  With a_j the layer's input (H = 256), G = W^T W, s_j = G a_j:
//   ||h_j||^2 = a_j . s_j        v = sum_j w_j le_j = W u,  u = sum_j (w_j / c_j) a_j  (per ray)        t_j = le_j . g_v = (a_j . q) / c_j,  q = W^T g_v  (per ray)
//   g_a,j = W^T g_h,j = (w_j / c_j) q - beta_j s_j,    beta_j = [||h_j|| >= eps] w_j t_j / (c_j ||h_j||)
//   dW = sum_j g_h,j a_j^T = sum_rays g_v u^T  -  W (A^T diag(beta) A)
// Per sample: one H x H product (S = A G), two H-wide dot products and one H x H rank-1 share (A^T diag(beta) A); the E-wide products are per RAY.  The layer-wise path
// above forms h, le and g_h as [points, E] arrays and runs three points x E x H products.  Same derivatives (tests: the autograd goldens train_lerf*).
// Scratch per sample (pp, 8 floats): alpha, trans, lt, xx, sg, wgt, c, ||h||.
// ------------------------------------------------------------------------------------------------------------------------------------------------
constexpr int LTG_PP = 8;

// In-place EXCLUSIVE prefix sums of d[0 .. s) in double by one wave (call with the 64 lanes of wave 0 only; REVERSE: suffix sums, d[j] = sum_{i > j}).  Each lane takes
// a contiguous run, the runs' totals go through a wave scan.  The per-ray weights need a running sum over the samples (ATen's cumsum accumulates in double): taken by
// one thread it was 192 dependent software exp / log evaluations per ray -- most of these kernels' time.
template <bool REVERSE>
__device__ __forceinline__ void lt_excl_scan_wave0(double *d, int s, int lane)
{
    const int per = (s + 63) / 64;
    auto at = [&](int i) { return REVERSE ? s - 1 - i : i; };
    double run = 0.0;
    for (int e = 0; e < per; e++) { const int i = lane * per + e; if (i < s) run += d[at(i)]; }
    double incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    double acc = incl - run;
    for (int e = 0; e < per; e++) { const int i = lane * per + e; if (i < s) { const double t = d[at(i)]; d[at(i)] = acc; acc += t; } }
}

// per ray: the weights (as k_lt_ray), c_j and ||h_j|| from a_j . s_j, u = sum_j (w_j / c_j) a_j.  blockDim = 256; hd <= 256 features (one per thread)
__global__ void __launch_bounds__(256) k_ltg_ray_u(int s, int hd, const float *__restrict__ sig33, int ss, const uint8_t *__restrict__ keep, const float *__restrict__ z,
                                                   const float *__restrict__ dirs, int d_stride, const float *__restrict__ noise, float noise_std, const float *__restrict__ a,
                                                   int as, const float *__restrict__ sg_a, int sgs, float *__restrict__ pp, float *__restrict__ u, float *__restrict__ weights_out)
{
    extern __shared__ float lds[];
    float *alpha = lds, *trans = lds + s, *xx = lds + 2 * s, *lt = lds + 3 * s, *wgt = lds + 4 * s, *sg = lds + 5 * s, *cj = lds + 6 * s;
    double *dsum = reinterpret_cast<double *>(lds + 7 * s + (s & 1));          // [s] (8-byte aligned)
    const int64_t ray = blockIdx.x;
    const int64_t p0 = ray * s;
    const float *dv = dirs + ray * d_stride;
    const float dn = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < s; j += blockDim.x) {
        float sr = (keep && !keep[p0 + j]) ? 0.0f : sig33[(p0 + j) * ss];
        if (noise) sr = sr + noise[p0 + j] * noise_std;
        sg[j] = sr;
        float dist = (j + 1 < s) ? (z[p0 + j + 1] - z[p0 + j]) : 1e10f;
        dist = dist * dn;
        const float x = -(sr > 0.0f ? sr : 0.0f) * dist;
        xx[j] = x;
        const float al = -nrf_expf(x) + 1.0f;
        alpha[j] = al;
        const float om = 1.0f - al;
        dsum[j] = (double)nrf_logf(om > 1e-10f ? om : 1e-10f);
    }
    __syncthreads();
    if (wave == 0) lt_excl_scan_wave0<false>(dsum, s, lane);          // log-space transmittance: the double running sum, each prefix rounded to fp32 below
    __syncthreads();
    for (int j = threadIdx.x; j < s; j += blockDim.x) {
        const float tprev = (float)dsum[j];
        lt[j] = tprev; trans[j] = nrf_expf(tprev);
        wgt[j] = alpha[j] * trans[j];
    }
    __syncthreads();          // wgt[] of every sample is in LDS
    // ||h_j|| (a wave per sample) and, from the row a_j the wave has just read, its term of u = sum_j (w_j / c_j) a_j: every wave sums its own samples (in double), the
    // four waves' partial rows are added at the end -- a second pass over A, one thread per column and 192 dependent additions long, is gone
    constexpr int UK = 4;          // columns per lane (hd <= 256)
    double uacc[UK] = {0.0, 0.0, 0.0, 0.0};
    for (int j = wave; j < s; j += 4) {
        const float *ar = a + (p0 + j) * as, *sr = sg_a + (p0 + j) * sgs;
        float av[UK];
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < UK; i++) { const int k = lane + 64 * i; av[i] = k < hd ? ar[k] : 0.0f; if (k < hd) q += (double)av[i] * (double)sr[k]; }
        q = lt_wave_sum(q);
        const float nr = (float)sqrt(q > 0.0 ? q : 0.0);
        const float c = fmaxf(nr, 1e-8f);
        if (lane == 0) { cj[j] = c; pp[(p0 + j) * LTG_PP + 7] = nr; }
        const float wc = wgt[j] / c;
#pragma unroll
        for (int i = 0; i < UK; i++) uacc[i] += (double)(wc * av[i]);
    }
    double *upart = reinterpret_cast<double *>(lds + 7 * s + (s & 1)) + s;          // [4][hd] behind dsum (the launch reserves it)
#pragma unroll
    for (int i = 0; i < UK; i++) { const int k = lane + 64 * i; if (k < hd) upart[wave * hd + k] = uacc[i]; }
    __syncthreads();
    for (int j = threadIdx.x; j < s; j += blockDim.x) {
        float *o = pp + (p0 + j) * LTG_PP;
        o[0] = alpha[j]; o[1] = trans[j]; o[2] = lt[j]; o[3] = xx[j]; o[4] = sg[j]; o[5] = wgt[j]; o[6] = cj[j];
        if (weights_out) weights_out[p0 + j] = wgt[j];
    }
    if ((int)threadIdx.x < hd) {
        const int k = threadIdx.x;
        u[ray * hd + k] = (float)(((upart[k] + upart[hd + k]) + upart[2 * hd + k]) + upart[3 * hd + k]);
    }
}

// per ray: v = W u (given), rendered = v / max(||v||, eps), g_v = d loss / d v (RenderCLIPEmbedding's normalize backward, as k_lt_ray)
__global__ void __launch_bounds__(256) k_ltg_ray_v(int e, const float *__restrict__ v, const float *__restrict__ g_rendered, float *__restrict__ gv, float *__restrict__ rendered)
{
    __shared__ double sh[4];
    const int64_t ray = blockIdx.x;
    double vss = 0.0, gdot = 0.0;
    for (int k = threadIdx.x; k < e; k += blockDim.x) {
        const float vk = v[ray * e + k];
        vss += (double)vk * (double)vk;
        gdot += (double)g_rendered[ray * e + k] * (double)vk;
    }
    vss = lt_block_sum(vss, sh);
    gdot = lt_block_sum(gdot, sh);
    const float vn = (float)sqrt(vss), vc = fmaxf(vn, 1e-8f);
    const bool through_norm = vn >= 1e-8f && vn > 0.0f;
    for (int k = threadIdx.x; k < e; k += blockDim.x) {
        const float vk = v[ray * e + k];
        if (rendered) rendered[ray * e + k] = vk / vc;
        float g = g_rendered[ray * e + k] / vc;
        if (through_norm) g -= (float)(gdot / ((double)vc * (double)vc)) * (vk / vn);
        gv[ray * e + k] = g;
    }
}

// per ray: t_j, beta_j; g_a,j = ((w_j / c_j) q - beta_j s_j) . [a_j > 0] written OVER s_j; ba_j = beta_j a_j; the weights' backward -> g33 column 0 (as k_lt_ray's tail)
__global__ void __launch_bounds__(256) k_ltg_ray_g(int s, int hd, const float *__restrict__ q, const float *__restrict__ a, int as, float *__restrict__ sg_a, int sgs,
                                                   float *__restrict__ ba, int bs, const float *__restrict__ pp, const uint8_t *__restrict__ keep, const float *__restrict__ z,
                                                   const float *__restrict__ dirs, int d_stride, float *__restrict__ g33, int gs, int beta_only)
{
    extern __shared__ float lds[];
    float *gw = lds, *qs = lds + s;
    double *dsum = reinterpret_cast<double *>(lds + s + hd + ((s + hd) & 1));          // [s]
    const int64_t ray = blockIdx.x;
    const int64_t p0 = ray * s;
    for (int k = threadIdx.x; k < hd; k += blockDim.x) qs[k] = q[ray * hd + k];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = wave; j < s; j += 4) {
        const float *ar = a + (p0 + j) * as;
        float *sr = sg_a + (p0 + j) * sgs, *br = ba + (p0 + j) * bs;
        const float *o = pp + (p0 + j) * LTG_PP;
        const float w = o[5], c = o[6], nr = o[7];
        double d = 0.0;
        for (int k = lane; k < hd; k += 64) d += (double)ar[k] * (double)qs[k];
        d = lt_wave_sum(d);
        const float t = (float)(d / (double)c);
        const bool thr = nr >= 1e-8f && nr > 0.0f;
        const float beta = thr ? (float)((double)w * (double)t / ((double)c * (double)nr)) : 0.0f;
        const float wc = w / c;
        for (int k = lane; k < hd; k += 64) {
            const float av = ar[k];
            const float ga = wc * qs[k] - beta * sr[k];
            sr[k] = av > 0.0f ? ga : 0.0f;          // a is a ReLU output: its mask here instead of in a pass of its own (what run_relu_mask would do before the next layer)
            if (!beta_only) br[k] = beta * av;
        }
        if (beta_only && lane == 0) br[0] = beta;     // (the row beta_j a_j is formed by the weight-gradient product itself: its X operand is a, scaled per point)
        if (lane == 0) gw[j] = t;
    }
    __syncthreads();
    // the weights' backward (RawToLEOutputs :38-66, TruncExp::backward = grad * exp(clamp(x, -100, 5))): per-sample terms in parallel, their suffix sums by one wave
    for (int j = threadIdx.x; j < s; j += blockDim.x) {
        const float *o = pp + (p0 + j) * LTG_PP;
        const float ltj = o[2];
        const float cl = ltj < -100.0f ? -100.0f : (ltj > 5.0f ? 5.0f : ltj);
        dsum[j] = (double)(gw[j] * o[0] * nrf_expf(cl));
    }
    __syncthreads();
    if (wave == 0) lt_excl_scan_wave0<true>(dsum, s, lane);
    __syncthreads();
    {
        const float *dv = dirs + ray * d_stride;
        const float dn = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        for (int j = threadIdx.x; j < s; j += blockDim.x) {
            const float *o = pp + (p0 + j) * LTG_PP;
            const float alpha = o[0], trans = o[1], xj = o[3], sgj = o[4];
            float g_alpha = gw[j] * trans;
            const float om = 1.0f - alpha;
            if (om >= 1e-10f) g_alpha -= (float)dsum[j] / om;
            const float cx = xj < -100.0f ? -100.0f : (xj > 5.0f ? 5.0f : xj);
            const float g_x = -g_alpha * nrf_expf(cx);
            float dist = (j + 1 < s) ? (z[p0 + j + 1] - z[p0 + j]) : 1e10f;
            dist = dist * dn;
            float gsig = (sgj > 0.0f) ? -g_x * dist : 0.0f;
            if (keep && !keep[p0 + j]) gsig = 0.0f;
            g33[(p0 + j) * gs] = gsig;
        }
    }
}

// NRF_LERF_TRAIN_GRAM=0 in the environment: the layer-wise last layer (A/B runs); the Gram form needs the library GEMMs and a head of the built proportions
static std::atomic<int> g_lerf_train_gram{-1};          // -1: not decided yet (environment), 0 / 1
static bool lerf_train_gram(const nrf_mlp *m, int s)
{
    int st = g_lerf_train_gram.load();
    if (st < 0) { st = !(getenv("NRF_LERF_TRAIN_GRAM") && getenv("NRF_LERF_TRAIN_GRAM")[0] == '0'); g_lerf_train_gram.store(st); }
    const bool on = st != 0;
    const int hd = m->small.hidden_dim, E = m->small.hidden_dim_color;
    (void)E;
    return on && fp32_gemm_available() && m->small.num_layers >= 2 && hd <= 256 && (size_t)(9 * s + 9 * hd + 2) * sizeof(float) <= 60 * 1024 && !m->layers.back().d_bias;
}

// dw [E][hd] -= W [E][hd] . M [hd][hd]   (the Gram form's last term: a product of two small matrices, one thread per entry, k ascending)
__global__ void k_lt_sub_wm(int E, int hd, const float *__restrict__ w, const float *__restrict__ mm, float *__restrict__ dw)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E * hd) return;
    const int o = e / hd, j = e - o * hd;
    float acc = 0.0f;
    for (int k = 0; k < hd; k++) acc = fmaf(w[(size_t)o * hd + k], mm[(size_t)k * hd + j], acc);
    dw[e] -= acc;
}

// dst[pt][d_col + k] = src[pt][s_col + k] (+ add[pt][a_col + k]), k < ncols
__global__ void k_lt_copy_cols(int64_t p, int ncols, const float *__restrict__ src, int s_stride, int s_col, const float *__restrict__ add, int a_stride, int a_col,
                               float *__restrict__ dst, int d_stride, int d_col)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p * ncols) return;
    const int64_t pt = idx / ncols; const int k = (int)(idx - pt * ncols);
    float v = src[pt * s_stride + s_col + k];
    if (add) v = v + add[pt * a_stride + a_col + k];
    dst[pt * d_stride + d_col + k] = v;
}

// emb[q][8 l .. 8 l + 8) = the eight fp16 features of level l in column src[q] of the level-major table [16][cols] (16 bytes each) as fp32; keep[q] = keep_cols[src[q]].
// One thread per (point, level): the 16 threads of a point write its 512-byte row piece by piece, the reads of a level follow the (mostly ascending) map.
__global__ void k_lt_gather_feats(int64_t c, const uint4 *__restrict__ feats, int64_t cols, const int32_t *__restrict__ src, const uint8_t *__restrict__ keep_cols, float *__restrict__ emb,
                                  uint8_t *__restrict__ keep)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c * 16) return;
    const int64_t q = idx >> 4; const int l = (int)(idx & 15);
    const int64_t col = src[q];
    const uint4 v = feats[(int64_t)l * cols + col];
    const __half2 *hp = reinterpret_cast<const __half2 *>(&v);
    const float2 a = __half22float2(hp[0]), b = __half22float2(hp[1]), d = __half22float2(hp[2]), e = __half22float2(hp[3]);
    float4 *o = reinterpret_cast<float4 *>(emb + q * 128 + l * 8);
    o[0] = float4{a.x, a.y, b.x, b.y};
    o[1] = float4{d.x, d.y, e.x, e.y};
    if (l == 0) keep[q] = keep_cols[col];
}

static size_t head_ws_bytes(const nrf_mlp *m, int64_t n, int s)
{
    const int64_t cp = lt_chunk_pts(m, s);
    const int64_t rays = n < 1 ? 1 : (cp / s > 0 ? cp / s : 1);
    const int64_t c = (n < rays ? (n < 1 ? 1 : n) : rays) * s;
    const size_t buf = align_up((size_t)c * lt_width(m, s) * sizeof(float), 256);
    return buf * (m->layers.size() + 4) + align_up((size_t)c * sizeof(float), 256) + 1024;
}

// one chunk of whole rays: c = rays * s points
static int head_backward_chunk(const nrf_mlp *m, const float *emb, const uint8_t *keep, const float *z, const float *dirs, int d_stride, const float *noise, float noise_std,
                               int64_t rays, int s, const float *g_rendered, float *g_params, float *g_emb, float *rendered, float *weights, float *base, size_t buf,
                               float *nrm, const float *gram_in, hipStream_t st)
{
    const float *gram = gram_in;          // [hd x hd] Gram matrix, followed by this call's per-ray rows (nrf_lerf_head_backward)
    const auto &d = m->small;                    // MLP_LERF reuses: input_ch, num_layers, hidden_dim, geo_feat_dim; hidden_dim_color = embed dim (mlp.hip)
    const int W = gram ? lt_width(m, s) : m->max_width, nl = d.num_layers, NL = 2 * nl, E = d.hidden_dim_color, in = d.input_ch, geo = d.geo_feat_dim;
    const int64_t c = rays * s;
    const Seg none{nullptr, 0, 0, 0};
    std::vector<float *> H(NL);
    for (int l = 0; l < NL; l++) H[l] = base + (size_t)l * buf;
    float *G[4] = {base + (size_t)NL * buf, base + (size_t)(NL + 1) * buf, base + (size_t)(NL + 2) * buf, base + (size_t)(NL + 3) * buf};
    // ---- forward, every layer output kept (post-ReLU for hidden layers)                            LeRF.cpp:86-102 ----
    const Seg xin{emb, in, 0, in};
    Seg cur = xin;
    // the sigma net's output row [sigma | geo] is written 3 floats into its buffer row where that fits, so that the geo columns -- the LE net's first input segment --
    // start on a 16-byte boundary: the LE net's first layer and its weight gradient then take the vector paths (whole-row kernel, K = geo + in = 160)
    const int yo = (W >= 4 + geo && (W & 3) == 0) ? 3 : 0;
    for (int l = 0; l < nl; l++) {
        NRF_TRY(run_linear_fast(c, cur, none, m, m->layers[l], l != nl - 1, H[l], W, l == nl - 1 ? yo : 0, st));
        cur = Seg{H[l], W, 0, m->layers[l].out};
    }
    const float *h33 = H[nl - 1] + yo;                                    // column 0 = sigma_le, 1.. = geo_feat_le
    const Seg sgeo{h33, W, 1, geo};
    for (int l = 0; l < nl - (gram ? 1 : 0); l++) {          // Gram form: the last layer is never applied per sample
        NRF_TRY(run_linear_fast(c, l == 0 ? sgeo : cur, l == 0 ? xin : none, m, m->layers[nl + l], l != nl - 1, H[nl + l], W, 0, st));
        cur = Seg{H[nl + l], W, 0, m->layers[nl + l].out};
    }
    float *hle = H[NL - 1];                                               // h [c, E] -> le -> g_h, in place
    float *g33 = G[0];
    Seg g{hle, W, 0, E};
    int gi = 1, l_top = NL - 1;
    bool gram_masked = false;
    if (gram) {
        // ---- last layer, normalize, RenderCLIPEmbedding and their backward in the layer's INPUT space (see k_ltg_ray_u) ----
        const LinearLayer &L1 = m->layers[NL - 1];
        const int hd = L1.in;
        const float *wle = m->d_params + L1.w_off;                         // W [E][hd]
        const float *a = H[NL - 2];                                        // [c, hd] (stride W): post-ReLU input of the last layer
        float *sga = hle;                                                  // S = A G, then g_a in place: columns [0, hd) of the buffer h would have taken
        float *ba = G[1];                                                  // beta_j a_j
        float *pp = G[2];                                                  // per-sample scalars [c][8]
        float *mm = const_cast<float *>(gram) + (size_t)hd * hd, *u = mm + (size_t)hd * hd, *v = u + (size_t)rays * hd, *gv = v + (size_t)rays * E, *q = gv + (size_t)rays * E;      // M [hd x hd]; per-ray rows
        const int arith = c >= 4096 ? train_gemm_for(m) : 0;               // the split-precision matrix-core products of gemm_bf16x3.hip instead of rocBLAS
        if (arith) NRF_TRY(gemm_nt_split(arith, c, hd, Seg{a, W, 0, hd}, none, gram, hd, sga, W, nullptr, 0, nullptr, 0, st));           // S = A G = A G^T (G symmetric): an NT product
        else NRF_TRY(gemm_rm(st, false, false, c, hd, hd, 1.0f, a, W, gram, hd, 0.0f, sga, W));                             // S = A G   (G symmetric)
        hipLaunchKernelGGL(k_ltg_ray_u, dim3((unsigned)rays), dim3(256), (size_t)(7 * s + 1) * sizeof(float) + (size_t)(s + 4 * hd) * sizeof(double), st, s, hd, h33, W, keep, z, dirs, d_stride, noise, noise_std, a, W,
                           (const float *)sga, W, pp, u, weights);
        NRF_LAUNCH_CHECK();
        const int arith_r = rays >= 256 ? arith : 0;                      // (the per-ray products: a few thousand rows)
        if (arith_r) NRF_TRY(gemm_nt_split(arith_r, rays, E, Seg{u, hd, 0, hd}, none, wle, hd, v, E, nullptr, 0, nullptr, 0, st));          // V = U W^T
        else NRF_TRY(gemm_rm(st, false, true, rays, E, hd, 1.0f, u, hd, wle, hd, 0.0f, v, E));                             // V = U W^T
        hipLaunchKernelGGL(k_ltg_ray_v, dim3((unsigned)rays), dim3(256), 0, st, E, (const float *)v, g_rendered, gv, rendered);
        NRF_LAUNCH_CHECK();
        if (arith_r) NRF_TRY(gemm_nt_split(arith_r, rays, hd, Seg{gv, E, 0, E}, none, L1.d_wt, E, q, hd, nullptr, 0, nullptr, 0, st));     // Q = G_V W = G_V (W^T)^T: W^T [hd][E] is the layer's d_wt
        else NRF_TRY(gemm_rm(st, false, false, rays, hd, E, 1.0f, gv, E, wle, hd, 0.0f, q, hd));                           // Q = G_V W
        hipLaunchKernelGGL(k_ltg_ray_g, dim3((unsigned)rays), dim3(256), (size_t)(s + hd + 1) * sizeof(float) + (size_t)s * sizeof(double), st, s, hd, (const float *)q, a, W, sga, W, ba, W, (const float *)pp, keep, z,
                           dirs, d_stride, g33, W, arith ? 1 : 0);
        NRF_LAUNCH_CHECK();
        float *dw = g_params + L1.w_off;                                   // [E][hd]
        if (arith_r) NRF_TRY(gemm_tn_bf16x3(rays, Seg{gv, E, 0, E}, Seg{u, hd, 0, hd}, E, hd, 0, dw, st));                  // dW += G_V^T U
        else NRF_TRY(gemm_rm(st, true, false, E, hd, rays, 1.0f, gv, E, u, hd, 1.0f, dw, hd));                             // dW += G_V^T U
        if (arith) {                                                                                                       // M = A^T diag(beta) A: the weight-gradient product's shape
            NRF_HIP(hipMemsetAsync(mm, 0, (size_t)hd * hd * sizeof(float), st));
            NRF_TRY(gemm_tn_bf16x3(c, Seg{a, W, 0, hd}, Seg{a, W, 0, hd}, hd, hd, 0, mm, st, ba, W));          // X = diag(beta) A: beta_j sits in column 0 of ba's row j
        } else NRF_TRY(gemm_rm(st, true, false, hd, hd, c, 1.0f, a, W, ba, W, 0.0f, mm, hd));
        if (arith) {                                                                                                       // dW -= W M: 768 x 256 x 256, fp32 FMAs
            hipLaunchKernelGGL(k_lt_sub_wm, dim3((unsigned)ceil_div((int64_t)E * hd, (int64_t)256)), dim3(256), 0, st, E, hd, wle, (const float *)mm, dw);
            NRF_LAUNCH_CHECK();
        } else NRF_TRY(gemm_rm(st, false, false, E, hd, hd, -1.0f, wle, hd, mm, hd, 1.0f, dw, hd));                        // dW -= W M
        g = Seg{sga, W, 0, hd};
        gram_masked = true;                                                // k_ltg_ray_g has applied H[NL - 2]'s ReLU mask to g_a
        l_top = NL - 2;
        gi = 1;                                                            // G[1] (ba) is free again once M is formed: stream order
    } else {
    hipLaunchKernelGGL(k_lt_point_norm, dim3((unsigned)ceil_div(c, 4)), dim3(256), 0, st, c, E, hle, W, nrm);
    NRF_LAUNCH_CHECK();
    // ---- per ray: weights, RenderCLIPEmbedding, their backward ----
    const size_t lds = ((size_t)7 * s + (size_t)2 * E) * sizeof(float);
    hipLaunchKernelGGL(k_lt_ray, dim3((unsigned)rays), dim3(256), lds, st, s, E, h33, W, keep, z, dirs, d_stride, noise, noise_std, hle, W, (const float *)nrm, g_rendered, g33, W,
                       rendered, weights);
    NRF_LAUNCH_CHECK();
    }
    // ---- LE net backward (last layer first)                                                        LeRF.cpp:97-103 ----
    // (bf16x3 products: the ReLU mask of the NEXT stage is applied by the back-propagation product's epilogue -- `premasked` -- instead of by a pass of its own)
    const bool fuse = run_backprop_fuses_mask(m, c);
    bool premasked = gram_masked;
    for (int l = l_top; l >= nl; l--) {
        const LinearLayer &L = m->layers[l];
        if (l != NL - 1 && !premasked) NRF_TRY(run_relu_mask(c, L.out, const_cast<float *>(g.p), g.stride, H[l], W, st));
        const bool first = (l == nl);
        if (first && c >= 4096 && train_gemm_for(m) != 0 && L.out >= 32 && W >= ((yo + 1 + geo + 3) & ~3)) {
            // cat[geo, in]: both column segments in ONE pass over g (bf16x3 TN product).  The geo columns: as they are where they start on a 16-byte boundary (yo == 3);
            // otherwise as the aligned read [sigma | geo | up to 3 more floats of the row] whose first column and padding are dropped in the slice sum
            if (yo == 3 && (geo & 3) == 0) NRF_TRY(gemm_tn_bf16x3_2(c, g, xin, geo, sgeo, 0, 0, geo, L.out, L.in, g_params + L.w_off, st));
            else NRF_TRY(gemm_tn_bf16x3_2(c, g, xin, geo, Seg{h33, W, 0, (1 + geo + 3) & ~3}, 0, 1, geo, L.out, L.in, g_params + L.w_off, st));
        }
        else
        NRF_TRY(run_grad_w_fast(c, g, first ? sgeo : Seg{H[l - 1], W, 0, L.in}, first ? xin : none, L.out, L.in, g_params + L.w_off, st, train_gemm_for(m)));
        float *dst = G[gi]; gi = gi == 3 ? 1 : gi + 1;
        premasked = fuse && !first;                          // dst = d / d H[l - 1], a ReLU output
        NRF_TRY(run_backprop_fast(c, g, m, L, dst, W, st, premasked ? H[l - 1] : nullptr, W));
        g = Seg{dst, W, 0, L.in};
    }
    // g = d / d cat[geo, in]: the sigma net's output gradient = (g_sigma [already in g33 column 0], g_geo)
    const float *g_x0 = g.p;
    hipLaunchKernelGGL(k_lt_copy_cols, dim3((unsigned)ceil_div(c * geo, 256)), dim3(256), 0, st, c, geo, g_x0, W, 0, (const float *)nullptr, 0, 0, g33, W, 1);
    NRF_LAUNCH_CHECK();
    g = Seg{g33, W, 0, 1 + geo};
    // the gradient buffers still free: not g33 (G[0]), not the one holding g_x0
    float *freeb[2]; int nf = 0;
    for (int q = 1; q < 4 && nf < 2; q++) if (G[q] != g_x0) freeb[nf++] = G[q];
    int fi = 0;
    bool emb_done = false;
    premasked = false;
    for (int l = nl - 1; l >= 0; l--) {
        const LinearLayer &L = m->layers[l];
        if (l != nl - 1 && !premasked) NRF_TRY(run_relu_mask(c, L.out, const_cast<float *>(g.p), g.stride, H[l], W, st));
        NRF_TRY(run_grad_w_fast(c, g, l == 0 ? xin : Seg{H[l - 1], W, 0, L.in}, none, L.out, L.in, g_params + L.w_off, st, train_gemm_for(m)));
        if (l == 0 && !g_emb) break;
        float *dst = freeb[fi]; fi ^= 1;
        premasked = fuse && l >= 1;                          // dst = d / d H[l - 1]
        if (l == 0 && fuse) {
            // d / d emb = through the sigma net + through the LE net's cat[geo, in]: the second path's gradient is added by the product's write-out, straight into g_emb
            NRF_TRY(run_backprop_fast(c, g, m, L, g_emb, in, st, nullptr, 0, g_x0 + geo, W));
            emb_done = true;
            break;
        }
        NRF_TRY(run_backprop_fast(c, g, m, L, dst, W, st, premasked ? H[l - 1] : nullptr, W));
        g = Seg{dst, W, 0, L.in};
    }
    if (g_emb && !emb_done) {                                              // (fp32 products: the sum as a pass of its own)
        hipLaunchKernelGGL(k_lt_copy_cols, dim3((unsigned)ceil_div(c * in, 256)), dim3(256), 0, st, c, in, g.p, g.stride, 0, g_x0, W, geo, g_emb, in, 0);
        NRF_LAUNCH_CHECK();
    }
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_huber_rows_nanmean(const float *d_pred, const float *d_target, int64_t n, int e, float delta, float *d_loss, float *d_grad, void *stream)
{
    NRF_CHECK_ARG(d_pred && d_target && d_loss && n > 0 && e > 0 && delta > 0.0f, "nrf_huber_rows_nanmean: bad argument");
    hipStream_t st = as_stream(stream);
    float *tmp = nullptr;
    NRF_HIP(scratch_take(reinterpret_cast<void **>(&tmp), (size_t)(n + 1) * sizeof(float), st));
    int rc = NRF_OK;
    hipLaunchKernelGGL(k_huber_rows, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, n, e, delta, d_pred, d_target, tmp);
    hipLaunchKernelGGL(k_nanmean, dim3(1), dim3(256), 0, st, n, (const float *)tmp, d_loss, tmp + n);
    if (d_grad) hipLaunchKernelGGL(k_huber_rows_grad, dim3((unsigned)ceil_div(n * e, 256)), dim3(256), 0, st, n, e, delta, d_pred, d_target, (const float *)tmp, (const float *)(tmp + n), d_grad);
    if (hipGetLastError() != hipSuccess) { set_error("nrf_huber_rows_nanmean: launch failed"); rc = NRF_ERR_HIP; }
    (void)scratch_give(tmp, st);
    return rc;
}

size_t nrf_lerf_head_backward_workspace_bytes(const nrf_mlp *lerf, int64_t n, int s)
{
    return (lerf && lerf->family == MLP_LERF && s >= 1) ? head_ws_bytes(lerf, n, s) : 0;
}

int nrf_lerf_head_backward(const nrf_mlp *lerf, const float *d_emb, const uint8_t *d_keep, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s,
                           const float *d_noise, float noise_std, const float *d_g_rendered, float *d_g_params, float *d_g_emb, float *d_rendered, float *d_weights,
                           void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(lerf && d_emb && d_z && d_dirs && d_g_rendered && d_g_params && n >= 0 && s >= 1 && d_stride >= 3, "nrf_lerf_head_backward: bad argument");
    if (lerf->family != MLP_LERF) { set_error("nrf_lerf_head_backward: the handle is not a LeRF head (nrf_mlp_lerf_create)"); return NRF_ERR_INVALID_ARG; }
    const int E = lerf->small.hidden_dim_color;
    if (((size_t)7 * s + (size_t)2 * E) * sizeof(float) > 60 * 1024) { set_error("nrf_lerf_head_backward: %d samples x %d embedding dims exceed the per-ray LDS image", s, E); return NRF_ERR_UNSUPPORTED; }
    if (workspace_bytes < head_ws_bytes(lerf, n, s)) { set_error("nrf_lerf_head_backward: workspace %zu < %zu bytes", workspace_bytes, head_ws_bytes(lerf, n, s)); return NRF_ERR_WORKSPACE; }
    if (n == 0) return NRF_OK;
    hipStream_t st = as_stream(stream);
    const int64_t cpts = lt_chunk_pts(lerf, s);
    const int64_t rays_per = lt_rays_per_pass(n, cpts, s);
    const int64_t cmax = (n < rays_per ? n : rays_per) * s;
    const size_t buf = align_up((size_t)cmax * lt_width(lerf, s) * sizeof(float), 256) / sizeof(float);
    float *base = reinterpret_cast<float *>(d_workspace);
    float *nrm = base + buf * (lerf->layers.size() + 4);
    const int in = lerf->small.input_ch;
    // Gram form of the last layer: G = W^T W once per call (the weights change every step), stream-ordered scratch
    float *gram = nullptr;
    if (lerf_train_gram(lerf, s)) {
        const LinearLayer &L1 = lerf->layers.back();
        const int64_t rmax = n < rays_per ? n : rays_per;          // G, M, then the per-ray rows u, v, g_v, q of one chunk
        NRF_HIP(scratch_take(reinterpret_cast<void **>(&gram), ((size_t)2 * L1.in * L1.in + (size_t)rmax * (2 * L1.in + 2 * L1.out)) * sizeof(float), st));
        const float *wle = lerf->d_params + L1.w_off;
        int rcg;
        if (lerf->lerf_gram_current && lerf->d_lerf_gram && L1.in == 256)
            // the Gram matrix the device packer formed at the last parameter upload (double accumulation, unscaled copy behind the packed one): nothing to compute
            rcg = hipMemcpyAsync(gram, lerf->d_lerf_gram + (size_t)L1.in * L1.in, (size_t)L1.in * L1.in * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess ? NRF_OK : NRF_ERR_HIP;
        else rcg = gemm_rm(st, true, false, L1.in, L1.in, L1.out, 1.0f, wle, L1.in, wle, L1.in, 0.0f, gram, L1.in);
        if (rcg != NRF_OK) { (void)scratch_give(gram, st); return rcg; }
    }
    int rc = NRF_OK;
    for (int64_t r0 = 0; r0 < n && rc == NRF_OK; r0 += rays_per) {
        const int64_t rays = (n - r0) < rays_per ? (n - r0) : rays_per;
        const int64_t p0 = r0 * s;
        rc = head_backward_chunk(lerf, d_emb + p0 * in, d_keep ? d_keep + p0 : nullptr, d_z + p0, d_dirs + r0 * d_stride, d_stride, d_noise ? d_noise + p0 : nullptr, noise_std,
                                 rays, s, d_g_rendered + r0 * E, d_g_params, d_g_emb ? d_g_emb + p0 * in : nullptr, d_rendered ? d_rendered + r0 * E : nullptr,
                                 d_weights ? d_weights + p0 : nullptr, base, buf, nrm, gram, st);
    }
    if (gram) (void)scratch_give(gram, st);
    return rc;
}

// ... with the language grid in front: pts [n, s, 3] -> nrf_hash_encode (fp32 rows) -> the head's backward -> nrf_hash_backward_rays, chunk by chunk of whole rays
size_t nrf_lerf_backward_points_workspace_bytes(const nrf_lerf_renderer *r, int64_t n, int s)
{
    if (!r || s < 1) return 0;
    const nrf_mlp *m = nrf_lerf_renderer_head(r);
    const nrf_hash *h = nrf_lerf_renderer_lang_embed(r);
    const int64_t cpts = lt_chunk_pts(m, s);
    const int64_t rays_per = lt_rays_per_pass(n, cpts, s);
    const int64_t c = (n < rays_per ? (n < 1 ? 1 : n) : rays_per) * s;
    const int in = nrf_hash_output_dims(h);
    return head_ws_bytes(m, n, s) + 2 * align_up((size_t)c * in * sizeof(float), 256) + align_up((size_t)c, 256) + 1024;
}

static int lerf_backward_points_impl(const nrf_lerf_renderer *r, const void *d_feats_lm, int64_t cols, const uint8_t *d_keep_cols, const int32_t *d_src, const float *d_pts, const float *d_z,
                                     const float *d_dirs, int d_stride, int64_t n, int s, const float *d_noise, float noise_std, const float *d_g_rendered, float *d_g_lerf_params,
                                     float *d_g_table, void *d_workspace, size_t workspace_bytes, void *stream);

int nrf_lerf_backward_points(const nrf_lerf_renderer *r, const float *d_pts, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, const float *d_noise,
                             float noise_std, const float *d_g_rendered, float *d_g_lerf_params, float *d_g_table, void *d_workspace, size_t workspace_bytes, void *stream)
{
    return lerf_backward_points_impl(r, nullptr, 0, nullptr, nullptr, d_pts, d_z, d_dirs, d_stride, n, s, d_noise, noise_std, d_g_rendered, d_g_lerf_params, d_g_table, d_workspace,
                                     workspace_bytes, stream);
}

int nrf_lerf_backward_points_src(const nrf_lerf_renderer *r, const void *d_feats_lm, int64_t cols, const uint8_t *d_keep_cols, const int32_t *d_src, const float *d_pts, const float *d_z,
                                 const float *d_dirs, int d_stride, int64_t n, int s, const float *d_noise, float noise_std, const float *d_g_rendered, float *d_g_lerf_params,
                                 float *d_g_table, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(d_feats_lm && d_keep_cols && d_src && cols >= n * (int64_t)s, "nrf_lerf_backward_points_src: the feature view (table, keep mask by column, merge map) is incomplete");
    return lerf_backward_points_impl(r, d_feats_lm, cols, d_keep_cols, d_src, d_pts, d_z, d_dirs, d_stride, n, s, d_noise, noise_std, d_g_rendered, d_g_lerf_params, d_g_table, d_workspace,
                                     workspace_bytes, stream);
}

static int lerf_backward_points_impl(const nrf_lerf_renderer *r, const void *d_feats_lm, int64_t cols, const uint8_t *d_keep_cols, const int32_t *d_src, const float *d_pts, const float *d_z,
                                     const float *d_dirs, int d_stride, int64_t n, int s, const float *d_noise, float noise_std, const float *d_g_rendered, float *d_g_lerf_params,
                                     float *d_g_table, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(r && d_pts && d_z && d_dirs && d_g_rendered && d_g_lerf_params && d_g_table && n >= 0 && s >= 1, "nrf_lerf_backward_points: bad argument");
    const nrf_mlp *m = nrf_lerf_renderer_head(r);
    const nrf_hash *h = nrf_lerf_renderer_lang_embed(r);
    const int in = nrf_hash_output_dims(h), E = m->small.hidden_dim_color;
    if (in != m->small.input_ch) { set_error("nrf_lerf_backward_points: the grid yields %d features, the head expects %d", in, m->small.input_ch); return NRF_ERR_INVALID_ARG; }
    if (workspace_bytes < nrf_lerf_backward_points_workspace_bytes(r, n, s)) { set_error("nrf_lerf_backward_points: workspace %zu < %zu bytes", workspace_bytes, nrf_lerf_backward_points_workspace_bytes(r, n, s)); return NRF_ERR_WORKSPACE; }
    if (n == 0) return NRF_OK;
    const int64_t cpts = lt_chunk_pts(m, s);
    const int64_t rays_per = lt_rays_per_pass(n, cpts, s);
    const int64_t cmax = (n < rays_per ? n : rays_per) * s;
    char *ws = static_cast<char *>(d_workspace);
    const size_t hb = head_ws_bytes(m, n, s), eb = align_up((size_t)cmax * in * sizeof(float), 256);
    float *emb = reinterpret_cast<float *>(ws + align_up(hb, 256)), *g_emb = reinterpret_cast<float *>(ws + align_up(hb, 256) + eb);
    uint8_t *keep = reinterpret_cast<uint8_t *>(ws + align_up(hb, 256) + 2 * eb);
    for (int64_t r0 = 0; r0 < n; r0 += rays_per) {
        const int64_t rays = (n - r0) < rays_per ? (n - r0) : rays_per;
        const int64_t p0 = r0 * s, c = rays * s;
        if (d_feats_lm) {          // the rows the forward render has just encoded, read through its merge map (fp16 values, as nrf_hash_encode's rows hold)
            if (in != 128) { set_error("nrf_lerf_backward_points_src: the feature view is the L16 F8 grid's"); return NRF_ERR_INVALID_ARG; }
            hipLaunchKernelGGL(nrf::k_lt_gather_feats, dim3((unsigned)nrf::ceil_div(c * 16, (int64_t)256)), dim3(256), 0, nrf::as_stream(stream), c, static_cast<const uint4 *>(d_feats_lm), cols,
                               d_src + p0, d_keep_cols, emb, keep);
            NRF_LAUNCH_CHECK();
        } else
        NRF_TRY(nrf_hash_encode(h, d_pts + p0 * 3, c, emb, keep, stream));                                         // lang_embed_fn->forward (LeRFRenderer.cpp:34)
        NRF_TRY(nrf_lerf_head_backward(m, emb, keep, d_z + p0, d_dirs + r0 * d_stride, d_stride, rays, s, d_noise ? d_noise + p0 : nullptr, noise_std, d_g_rendered + r0 * E,
                                       d_g_lerf_params, g_emb, nullptr, nullptr, d_workspace, hb, stream));
        NRF_TRY(nrf_hash_backward_rays(h, d_pts + p0 * 3, rays, s, g_emb, d_g_table, stream));                        // CuHashEmbedderBackwardKernel's gradient (CuHashEmbedder.cu:105-216)
    }
    return NRF_OK;
}

// Debug / test entry (not part of the public header): the last layer of the head's backward in its Gram form (1, the default) or layer-wise (0); returns the previous setting
NRF_API int nrf_dbg_lerf_train_gram(int on)
{
    int prev = g_lerf_train_gram.load();
    if (prev < 0) prev = !(getenv("NRF_LERF_TRAIN_GRAM") && getenv("NRF_LERF_TRAIN_GRAM")[0] == '0');
    g_lerf_train_gram.store(on ? 1 : 0);
    return prev;
}

}  // extern "C"
