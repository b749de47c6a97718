// composite.hip -- per-ray stages: sigma->alpha compositing and hierarchical (inverse-CDF) sampling.
//   RawToOutputs   NeRFRenderer.h:199-282 (+ TruncExp::forward, CustomOps.cpp:5-9)
//   SamplePDF      Sampler.h:6-43 ;  z_mid / sort(cat(z, samples))  NeRFRenderer.h:427-431
//
// One 64-lane wavefront per ray, lane = sample.  The two prefix sums the reference computes with
// torch::cumsum (log-transmittance, CDF) are wave scans in DOUBLE: ATen's CPU cumsum accumulates fp32 in double
// and rounds every prefix to fp32, and a double scan's reassociation error (1e-16) disappears in that rounding,
// so the scan reproduces the sequential result.  exp/log/sigmoid come from include/nrf_math.h (the same bits on
// the CPU oracle and here), which makes the whole stage -- and with it the fine-pass sample set -- reproducible.  The pdf normaliser sum(w) is evaluated in ATen's own lane
// order (sum_vec) because it feeds searchsorted: the sample INDICES are bit-exact against the oracle.
#include "common.h"
#include "stoch.h"

#include <type_traits>

namespace nrf {

constexpr int MAX_S = 256;        // samples per ray per pass (coarse) / importance samples
constexpr int RAYS_PER_BLOCK = 4; // waves per block

template <class T>
__device__ __forceinline__ T wave_incl_scan_t(T v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

template <class T>
__device__ __forceinline__ T wave_sum_t(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ double wave_incl_scan(double v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------
// C1  RawToOutputs
// ------------------------------------------------------------------------------------------------
// FAST (the matrix-core precisions of the fused renderer, whose network outputs carry fp16-class error anyway): fp32 scan and sums, hardware exp / log /
// reciprocal (v_exp_f32, v_log_f32, v_rcp_f32) instead of the ATen-reproducing routines of nrf_math.h and the double scan -- the exact form is ~180 vector
// instructions per sample, this one ~25.  !FAST is the parity path and every stand-alone nrf_raw2outputs* entry.
template <bool FAST>
__global__ void __launch_bounds__(64 * RAYS_PER_BLOCK)
k_raw2outputs(int64_t n, int s, int c, int sigma_ch, int white, const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dirs,
              int d_stride, float *__restrict__ rgb, float *__restrict__ disp, float *__restrict__ acc, float *__restrict__ weights,
              float *__restrict__ depth, SigmaNoise nz, const int32_t *__restrict__ src, const float *__restrict__ raw2, int64_t n_split, uint32_t *__restrict__ flag)
{
    // flag (optional): *flag |= 1 when any network output this launch reads is an inf or a NaN -- the sign that a split-precision / fp16 activation left the fp16 range
    // further up (mlp_small_mfma.hip; an inf that meets a ReLU or a zero weight can vanish, one that reaches an output layer cannot).  Four FMAs per sample: x * 0 + t
    // is t for every finite x and NaN otherwise (relu(sigma) below would swallow a NaN sigma silently: NaN > 0 is false).
    float bad_acc = 0.0f;
    // src: sample i's network output is row src[i] -- of raw when src[i] < n_split, else row src[i] - n_split of raw2 (the renderer's fine passes keep the outputs of the
    // coarse depths and of the new samples where they were computed: a ray's samples are two contiguous runs of rows, merged by depth)
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float *dv = dirs + ray * d_stride;
    float nn = dv[0] * dv[0]; nn = nn + dv[1] * dv[1]; nn = nn + dv[2] * dv[2];
    const float nrm = sqrtf(nn);                                   // torch::norm(rays_d, 2, -1)
    const float *zr = z + ray * s;
    using acc_t = typename std::conditional<FAST, float, double>::type;
    acc_t carry = 0;                                               // sum of log(1-alpha) over previous 64-sample blocks
    acc_t sr = 0, sg = 0, sb = 0, sw = 0, swz = 0;
    // FAST, s <= 256 (the renderer's fine pass: 192 merged samples): the wave's inputs of ALL its 64-sample blocks are requested before the first is used -- merge-map
    // entries, then the rows they name, the depths and their right neighbours.  Block by block the kernel was a chain of dependent loads per block (map -> row) with one
    // ray per wave to hide it behind; the arithmetic and its order are unchanged.
    constexpr int PRE = FAST ? 4 : 0;
    const bool pre = FAST && s <= 64 * PRE && c == 4 && ((reinterpret_cast<uintptr_t>(raw) & 15) == 0) && (!raw2 || (reinterpret_cast<uintptr_t>(raw2) & 15) == 0);
    float4 pr4[PRE > 0 ? PRE : 1];
    float pz[PRE > 0 ? PRE : 1], pzn[PRE > 0 ? PRE : 1];
    if constexpr (FAST) {
        if (pre) {
            int64_t prow[PRE];
#pragma unroll
            for (int b = 0; b < PRE; b++) {
                const int j = b * 64 + lane;
                prow[b] = ray * s + (j < s ? j : s - 1);
                if (src && b * 64 < s) prow[b] = src[prow[b]];
            }
#pragma unroll
            for (int b = 0; b < PRE; b++) {
                if (b * 64 >= s) continue;                       // wave-uniform
                const int j = b * 64 + lane, jc = j < s ? j : s - 1;
                const float *r = raw + prow[b] * 4;
                if (src) r = prow[b] < n_split ? raw + prow[b] * 4 : raw2 + (prow[b] - n_split) * 4;
                pr4[b] = *reinterpret_cast<const float4 *>(r);
                pz[b] = zr[jc];
                pzn[b] = zr[jc + 1 < s ? jc + 1 : jc];
            }
        }
    }
    for (int base = 0; base < s; base += 64) {
        const int j = base + lane;
        const bool live = j < s;
        float w = 0.0f, zj = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, lg = 0.0f;
        float alpha = 0.0f;
        if (live) {
            float4 r4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
            bool vec;
            const float *r = nullptr;
            float dist;
            if (FAST && pre) {
                const int b = base >> 6;
                // a compile-time index per block (the arrays stay in registers)
                r4 = b == 0 ? pr4[0] : b == 1 ? pr4[PRE > 1 ? 1 : 0] : b == 2 ? pr4[PRE > 2 ? 2 : 0] : pr4[PRE > 3 ? 3 : 0];
                zj = b == 0 ? pz[0] : b == 1 ? pz[PRE > 1 ? 1 : 0] : b == 2 ? pz[PRE > 2 ? 2 : 0] : pz[PRE > 3 ? 3 : 0];
                const float zn = b == 0 ? pzn[0] : b == 1 ? pzn[PRE > 1 ? 1 : 0] : b == 2 ? pzn[PRE > 2 ? 2 : 0] : pzn[PRE > 3 ? 3 : 0];
                dist = (j + 1 < s) ? (zn - zj) : 1e10f;
                vec = true;
            } else {
                r = raw + (ray * s + j) * c;
                if (src) {
                    const int64_t row = src[ray * s + j];
                    r = row < n_split ? raw + row * c : raw2 + (row - n_split) * c;
                }
                zj = zr[j];
                dist = (j + 1 < s) ? (zr[j + 1] - zj) : 1e10f;   // NeRFRenderer.h:239-240
                // c == 4 (rgb, sigma): one 16-byte load per sample instead of four dword loads
                vec = (c == 4) && ((reinterpret_cast<uintptr_t>(raw) & 15) == 0);
                if (vec) r4 = *reinterpret_cast<const float4 *>(r);
            }
            dist = dist * nrm;                                      // :241
            if (flag) {
                if (vec) { bad_acc = __builtin_fmaf(r4.x, 0.0f, bad_acc); bad_acc = __builtin_fmaf(r4.y, 0.0f, bad_acc); bad_acc = __builtin_fmaf(r4.z, 0.0f, bad_acc); bad_acc = __builtin_fmaf(r4.w, 0.0f, bad_acc); }
                else { bad_acc = __builtin_fmaf(r[sigma_ch], 0.0f, bad_acc); if (rgb) { bad_acc = __builtin_fmaf(r[0], 0.0f, bad_acc); bad_acc = __builtin_fmaf(r[1], 0.0f, bad_acc); bad_acc = __builtin_fmaf(r[2], 0.0f, bad_acc); } }
            }
            float sg = vec ? (sigma_ch == 3 ? r4.w : sigma_ch == 0 ? r4.x : sigma_ch == 1 ? r4.y : r4.z) : r[sigma_ch];
            if (nz.on)                                              // raw_noise_std > 0 (:251-252)
                sg = sg + (nz.arr ? nz.arr[ray * s + j] : nrf_rng_normal(nz.g.seed, nz.stream, (uint64_t)((nz.g.ray_base + ray) * s + j))) * nz.std;
            const float sig = sg > 0.0f ? sg : 0.0f;                // relu
            alpha = -(FAST ? __expf(-sig * dist) : nrf_expf(-sig * dist)) + 1.0f;      // :234
            const float om = 1.0f - alpha;
            lg = FAST ? __logf(om > 1e-10f ? om : 1e-10f) : nrf_logf(om > 1e-10f ? om : 1e-10f);      // :265
            if (rgb) {
                auto sigm = [](float x) { return FAST ? __frcp_rn(1.0f + __expf(-x)) : nrf_sigmoidf(x); };
                cr = sigm(vec ? r4.x : r[0]);                       // sigmoid, :250
                cg = sigm(vec ? r4.y : r[1]);
                cb = sigm(vec ? r4.z : r[2]);
            }
        }
        const acc_t incl = wave_incl_scan_t<acc_t>((acc_t)lg, lane);
        const acc_t excl = carry + (incl - (acc_t)lg);              // exclusive prefix: cat[0, cumsum][:-1] (:263-266)
        carry += __shfl(incl, 63);
        if (live) {
            const float trans = FAST ? __expf((float)excl) : nrf_expf((float)excl);      // TruncExp forward = exp (:267)
            w = alpha * trans;
            if (weights) weights[ray * s + j] = w;
            sr += (acc_t)(w * cr); sg += (acc_t)(w * cg); sb += (acc_t)(w * cb);
            sw += (acc_t)w; swz += (acc_t)(w * zj);
        }
    }
    sr = wave_sum_t<acc_t>(sr); sg = wave_sum_t<acc_t>(sg); sb = wave_sum_t<acc_t>(sb); sw = wave_sum_t<acc_t>(sw); swz = wave_sum_t<acc_t>(swz);
    if (flag) { if (__any(bad_acc != bad_acc) && lane == 0) atomicOr(flag, 1u); }
    if (lane == 0) {
        const float a = (float)sw;
        const float dep = (float)swz / (a > 1e-10f ? a : 1e-10f);  // :272
        float rr = (float)sr, gg = (float)sg, bb = (float)sb;
        if (white) { const float bg = 1.0f - a; rr = rr + bg; gg = gg + bg; bb = bb + bg; }   // :276-277
        if (rgb) { rgb[ray * 3] = rr; rgb[ray * 3 + 1] = gg; rgb[ray * 3 + 2] = bb; }
        if (depth) depth[ray] = dep;
        if (disp) disp[ray] = 1.0f / (dep > 1e-10f ? dep : 1e-10f);                           // :273
        if (acc) acc[ray] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// torch::sum over a contiguous fp32 row, in ATen's CPU order for a `vec`-lane build (see the oracle's
// aten_row_sum_f32 for the derivation from SumKernel.cpp).  x lives in LDS; called by the whole wave,
// result valid in every lane.  vec == 0: double accumulation (order-free definition).
// ------------------------------------------------------------------------------------------------
__device__ float aten_row_sum(const float *x, int n, int vec, float *scratch /*>= 64 floats LDS*/, int lane)
{
    if (vec <= 0) {
        double sacc = 0.0;
        for (int i = lane; i < n; i += 64) sacc += (double)x[i];
        return (float)wave_sum(sacc);
    }
    float result = 0.0f;
    if (n < vec) {
        if (lane == 0) {
            float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int q = n / 4;
            for (int i = 0; i < q; i++)
                for (int k = 0; k < 4; k++) ps[k] += x[i * 4 + k];
            for (int i = q * 4; i < n; i++) ps[0] += x[i];
            for (int k = 1; k < 4; k++) ps[0] += ps[k];
            result = ps[0];
        }
        return __shfl(result, 0);
    }
    // vec <= 16 so the 4 x vec interleaved accumulators fit one wave: lane = k*vec + l
    const int nv = n / vec, q = nv / 4;
    if (lane < 4 * vec) {
        const int k = lane / vec, l = lane - k * vec;
        float p = 0.0f;
        for (int i = 0; i < q; i++) p += x[(i * 4 + k) * vec + l];
        if (k == 0)
            for (int j = q * 4; j < nv; j++) p += x[j * vec + l];
        scratch[lane] = p;
    }
    wave_sync();
    if (lane == 0) {
        float a = 0.0f;
        for (int i = nv * vec; i < n; i++) a += x[i];
        for (int l = 0; l < vec; l++) {
            float p0 = scratch[l];
            for (int k = 1; k < 4; k++) p0 += scratch[k * vec + l];
            a += p0;
        }
        result = a;
    }
    return __shfl(result, 0);
}

// Shared body of SamplePDF for one ray handled by one wave.  bins[nb], wts[nb-1] and the outputs live in LDS.
//   cdf[nb] (out), samples[ns] (out), inds (optional global int64 out)
// u: this ray's draw row (the shared linspace table when det, Sampler.h:21; a row of torch::rand when not, :23), or NULL ->
// generated: element j is nrf_rng_uniform(seed, NRF_RNG_U_PDF, u_base + j)
__device__ void sample_pdf_wave(const float *bins, float *wts, int nb, const float *__restrict__ u, uint64_t u_seed, uint64_t u_base, int ns, int sum_vec,
                                float *cdf, float *samples, float *scratch, int64_t *inds, int lane)
{
    const int nw = nb - 1;
    for (int k = lane; k < nw; k += 64) wts[k] = wts[k] + 1e-8f;                 // Sampler.h:10
    wave_sync();
    const float fsum = aten_row_sum(wts, nw, sum_vec, scratch, lane);            // :11 sum(weights, -1, true)
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.0f;                                                // :13
    for (int base = 0; base < nw; base += 64) {
        const int k = base + lane;
        const float pdf = (k < nw) ? (wts[k] / fsum) : 0.0f;                     // :11
        const double incl = carry + wave_incl_scan((double)pdf, lane);           // :12 cumsum (double accumulate)
        if (k < nw) cdf[k + 1] = (float)incl;
        carry = __shfl(incl, 63);
    }
    wave_sync();
    for (int j = lane; j < ns; j += 64) {
        const float uj = u ? u[j] : nrf_rng_uniform(u_seed, NRF_RNG_U_PDF, u_base + (uint64_t)j);
        int lo = 0, hi = nb;                                                     // searchsorted(cdf, u, right=True) (:28)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
        }
        const int ind = lo;
        const int below = ind - 1 > 0 ? ind - 1 : 0;                             // :29
        const int above = ind < nb - 1 ? ind : nb - 1;                           // :30
        float denom = cdf[above] - cdf[below];                                   // :37
        if (denom < 1e-5f) denom = 1.0f;                                         // :38
        const float t = (uj - cdf[below]) / denom;                               // :39
        samples[j] = bins[below] + t * (bins[above] - bins[below]);              // :40
        if (inds) inds[j] = ind;
    }
    wave_sync();
}

struct PdfLds {
    float bins[MAX_S];
    float wts[MAX_S];
    float cdf[MAX_S + 4];
    float merged[2 * MAX_S];
    float scratch[64];
};

__global__ void __launch_bounds__(64 * RAYS_PER_BLOCK)
k_sample_pdf(int64_t n, int nb, int ns, int sum_vec, const float *__restrict__ bins, const float *__restrict__ weights,
             const float *__restrict__ u, int64_t u_stride, float *__restrict__ samples, int64_t *__restrict__ inds)
{
    __shared__ PdfLds lds[RAYS_PER_BLOCK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= n) return;
    PdfLds &L = lds[wv];
    for (int k = lane; k < nb; k += 64) L.bins[k] = bins[ray * nb + k];
    for (int k = lane; k < nb - 1; k += 64) L.wts[k] = weights[ray * (nb - 1) + k];
    wave_sync();
    sample_pdf_wave(L.bins, L.wts, nb, u + ray * u_stride, 0, 0, ns, sum_vec, L.cdf, L.merged, L.scratch, inds ? inds + ray * ns : nullptr, lane);
    for (int j = lane; j < ns; j += 64) samples[ray * ns + j] = L.merged[j];
}

// NeRFRenderer.h:427-431: z_mid -> SamplePDF(weights[1:-1]) -> sort(cat(z, samples)).
__global__ void __launch_bounds__(64 * RAYS_PER_BLOCK)
k_fine_depths(int64_t n, int s, int ns, int sum_vec, const float *__restrict__ z, const float *__restrict__ weights,
              const float *__restrict__ u, int64_t u_stride, RngRef g, float *__restrict__ zf, int32_t *__restrict__ src, float *__restrict__ z_new)
{
    // src / z_new (optional, the renderer's feature reuse): src[ray][slot] = where slot's depth came from -- ray * s + i for coarse sample i, n * s + ray * ns + j for
    // new sample j (a column of a feature table that holds the coarse pass's columns first, then the new samples'); z_new[ray][j] = the new samples, unsorted
    __shared__ PdfLds lds[RAYS_PER_BLOCK];
    __shared__ float zs[RAYS_PER_BLOCK][MAX_S];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= n) return;
    PdfLds &L = lds[wv];
    float *zl = zs[wv];
    for (int k = lane; k < s; k += 64) zl[k] = z[ray * s + k];
    wave_sync();
    const int nb = s - 1;
    for (int k = lane; k < nb; k += 64) L.bins[k] = 0.5f * (zl[k + 1] + zl[k]);                  // :427
    for (int k = lane; k < nb - 1; k += 64) L.wts[k] = weights[ray * s + 1 + k];                 // weights[..., 1:-1] (:428)
    wave_sync();
    float *smp = L.merged;                  // samples [ns]
    sample_pdf_wave(L.bins, L.wts, nb, u ? u + ray * u_stride : nullptr, g.seed, (uint64_t)((g.ray_base + ray) * ns), ns, sum_vec, L.cdf, smp, L.scratch, nullptr, lane);
    // ---- stable ascending sort of cat(z[0..s), samples[0..ns)) by merge ranks (both runs are sorted unless
    //      fp32 rounding broke monotonicity by an ulp: detected, then ranked by exhaustive counting) ----
    bool bad = false;
    for (int k = lane; k + 1 < s; k += 64) bad |= zl[k] > zl[k + 1];
    for (int k = lane; k + 1 < ns; k += 64) bad |= smp[k] > smp[k + 1];
    float *outp = zf + ray * (s + ns);
    int32_t *srcp = src ? src + ray * (s + ns) : nullptr;
    const int32_t src_z = (int32_t)(ray * s), src_new = (int32_t)(n * s + ray * ns);
    if (z_new) for (int j = lane; j < ns; j += 64) z_new[ray * ns + j] = smp[j];
    if (!__any(bad)) {
        for (int i = lane; i < s; i += 64) {             // rank(z_i) = i + #{samples < z_i}   (z precedes samples on ties)
            const float v = zl[i];
            int lo = 0, hi = ns;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (smp[mid] < v) lo = mid + 1; else hi = mid; }
            outp[i + lo] = v;
            if (srcp) srcp[i + lo] = src_z + i;
        }
        for (int j = lane; j < ns; j += 64) {            // rank(sample_j) = j + #{z <= sample_j}
            const float v = smp[j];
            int lo = 0, hi = s;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (zl[mid] <= v) lo = mid + 1; else hi = mid; }
            outp[j + lo] = v;
            if (srcp) srcp[j + lo] = src_new + j;
        }
    } else {
        const int tot = s + ns;
        for (int i = lane; i < tot; i += 64) {
            const float v = i < s ? zl[i] : smp[i - s];
            int rank = 0;
            for (int k = 0; k < tot; k++) {
                const float o = k < s ? zl[k] : smp[k - s];
                rank += (o < v) || (o == v && k < i);
            }
            outp[rank] = v;
            if (srcp) srcp[rank] = i < s ? src_z + i : src_new + (i - s);
        }
    }
}

// RenderCLIPEmbedding (LeRFRenderer.h:45-54): out = normalize(sum_s w_s * e_s, eps 1e-8).  One workgroup per ray; thread = embedding
// channel(s); the sum over samples and the squared norm accumulate in double (order-free, same definition as the oracle).
__global__ void __launch_bounds__(256) k_clip_embedding(int s, int stride, int dim, const float *__restrict__ e, const float *__restrict__ w, float *__restrict__ out,
                                                        uint32_t *__restrict__ flag)
{
    // flag (optional): *flag |= 1 when the weighted sum holds an inf or a NaN (its squared norm is then not finite): the LeRF pass's non-finite word
    __shared__ double red[4];
    const int64_t ray = blockIdx.x;
    const float *er = e + ray * (int64_t)s * stride;
    const float *wr = w + ray * s;
    double ss = 0.0;
    for (int k = threadIdx.x; k < dim; k += 256) {
        double acc = 0.0;
        for (int j = 0; j < s; j++) acc += (double)(wr[j] * er[(int64_t)j * stride + k]);
        const float v = (float)acc;
        out[ray * dim + k] = v;
        ss += (double)v * (double)v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (flag && threadIdx.x == 0 && !(red[0] + red[1] + red[2] + red[3] <= 1.7e308)) atomicOr(flag, 1u);
    const float nrm = fmaxf((float)sqrt(red[0] + red[1] + red[2] + red[3]), 1e-8f);
    for (int k = threadIdx.x; k < dim; k += 256) out[ray * dim + k] = out[ray * dim + k] / nrm;
}

int launch_clip_embedding(const float *embeds, int embed_stride, int embed_dim, const float *weights, int64_t n, int s, float *out, hipStream_t st, uint32_t *flag)
{
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_clip_embedding, dim3((unsigned)n), dim3(256), 0, st, s, embed_stride, embed_dim, embeds, weights, out, flag);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_raw2outputs(const float *raw, const float *z, const float *dirs, int d_stride, int64_t n, int s, int c, int sigma_ch, int white, float *rgb,
                       float *disp, float *acc, float *weights, float *depth, const SigmaNoise &nz, hipStream_t st, bool fast, const int32_t *src, const float *raw2, int64_t n_split,
                       uint32_t *flag)
{
    if (src && !raw2) { raw2 = raw; n_split = 0; }          // one array of rows
    if (n == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_COMPOSITE, st);
    if (fast) hipLaunchKernelGGL(k_raw2outputs<true>, dim3((unsigned)ceil_div(n, RAYS_PER_BLOCK)), dim3(64 * RAYS_PER_BLOCK), 0, st, n, s, c, sigma_ch, white,
                                 raw, z, dirs, d_stride, rgb, disp, acc, weights, depth, nz, src, raw2, n_split, flag);
    else hipLaunchKernelGGL(k_raw2outputs<false>, dim3((unsigned)ceil_div(n, RAYS_PER_BLOCK)), dim3(64 * RAYS_PER_BLOCK), 0, st, n, s, c, sigma_ch, white,
                       raw, z, dirs, d_stride, rgb, disp, acc, weights, depth, nz, src, raw2, n_split, flag);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// u == NULL: per-ray uniform draws generated from g (SamplePDF det = false with the library's counter RNG)
int launch_fine_depths(const float *z, const float *weights, int64_t n, int s, const float *u, int64_t u_stride, const RngRef &g, int ns, int sum_vec,
                       float *zf, hipStream_t st, int32_t *src, float *z_new)
{
    // s = 2: one bin edge and NO weight (weights[1:-1] is empty): the CDF is the single 0, every u lands past it and every sample is that edge (Sampler.h:28-40 on empty
    // tensors); s = 3: two edges, one weight
    NRF_CHECK_ARG(s >= 2 && s <= MAX_S && ns >= 1 && ns <= MAX_S, "nrf_fine_depths: n_samples %d / n_importance %d outside the built range [2,%d] / [1,%d]", s, ns, MAX_S, MAX_S);
    NRF_CHECK_ARG(sum_vec == 0 || sum_vec == 4 || sum_vec == 8 || sum_vec == 16, "nrf_fine_depths: sum_vec must be 0, 4, 8 or 16");
    if (n == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_SAMPLE, st);
    hipLaunchKernelGGL(k_fine_depths, dim3((unsigned)ceil_div(n, RAYS_PER_BLOCK)), dim3(64 * RAYS_PER_BLOCK), 0, st, n, s, ns, sum_vec, z, weights, u, u_stride, g, zf, src, z_new);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_raw2outputs(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c, int white_bkgr,
                    float *d_rgb, float *d_disp, float *d_acc, float *d_weights, float *d_depth, void *stream)
{
    NRF_CHECK_ARG(d_raw && d_z && d_dirs && n >= 0 && s >= 1 && c >= 4 && d_stride >= 3, "nrf_raw2outputs: bad argument");
    return launch_raw2outputs(d_raw, d_z, d_dirs, d_stride, n, s, c, 3, white_bkgr, d_rgb, d_disp, d_acc, d_weights, d_depth, SigmaNoise{}, as_stream(stream));
}

int nrf_raw2outputs_noise(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c, int white_bkgr,
                          const float *d_noise, float noise_std, float *d_rgb, float *d_disp, float *d_acc, float *d_weights, float *d_depth, void *stream)
{
    NRF_CHECK_ARG(d_raw && d_z && d_dirs && d_noise && n >= 0 && s >= 1 && c >= 4 && d_stride >= 3, "nrf_raw2outputs_noise: bad argument");
    SigmaNoise nz{};
    nz.on = 1; nz.arr = d_noise; nz.std = noise_std;
    return launch_raw2outputs(d_raw, d_z, d_dirs, d_stride, n, s, c, 3, white_bkgr, d_rgb, d_disp, d_acc, d_weights, d_depth, nz, as_stream(stream));
}

int nrf_raw2weights(const float *d_raw, int c, int sigma_ch, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s,
                    float *d_weights, float *d_depth, float *d_disp, float *d_acc, void *stream)
{
    NRF_CHECK_ARG(d_raw && d_z && d_dirs && n >= 0 && s >= 1 && c >= 1 && sigma_ch >= 0 && sigma_ch < c && d_stride >= 3, "nrf_raw2weights: bad argument");
    return launch_raw2outputs(d_raw, d_z, d_dirs, d_stride, n, s, c, sigma_ch, 0, nullptr, d_disp, d_acc, d_weights, d_depth, SigmaNoise{}, as_stream(stream));
}

int nrf_raw2weights_gather(const float *d_raw, int c, int sigma_ch, const int32_t *d_src, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s,
                           float *d_weights, float *d_depth, float *d_disp, float *d_acc, void *stream)
{
    NRF_CHECK_ARG(d_raw && d_src && d_z && d_dirs && n >= 0 && s >= 1 && c >= 1 && sigma_ch >= 0 && sigma_ch < c && d_stride >= 3, "nrf_raw2weights_gather: bad argument");
    return launch_raw2outputs(d_raw, d_z, d_dirs, d_stride, n, s, c, sigma_ch, 0, nullptr, d_disp, d_acc, d_weights, d_depth, SigmaNoise{}, as_stream(stream), false, d_src, nullptr, 0);
}

int nrf_render_clip_embedding(const float *d_embeds, int embed_stride, int embed_dim, const float *d_weights, int64_t n, int s, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_embeds && d_weights && d_out && n >= 0 && s >= 1 && embed_dim >= 1 && embed_stride >= embed_dim, "nrf_render_clip_embedding: bad argument");
    if (n == 0) return NRF_OK;
    return launch_clip_embedding(d_embeds, embed_stride, embed_dim, d_weights, n, s, d_out, as_stream(stream), nullptr);
}

static int sample_pdf_entry(const float *d_bins, const float *d_weights, int64_t n, int nb, const float *d_u, int64_t u_stride, int ns, int sum_vec,
                            float *d_samples, int64_t *d_inds, void *stream)
{
    NRF_CHECK_ARG(d_bins && (d_weights || nb == 1) && d_u && d_samples && n >= 0, "nrf_sample_pdf: bad argument");      // one bin edge: no weight (an empty [n, 0] tensor has no storage)
    NRF_CHECK_ARG(nb >= 1 && nb <= MAX_S && ns >= 1 && ns <= 2 * MAX_S, "nrf_sample_pdf: nb %d / ns %d outside the built range (<= %d bins, <= %d samples)", nb, ns, MAX_S, 2 * MAX_S);
    NRF_CHECK_ARG(sum_vec == 0 || sum_vec == 4 || sum_vec == 8 || sum_vec == 16, "nrf_sample_pdf: sum_vec must be 0, 4, 8 or 16");
    if (n == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_SAMPLE, as_stream(stream));
    hipLaunchKernelGGL(k_sample_pdf, dim3((unsigned)ceil_div(n, RAYS_PER_BLOCK)), dim3(64 * RAYS_PER_BLOCK), 0, as_stream(stream), n, nb, ns, sum_vec,
                       d_bins, d_weights, d_u, u_stride, d_samples, d_inds);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_sample_pdf(const float *d_bins, const float *d_weights, int64_t n, int nb, const float *d_u, int ns, int sum_vec,
                   float *d_samples, int64_t *d_inds, void *stream)
{
    return sample_pdf_entry(d_bins, d_weights, n, nb, d_u, 0, ns, sum_vec, d_samples, d_inds, stream);
}

int nrf_sample_pdf_rand(const float *d_bins, const float *d_weights, int64_t n, int nb, const float *d_u, int ns, int sum_vec,
                        float *d_samples, int64_t *d_inds, void *stream)
{
    return sample_pdf_entry(d_bins, d_weights, n, nb, d_u, ns, ns, sum_vec, d_samples, d_inds, stream);
}

int nrf_fine_depths(const float *d_z, const float *d_weights, int64_t n, int s, const float *d_u, int ns, int sum_vec, float *d_z_fine, void *stream)
{
    NRF_CHECK_ARG(d_z && d_weights && d_u && d_z_fine && n >= 0, "nrf_fine_depths: bad argument");
    return launch_fine_depths(d_z, d_weights, n, s, d_u, 0, RngRef{0, 0}, ns, sum_vec, d_z_fine, as_stream(stream));
}

int nrf_fine_depths_merge(const float *d_z, const float *d_weights, int64_t n, int s, const float *d_u, int ns, int sum_vec, float *d_z_fine, int32_t *d_src,
                          float *d_z_new, void *stream)
{
    NRF_CHECK_ARG(d_z && d_weights && d_u && d_z_fine && d_src && d_z_new && n >= 0, "nrf_fine_depths_merge: bad argument");
    NRF_CHECK_ARG(n * (int64_t)(s + ns) < ((int64_t)1 << 31), "nrf_fine_depths_merge: n (s + ns) must fit the int32 column map");
    return launch_fine_depths(d_z, d_weights, n, s, d_u, 0, RngRef{0, 0}, ns, sum_vec, d_z_fine, as_stream(stream), d_src, d_z_new);
}

int nrf_fine_depths_rand(const float *d_z, const float *d_weights, int64_t n, int s, const float *d_u, int ns, int sum_vec, float *d_z_fine, void *stream)
{
    NRF_CHECK_ARG(d_z && d_weights && d_u && d_z_fine && n >= 0, "nrf_fine_depths_rand: bad argument");
    return launch_fine_depths(d_z, d_weights, n, s, d_u, ns, RngRef{0, 0}, ns, sum_vec, d_z_fine, as_stream(stream));
}

}  // extern "C"
