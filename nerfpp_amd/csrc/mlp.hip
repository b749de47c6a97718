// mlp.hip -- the radiance-field MLPs behind BaseNeRFImpl::forward (NeRF.h:33-42):
//   NeRFSmallImpl  NeRF.cpp:322-412      NeRFImpl  NeRF.cpp:41-126      LeRFImpl  LeRF.cpp:28-111
//
// This file holds the handle (parameter blob in the reference's checkpoint order) and the NRF_PREC_F32 path:
// a generic fused Linear(+bias)(+ReLU) kernel whose dot product is an fp32 FMA chain in ascending k starting
// from 0 -- the exact order the oracle restates, so NRF_PREC_F32 output equals the oracle bit for bit.
// The matrix-core paths (NRF_PREC_F16_MFMA) are in mlp_small_mfma.hip / mlp_nerf_mfma.hip.
#include "mlp.h"

#include <thread>

namespace nrf {

constexpr int LIN_TP = 16;   // points per block

// y[pt][y_off + o] = act( sum_k W[o][k] * concat(a,b)[pt][k] + bias[o] ),  W given transposed ([in][out]).
__global__ void __launch_bounds__(256) k_linear(int64_t npts, Seg a, Seg b, const float *__restrict__ wt, const float *__restrict__ bias,
                                                int out, int relu, float *__restrict__ y, int y_stride, int y_off)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [in][LIN_TP]
    const int in = a.n + b.n;
    const int64_t p0 = (int64_t)blockIdx.x * LIN_TP;
    for (int e = threadIdx.x; e < in * LIN_TP; e += blockDim.x) {
        const int j = e / in, k = e - j * in;                     // consecutive threads walk a row: coalesced
        const int64_t pt = p0 + j;
        float v = 0.0f;
        if (pt < npts) v = (k < a.n) ? a.p[pt * a.stride + a.off + k] : b.p[pt * b.stride + b.off + (k - a.n)];
        xs[k * LIN_TP + j] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < out; o += blockDim.x) {
        float acc[LIN_TP];
#pragma unroll
        for (int j = 0; j < LIN_TP; j++) acc[j] = 0.0f;
        // k ascending, one FMA per (k, point): the order the oracle restates.  Eight k-steps per trip so that their weight loads (one
        // 4-byte global load per thread and k, the only vector-memory traffic of the loop) are in flight together instead of one at a time.
        auto kstep = [&](int k, float w) {
            const float4 *xv = reinterpret_cast<const float4 *>(xs + k * LIN_TP);
#pragma unroll
            for (int q = 0; q < LIN_TP / 4; q++) {
                const float4 v = xv[q];
                acc[q * 4 + 0] = __builtin_fmaf(w, v.x, acc[q * 4 + 0]);
                acc[q * 4 + 1] = __builtin_fmaf(w, v.y, acc[q * 4 + 1]);
                acc[q * 4 + 2] = __builtin_fmaf(w, v.z, acc[q * 4 + 2]);
                acc[q * 4 + 3] = __builtin_fmaf(w, v.w, acc[q * 4 + 3]);
            }
        };
        int k = 0;
        for (; k + 8 <= in; k += 8) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w8[u] = wt[(size_t)(k + u) * out + o];
#pragma unroll
            for (int u = 0; u < 8; u++) kstep(k + u, w8[u]);
        }
        for (; k < in; k++) kstep(k, wt[(size_t)k * out + o]);
        const float bo = bias ? bias[o] : 0.0f;
#pragma unroll
        for (int j = 0; j < LIN_TP; j++) {
            const int64_t pt = p0 + j;
            if (pt < npts) {
                float v = acc[j];
                if (bias) v = v + bo;
                if (relu) v = v < 0.0f ? 0.0f : v;
                y[pt * y_stride + y_off + o] = v;
            }
        }
    }
}

// LeRFImpl: le = normalize(h, eps=1e-8) = h / max(||h||, eps); one wave per point.
__global__ void k_l2_normalize(int64_t npts, int n, const float *__restrict__ x, int x_stride, float *__restrict__ y, int y_stride)
{
    const int64_t pt = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (pt >= npts) return;
    const int lane = threadIdx.x & 63;
    double ss = 0.0;
    for (int k = lane; k < n; k += 64) { const float v = x[pt * x_stride + k]; ss += (double)v * (double)v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float nrm = fmaxf((float)sqrt(ss), 1e-8f);
    for (int k = lane; k < n; k += 64) y[pt * y_stride + k] = x[pt * x_stride + k] / nrm;
}

__global__ void k_copy_col(int64_t npts, const float *__restrict__ x, int x_stride, int x_col, float *__restrict__ y, int y_stride, int y_col)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npts) y[i * y_stride + y_col] = x[i * x_stride + x_col];
}

int run_linear(int64_t npts, Seg a, Seg b, const LinearLayer &L, int relu, float *y, int y_stride, int y_off, hipStream_t st)
{
    const int in = a.n + b.n;
    if (in != L.in) { set_error("internal: linear layer expects %d inputs, got %d", L.in, in); return NRF_ERR_INVALID_ARG; }
    const int threads = L.out >= 256 ? 256 : (int)ceil_div(L.out, 64) * 64;
    const size_t lds = (size_t)in * LIN_TP * sizeof(float);
    hipLaunchKernelGGL(k_linear, dim3((unsigned)ceil_div(npts, LIN_TP)), dim3(threads), lds, st, npts, a, b, L.d_wt, L.d_bias, L.out, relu, y, y_stride, y_off);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

static const int64_t F32_CHUNK = 1 << 18;   // points per pass of the generic path (bounds the scratch)

size_t mlp_workspace_bytes(const nrf_mlp *m, int64_t p, int prec)
{
    if (prec == NRF_PREC_F32) {
        const int64_t c = p < F32_CHUNK ? p : F32_CHUNK;
        return align_up((size_t)c * m->max_width * sizeof(float), 256) * 3;
    }
    return 256;
}

static int forward_f32(const nrf_mlp *m, const float *x, int xs, int64_t p, float *out, int os, float *ws, hipStream_t st)
{
    const size_t buf_elems = align_up((size_t)(p < F32_CHUNK ? p : F32_CHUNK) * m->max_width * sizeof(float), 256) / sizeof(float);
    float *A = ws, *B = ws + buf_elems, *C = ws + 2 * buf_elems;
    const int W = m->max_width;
    const Seg none{nullptr, 0, 0, 0};
    if (m->family == MLP_SMALL) {
        const auto &d = m->small;
        // sigma net (NeRF.cpp:372-381)
        Seg cur{x, xs, 0, d.input_ch};
        float *bufs[2] = {A, B};
        int li = 0;
        for (int l = 0; l < d.num_layers; l++, li++) {
            NRF_TRY(run_linear(p, cur, none, m->layers[li], l != d.num_layers - 1, bufs[l & 1], W, 0, st));
            cur = Seg{bufs[l & 1], W, 0, m->layers[li].out};
        }
        const float *sig = cur.p;   // column 0 = sigma, 1.. = geo features
        // colour net on cat[views, geo] (NeRF.cpp:384-391)
        Seg cv{x, xs, d.input_ch, d.input_ch_views};
        Seg cg{sig, W, 1, d.geo_feat_dim};
        float *cb[2] = {C, (sig == A) ? B : A};
        for (int l = 0; l < d.num_layers_color; l++, li++) {
            const bool last = (l == d.num_layers_color - 1);
            float *dst = last ? out : cb[l & 1];
            NRF_TRY(run_linear(p, l == 0 ? cv : cur, l == 0 ? cg : none, m->layers[li], !last, dst, last ? os : W, 0, st));
            cur = Seg{dst, last ? os : W, 0, m->layers[li].out};
        }
        // out = cat[colour, sigma] (NeRF.cpp:408)
        hipLaunchKernelGGL(k_copy_col, dim3((unsigned)ceil_div(p, 256)), dim3(256), 0, st, p, sig, W, 0, out, os, 3);
        NRF_LAUNCH_CHECK();
        if (d.use_pred_normal) {
            // predicted normals: cat[sigma, geo_feat, input_pts] -> ... -> 3, no final activation (NeRF.cpp:393-407); out = cat[colour, sigma, normals]
            // `sig` (the sigma net's last output: column 0 = sigma, 1.. = geo) is one of A / B; the colour chain may have reused the other and C: two scratch rows remain
            // free only if we are careful -- the head runs AFTER the colour net, out of buffers that no longer hold anything needed: every one except `sig`'s
            float *nb[2] = {(sig == A) ? B : A, C};
            Seg sg{sig, W, 0, 1 + d.geo_feat_dim};
            Seg ncur = sg;
            for (int l = 0; l < d.num_layers_normals; l++, li++) {
                const bool last = (l == d.num_layers_normals - 1);
                float *dst = last ? out : nb[l & 1];
                NRF_TRY(run_linear(p, l == 0 ? sg : ncur, l == 0 ? Seg{x, xs, 0, d.input_ch} : none, m->layers[li], !last, dst, last ? os : W, last ? 4 : 0, st));
                ncur = Seg{dst, last ? os : W, 0, m->layers[li].out};
            }
        }
        return NRF_OK;
    }
    if (m->family == MLP_NERF) {
        const auto &d = m->nerf;
        const Seg xin{x, xs, 0, d.input_ch};
        Seg cur = xin;
        bool cat_in = false;
        float *bufs[2] = {A, B};
        int li = 0;
        for (int l = 0; l < d.depth; l++, li++) {
            // after layer `skip`, h = cat[input_pts, h] (NeRF.cpp:103-104)
            NRF_TRY(run_linear(p, cat_in ? xin : cur, cat_in ? cur : none, m->layers[li], 1, bufs[l & 1], W, 0, st));
            cur = Seg{bufs[l & 1], W, 0, d.width};
            cat_in = (l == d.skip);
        }
        float *other = (cur.p == A) ? B : A;
        if (d.use_viewdirs) {
            const LinearLayer &views = m->layers[li], &feat = m->layers[li + 1], &alpha = m->layers[li + 2], &rgb = m->layers[li + 3];
            NRF_TRY(run_linear(p, cur, none, alpha, 0, out, os, 3, st));                                   // alpha -> out[:,3]
            NRF_TRY(run_linear(p, cur, none, feat, 0, other, W, 0, st));                                   // feature (no ReLU)
            NRF_TRY(run_linear(p, Seg{other, W, 0, d.width}, Seg{x, xs, d.input_ch, d.input_ch_views}, views, 1, C, W, 0, st));
            NRF_TRY(run_linear(p, Seg{C, W, 0, d.width / 2}, none, rgb, 0, out, os, 0, st));               // rgb -> out[:,0:3]
        } else {
            // output_linear(cat[h, input_pts]) (NeRF.cpp:121-124); cat_in can only be set if skip == depth-1
            NRF_TRY(run_linear(p, cur, xin, m->layers[li], 0, out, os, 0, st));
        }
        return NRF_OK;
    }
    // MLP_LERF (LeRF.cpp:86-108)
    {
        const auto &d = m->small;   // reuses: input_ch, num_layers, hidden_dim, geo_feat_dim; hidden_dim_color = embed dim
        const Seg xin{x, xs, 0, d.input_ch};
        Seg cur = xin;
        float *bufs[2] = {A, B};
        int li = 0;
        for (int l = 0; l < d.num_layers; l++, li++) {
            NRF_TRY(run_linear(p, cur, none, m->layers[li], l != d.num_layers - 1, bufs[l & 1], W, 0, st));
            cur = Seg{bufs[l & 1], W, 0, m->layers[li].out};
        }
        const float *sig = cur.p;
        float *cb[2] = {C, (sig == A) ? B : A};
        Seg g{sig, W, 1, d.geo_feat_dim};
        for (int l = 0; l < d.num_layers; l++, li++) {
            NRF_TRY(run_linear(p, l == 0 ? g : cur, l == 0 ? xin : none, m->layers[li], l != d.num_layers - 1, cb[l & 1], W, 0, st));
            cur = Seg{cb[l & 1], W, 0, m->layers[li].out};
        }
        const int E = d.hidden_dim_color;
        hipLaunchKernelGGL(k_l2_normalize, dim3((unsigned)ceil_div(p, 4)), dim3(256), 0, st, p, E, cur.p, W, out, os);
        NRF_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_copy_col, dim3((unsigned)ceil_div(p, 256)), dim3(256), 0, st, p, sig, W, 0, out, os, E);
        NRF_LAUNCH_CHECK();
        return NRF_OK;
    }
}

int mlp_small_forward_mfma(const nrf_mlp *m, const float *d_x, int x_stride, int64_t p, int split, float *d_out, int out_stride, hipStream_t st);
int mlp_nerf_forward_mfma(const nrf_mlp *m, const float *d_x, int x_stride, int64_t p, float *d_out, int out_stride, hipStream_t st);

int mlp_forward(const nrf_mlp *m, const float *d_x, int x_stride, int64_t p, int prec, float *d_out, int out_stride,
                void *d_ws, size_t ws_bytes, hipStream_t st)
{
    if (p == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_MLP, st);
    if (prec == NRF_PREC_F32) {
        if (ws_bytes < mlp_workspace_bytes(m, p, prec)) { set_error("mlp_forward: workspace %zu < %zu bytes", ws_bytes, mlp_workspace_bytes(m, p, prec)); return NRF_ERR_WORKSPACE; }
        for (int64_t p0 = 0; p0 < p; p0 += F32_CHUNK) {
            const int64_t c = (p - p0) < F32_CHUNK ? (p - p0) : F32_CHUNK;
            NRF_TRY(forward_f32(m, d_x + p0 * x_stride, x_stride, c, d_out + p0 * out_stride, out_stride, reinterpret_cast<float *>(d_ws), st));
        }
        return NRF_OK;
    }
    if (prec == NRF_PREC_F16_MFMA) {
        if (m->family == MLP_SMALL) return mlp_small_forward_mfma(m, d_x, x_stride, p, 0, d_out, out_stride, st);
        if (m->family == MLP_NERF) return mlp_nerf_forward_mfma(m, d_x, x_stride, p, d_out, out_stride, st);
        set_error("NRF_PREC_F16_MFMA is not built for this MLP family; use NRF_PREC_F32");
        return NRF_ERR_UNSUPPORTED;
    }
    if (prec == NRF_PREC_F16_SPLIT) {
        if (m->family == MLP_SMALL) return mlp_small_forward_mfma(m, d_x, x_stride, p, 1, d_out, out_stride, st);
        if (m->family == MLP_NERF) return mlp_nerf_forward_split_rows(m, d_x, x_stride, p, d_out, out_stride, st);
        set_error("NRF_PREC_F16_SPLIT is built for the NeRFSmall and NeRF families; use NRF_PREC_F32 or NRF_PREC_F16_MFMA");
        return NRF_ERR_UNSUPPORTED;
    }
    set_error("unknown precision %d", prec);
    return NRF_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------------------------------
// N1  backward of NeRFSmallImpl::forward (training step, NeRFExecutor.h:923 loss.backward()).  fp32, layer by layer:
//   forward with every layer output kept  ->  for each layer, last to first:  g *= (out > 0) ;  dW += g^T . in ;  g_in = g . W
// g . W is the same k_linear kernel run on the checkpoint-order blob (W [out][in] row-major IS the transposed operand of the
// backward product); dW is a TN GEMM over the points with one atomic add per output element and workgroup.
// ---------------------------------------------------------------------------------------------------
__global__ void k_relu_mask(int64_t total, int n, float *__restrict__ g, int g_stride, const float *__restrict__ act, int act_stride)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t pt = e / n; const int k = (int)(e - pt * n);
    if (!(act[pt * act_stride + k] > 0.0f)) g[pt * g_stride + k] = 0.0f;
}

int run_relu_mask(int64_t npts, int n, float *g, int g_stride, const float *act, int act_stride, hipStream_t st)
{
    hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)ceil_div(npts * n, 256)), dim3(256), 0, st, npts * n, n, g, g_stride, act, act_stride);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

constexpr int GW_PTS = 32;      // points staged per iteration
// dW[o][i] += sum_pt g[pt][o] * concat(a, b)[pt][i]   (out, in <= 64 per launch tile; 256 threads, a 4x4 register tile each)
__global__ void __launch_bounds__(256) k_grad_w(int64_t npts, Seg g, Seg a, Seg b, int out, int in, float *__restrict__ dw)
{
    __shared__ __attribute__((aligned(16))) float gs[GW_PTS][64 + 4], xs[GW_PTS][64 + 4];      // +4: rows stay 16-byte aligned for b128 reads
    const int o0 = blockIdx.y * 64, i0 = blockIdx.z * 64;
    const int to = (threadIdx.x >> 4) * 4, ti = (threadIdx.x & 15) * 4;
    float acc[4][4] = {};
    for (int64_t base = (int64_t)blockIdx.x * GW_PTS; base < npts; base += (int64_t)gridDim.x * GW_PTS) {
        for (int e = threadIdx.x; e < GW_PTS * 64; e += 256) {
            const int j = e >> 6, k = e & 63;
            const int64_t pt = base + j;
            float gv = 0.0f, xv = 0.0f;
            if (pt < npts) {
                if (o0 + k < out) gv = g.p[pt * g.stride + g.off + o0 + k];
                const int col = i0 + k;
                if (col < in) xv = (col < a.n) ? a.p[pt * a.stride + a.off + col] : b.p[pt * b.stride + b.off + (col - a.n)];
            }
            gs[j][k] = gv; xs[j][k] = xv;
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < GW_PTS; j++) {
            const float4 g4 = *reinterpret_cast<const float4 *>(&gs[j][to]), x4 = *reinterpret_cast<const float4 *>(&xs[j][ti]);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[q][r] = __builtin_fmaf(gv[q], xv[r], acc[q][r]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = o0 + to + q, i = i0 + ti + r;
            if (o < out && i < in && acc[q][r] != 0.0f) unsafeAtomicAdd(dw + (size_t)o * in + i, acc[q][r]);
        }
}

int run_grad_w(int64_t npts, Seg g, Seg a, Seg b, int out, int in, float *dw, hipStream_t st)
{
    const int64_t nb = ceil_div(npts, GW_PTS);
    dim3 grid((unsigned)(nb < 256 ? nb : 256), (unsigned)ceil_div(out, 64), (unsigned)ceil_div(in, 64));
    hipLaunchKernelGGL(k_grad_w, grid, dim3(256), 0, st, npts, g, a, b, out, in, dw);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// g_in[pt][k] = sum_o g[pt][o] * W[o][k]: k_linear with the blob matrix as its [in' = out][out' = in] operand
int run_backprop(int64_t npts, Seg g, const nrf_mlp *m, const LinearLayer &L, float *y, int y_stride, hipStream_t st)
{
    const Seg none{nullptr, 0, 0, 0};
    const int threads = L.in >= 256 ? 256 : (int)ceil_div(L.in, 64) * 64;
    const size_t lds = (size_t)L.out * LIN_TP * sizeof(float);
    hipLaunchKernelGGL(k_linear, dim3((unsigned)ceil_div(npts, LIN_TP)), dim3(threads), lds, st, npts, g, none, m->d_params + L.w_off, (const float *)nullptr, L.in, 0, y,
                       y_stride, 0);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

#ifndef NRF_BWD_CHUNK_LOG2
#define NRF_BWD_CHUNK_LOG2 20
#endif
static const int64_t BWD_CHUNK = (int64_t)1 << NRF_BWD_CHUNK_LOG2;          // points per pass of the fp32 backward (bounds the scratch; docs/history/profiles/round5/r5X_*)

size_t mlp_backward_workspace_bytes(const nrf_mlp *m, int64_t p)
{
    const int64_t c = p < BWD_CHUNK ? p : BWD_CHUNK;
    return align_up((size_t)c * m->max_width * sizeof(float), 256) * (m->layers.size() + 3);
}

int mlp_small_backward(const nrf_mlp *m, const float *x, int xs, const float *g_out, int gos, int64_t p, float *g_params, float *g_x, int gxs,
                       void *ws, size_t ws_bytes, hipStream_t st)
{
    if (m->family != MLP_SMALL) { set_error("nrf_mlp_backward: built for the NeRFSmall family"); return NRF_ERR_UNSUPPORTED; }
    if (ws_bytes < mlp_backward_workspace_bytes(m, p)) { set_error("nrf_mlp_backward: workspace %zu < %zu bytes", ws_bytes, mlp_backward_workspace_bytes(m, p)); return NRF_ERR_WORKSPACE; }
    const auto &d = m->small;
    // (a predicted-normals head, if the handle has one, sits behind these layers and receives no gradient: the training loss reads RGBMap only, NeRFExecutor.h:882-887,
    // and nothing else reads the normals -- its parameters' gradient stays zero, as in the reference's autograd)
    const int W = m->max_width, nl = d.num_layers + d.num_layers_color;
    const size_t buf = align_up((size_t)(p < BWD_CHUNK ? p : BWD_CHUNK) * W * sizeof(float), 256) / sizeof(float);
    float *base = reinterpret_cast<float *>(ws);
    std::vector<float *> H(nl);
    for (int l = 0; l < nl; l++) H[l] = base + (size_t)l * buf;
    float *G[3] = {base + (size_t)nl * buf, base + (size_t)(nl + 1) * buf, base + (size_t)(nl + 2) * buf};
    const Seg none{nullptr, 0, 0, 0};
    for (int64_t p0 = 0; p0 < p; p0 += BWD_CHUNK) {
        const int64_t c = (p - p0) < BWD_CHUNK ? (p - p0) : BWD_CHUNK;
        const float *xc = x + p0 * xs;
        const float *gc = g_out + p0 * gos;
        // ---- forward, keeping every layer output (post-ReLU for hidden layers) ----
        Seg cur{xc, xs, 0, d.input_ch};
        int li = 0;
        for (int l = 0; l < d.num_layers; l++, li++) {
            NRF_TRY(run_linear_fast(c, cur, none, m, m->layers[li], l != d.num_layers - 1, H[li], W, 0, st));
            cur = Seg{H[li], W, 0, m->layers[li].out};
        }
        const int sig_l = li - 1;                             // H[sig_l]: column 0 = sigma, 1.. = geo
        const Seg cv{xc, xs, d.input_ch, d.input_ch_views}, cg{H[sig_l], W, 1, d.geo_feat_dim};
        for (int l = 0; l < d.num_layers_color; l++, li++) {
            NRF_TRY(run_linear_fast(c, l == 0 ? cv : cur, l == 0 ? cg : none, m, m->layers[li], l != d.num_layers_color - 1, H[li], W, 0, st));
            cur = Seg{H[li], W, 0, m->layers[li].out};
        }
        // ---- colour net backward ----
        Seg g{gc, gos, 0, 3};
        int gi = 0;
        for (int l = nl - 1; l > sig_l; l--) {
            const LinearLayer &L = m->layers[l];
            if (l != nl - 1) {
                hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)ceil_div(c * L.out, 256)), dim3(256), 0, st, c * L.out, L.out, const_cast<float *>(g.p), g.stride, H[l], W);
                NRF_LAUNCH_CHECK();
            }
            const bool first = (l == sig_l + 1);
            NRF_TRY(run_grad_w_fast(c, g, first ? cv : Seg{H[l - 1], W, 0, L.in}, first ? cg : none, L.out, L.in, g_params + L.w_off, st, train_gemm_for(m)));
            float *dst = G[gi]; gi = (gi + 1) % 3;
            NRF_TRY(run_backprop_fast(c, g, m, L, dst, W, st));
            g = Seg{dst, W, 0, L.in};
        }
        // sigma-net output gradient: column 0 = d/d sigma (g_out[:,3]), 1..geo = d/d geo (columns views.. of the colour input gradient)
        float *gh = G[gi]; gi = (gi + 1) % 3;
        hipLaunchKernelGGL(k_copy_col, dim3((unsigned)ceil_div(c, 256)), dim3(256), 0, st, c, gc, gos, 3, gh, W, 0);
        NRF_LAUNCH_CHECK();
        for (int k = 0; k < d.geo_feat_dim; k++) {
            hipLaunchKernelGGL(k_copy_col, dim3((unsigned)ceil_div(c, 256)), dim3(256), 0, st, c, g.p, g.stride, d.input_ch_views + k, gh, W, 1 + k);
            NRF_LAUNCH_CHECK();
        }
        g = Seg{gh, W, 0, 1 + d.geo_feat_dim};
        for (int l = sig_l; l >= 0; l--) {
            const LinearLayer &L = m->layers[l];
            if (l != sig_l) {
                hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)ceil_div(c * L.out, 256)), dim3(256), 0, st, c * L.out, L.out, const_cast<float *>(g.p), g.stride, H[l], W);
                NRF_LAUNCH_CHECK();
            }
            NRF_TRY(run_grad_w_fast(c, g, l == 0 ? Seg{xc, xs, 0, d.input_ch} : Seg{H[l - 1], W, 0, L.in}, none, L.out, L.in, g_params + L.w_off, st, train_gemm_for(m)));
            if (l == 0 && !g_x) break;
            float *dst = (l == 0) ? g_x + p0 * gxs : G[gi];
            gi = (gi + 1) % 3;
            NRF_TRY(run_backprop_fast(c, g, m, L, dst, l == 0 ? gxs : W, st));
            g = Seg{dst, l == 0 ? gxs : W, 0, L.in};
        }
    }
    return NRF_OK;
}

// ---------------------------------------------------------------------------------------------------
// N1 for the classic model: backward of NeRFImpl::forward (NeRF.cpp:92-126), fp32, from the same building blocks.  The layers have biases (d loss / d bias = the column
// sums of the layer's output gradient); after layer `skip` the input is cat[input_pts, h] (:103-104); with view directions the head is alpha_linear(h), feature_linear(h)
// (no ReLU), relu(views_linears_0(cat[feature, views])), rgb_linear (:108-120), without them output_linear(cat[h, input_pts]) (:121-124).
// Pinned by LibTorch autograd through the compiled NeRF.cpp (goldens mlp_nerf_bwd*, train_classic).
// ---------------------------------------------------------------------------------------------------
// db[o] += sum_pt g[pt][o]
__global__ void __launch_bounds__(256) k_grad_b(int64_t npts, Seg g, int out, float *__restrict__ db)
{
    const int o = blockIdx.y * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;                       // four point phases per block
    // four rows in flight per thread (a classic training step spends 288 of these on 131 072 x 256 gradients: at one dependent load per thread they ran at 1.3 TB/s)
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (o < out) {
        const int64_t step = (int64_t)gridDim.x * 4;
        const float *p = g.p + g.off + o;
        int64_t pt = (int64_t)blockIdx.x * 4 + sub;
        for (; pt + 3 * step < npts; pt += 4 * step) {
            a0 += p[pt * g.stride]; a1 += p[(pt + step) * g.stride]; a2 += p[(pt + 2 * step) * g.stride]; a3 += p[(pt + 3 * step) * g.stride];
        }
        for (; pt < npts; pt += step) a0 += p[pt * g.stride];
    }
    const float acc = (a0 + a1) + (a2 + a3);
    if (o < out && acc != 0.0f) unsafeAtomicAdd(db + o, acc);
}

int run_grad_b(int64_t npts, Seg g, int out, float *db, hipStream_t st)
{
    const int64_t nb = ceil_div(npts, 16);
    hipLaunchKernelGGL(k_grad_b, dim3((unsigned)(nb < 512 ? (nb < 1 ? 1 : nb) : 512), (unsigned)ceil_div(out, 64)), dim3(256), 0, st, npts, g, out, db);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// y[pt][k] = a[pt][a_off + k] (+ b[pt][b_off + k]), k < n
__global__ void k_sum_cols(int64_t npts, int n, const float *__restrict__ a, int a_stride, int a_off, const float *__restrict__ b, int b_stride, int b_off, float *__restrict__ y,
                           int y_stride, const float *__restrict__ mask, int mask_stride)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npts * n) return;
    const int64_t pt = e / n; const int k = (int)(e - pt * n);
    float v = a[pt * a_stride + a_off + k];
    if (b) v = v + b[pt * b_stride + b_off + k];
    if (mask) v = mask[pt * mask_stride + k] > 0.0f ? v : 0.0f;          // the ReLU mask of the stage that consumes y (what run_relu_mask would do in a pass of its own)
    y[pt * y_stride + k] = v;
}

// y[pt][k] = (a[pt][k] + s[pt] w[k]) . [mask[pt][k] > 0]: a sum whose second term is a rank-1 product (the back-propagation through a one-row layer)
__global__ void k_sum_rank1(int64_t npts, int n, const float *__restrict__ a, int a_stride, const float *__restrict__ s, int s_stride, const float *__restrict__ w, float *__restrict__ y,
                            int y_stride, const float *__restrict__ mask, int mask_stride)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npts * n) return;
    const int64_t pt = e / n; const int k = (int)(e - pt * n);
    float v = a[pt * a_stride + k] + s[pt * s_stride] * w[k];
    if (mask) v = mask[pt * mask_stride + k] > 0.0f ? v : 0.0f;
    y[pt * y_stride + k] = v;
}

static int run_sum_cols(int64_t npts, int n, const float *a, int a_stride, int a_off, const float *b, int b_stride, int b_off, float *y, int y_stride, hipStream_t st,
                        const float *mask = nullptr, int mask_stride = 0)
{
    hipLaunchKernelGGL(k_sum_cols, dim3((unsigned)ceil_div(npts * n, 256)), dim3(256), 0, st, npts, n, a, a_stride, a_off, b, b_stride, b_off, y, y_stride, mask, mask_stride);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// dW and db of one layer: the split-precision modes do both in the weight-gradient product's pass over g (gemm_tn_bf16x3_2 sums g's columns on its way through the
// registers); otherwise the two kernels
static int run_grad_wb_fast(int64_t npts, Seg g, Seg a, Seg b, int out, int in, float *dw, float *db, hipStream_t st, int arith)
{
    if (arith != 0 && npts >= 4096 && out >= 32 && a.n > 0) {
        NRF_TRY(gemm_tn_bf16x3_2(npts, g, a, 0, Seg{nullptr, 0, 0, 0}, 0, 0, 0, out, in, dw, st, db));
        if (b.n > 0) NRF_TRY(gemm_tn_bf16x3(npts, g, b, out, in, a.n, dw, st));
        return NRF_OK;
    }
    NRF_TRY(run_grad_w_fast(npts, g, a, b, out, in, dw, st, arith));
    return run_grad_b(npts, g, out, db, st);
}

static size_t mlp_nerf_backward_workspace_bytes(const nrf_mlp *m, int64_t p)
{
    const int64_t c = p < BWD_CHUNK ? p : BWD_CHUNK;
    const int pad = m->nerf.input_ch > m->nerf.input_ch_views ? m->nerf.input_ch : m->nerf.input_ch_views;
    return align_up((size_t)c * (size_t)((m->max_width + pad + 7) & ~7) * sizeof(float), 256) * (m->nerf.depth + 7);
}

int mlp_nerf_backward(const nrf_mlp *m, const float *x, int xs, const float *g_out, int gos, int64_t p, float *g_params, float *g_x, int gxs, void *ws, size_t ws_bytes,
                      hipStream_t st)
{
    if (ws_bytes < mlp_nerf_backward_workspace_bytes(m, p)) { set_error("nrf_mlp_backward: workspace %zu < %zu bytes", ws_bytes, mlp_nerf_backward_workspace_bytes(m, p)); return NRF_ERR_WORKSPACE; }
    const auto &d = m->nerf;
    const int D = d.depth, Wd = d.width, in = d.input_ch, iv = d.input_ch_views;
    // row stride of every scratch buffer: the widest row (cat[input_pts, h] / cat[feature, views]), rounded up to 8 floats so that every row starts 32-byte aligned (the
    // matrix-core products of gemm_bf16x3.hip load rows with 16-byte vectors)
    const int W = (m->max_width + (in > iv ? in : iv) + 7) & ~7;
    const size_t buf = align_up((size_t)(p < BWD_CHUNK ? p : BWD_CHUNK) * W * sizeof(float), 256) / sizeof(float);
    float *base = reinterpret_cast<float *>(ws);
    std::vector<float *> H(D);
    for (int l = 0; l < D; l++) H[l] = base + (size_t)l * buf;
    float *FEAT = base + (size_t)D * buf, *HV = base + (size_t)(D + 1) * buf;
    float *G[5] = {base + (size_t)(D + 2) * buf, base + (size_t)(D + 3) * buf, base + (size_t)(D + 4) * buf, base + (size_t)(D + 5) * buf, base + (size_t)(D + 6) * buf};
    const Seg none{nullptr, 0, 0, 0};
    auto bias_of = [&](const LinearLayer &L) { return g_params + L.w_off + (size_t)L.in * L.out; };
    for (int64_t p0 = 0; p0 < p; p0 += BWD_CHUNK) {
        const int64_t c = (p - p0) < BWD_CHUNK ? (p - p0) : BWD_CHUNK;
        const float *xc = x + p0 * xs, *gc = g_out + p0 * gos;
        const Seg xin{xc, xs, 0, in};
        // ---- forward, every layer output kept (NeRF.cpp:92-106) ----
        Seg cur = xin;
        bool cat_in = false;
        // the hidden layers' ReLU masks as bits, 32 bytes per row in the row's own padding (columns [Wd, Wd + 8) of the W-wide buffers): written by the forward products, read
        // by the back-propagation products in place of the 1 KB activation row
        const bool bits = W - Wd >= 8 && run_backprop_uses_mask_bits(m, c, Wd);
        auto bits_of = [&](float *h) { return bits ? reinterpret_cast<uint64_t *>(h + Wd) : nullptr; };
        const int bits_ld = W / 2;
        for (int l = 0; l < D; l++) {
            NRF_TRY(run_linear_fast(c, cat_in ? xin : cur, cat_in ? cur : none, m, m->layers[l], 1, H[l], W, 0, st, bits_of(H[l]), bits_ld));
            cur = Seg{H[l], W, 0, Wd};
            cat_in = (l == d.skip);
        }
        const Seg hlast = cur;
        float *gh = G[0];                                    // d loss / d (the last pts layer's post-ReLU output)
        bool gh_masked = false;                              // ... already multiplied by that output's ReLU mask
        float *gx_acc = nullptr;                             // d loss / d input_pts collected on the way (G[4] when wanted)
        auto add_gx = [&](const float *src, int src_stride, int src_off) -> int {
            if (!g_x) return NRF_OK;
            if (!gx_acc) { gx_acc = G[4]; return run_sum_cols(c, in, src, src_stride, src_off, nullptr, 0, 0, gx_acc, W, st); }
            return run_sum_cols(c, in, src, src_stride, src_off, gx_acc, W, 0, gx_acc, W, st);
        };
        if (d.use_viewdirs) {
            const LinearLayer &views = m->layers[D], &feat = m->layers[D + 1], &alpha = m->layers[D + 2], &rgb = m->layers[D + 3];
            NRF_TRY(run_linear_fast(c, hlast, none, m, feat, 0, FEAT, W, 0, st));                                              // feature (no ReLU)                 :110
            const Seg sfeat{FEAT, W, 0, Wd}, sviews{xc, xs, in, iv};
            NRF_TRY(run_linear_fast(c, sfeat, sviews, m, views, 1, HV, W, 0, st));                                             // relu(views_linears_0(cat))         :111-117
            const Seg g_rgb{gc, gos, 0, 3}, g_alpha{gc, gos, 3, 1};
            // rgb_linear                                                                                              :118
            NRF_TRY(run_grad_w_fast(c, g_rgb, Seg{HV, W, 0, Wd / 2}, none, 3, Wd / 2, g_params + rgb.w_off, st, train_gemm_for(m)));
            NRF_TRY(run_grad_b(c, g_rgb, 3, bias_of(rgb), st));
            if (run_backprop_fuses_mask(m, c)) NRF_TRY(run_backprop_fast(c, g_rgb, m, rgb, G[1], W, st, HV, W));          // (+ the views layer's ReLU mask, in the product's write-out)
            else {
                NRF_TRY(run_backprop_fast(c, g_rgb, m, rgb, G[1], W, st));
                NRF_TRY(run_relu_mask(c, Wd / 2, G[1], W, HV, W, st));
            }
            const Seg g_hv{G[1], W, 0, Wd / 2};
            // views_linears_0
            NRF_TRY(run_grad_wb_fast(c, g_hv, sfeat, sviews, Wd / 2, Wd + iv, g_params + views.w_off, bias_of(views), st, train_gemm_for(m)));
            NRF_TRY(run_backprop_fast(c, g_hv, m, views, G[2], W, st));                                                     // d / d cat[feature, views]
            const Seg g_feat{G[2], W, 0, Wd};
            // feature_linear and alpha_linear, both on h                                                              :108-110
            NRF_TRY(run_grad_wb_fast(c, g_feat, hlast, none, Wd, Wd, g_params + feat.w_off, bias_of(feat), st, train_gemm_for(m)));
            NRF_TRY(run_grad_w_fast(c, g_alpha, hlast, none, 1, Wd, g_params + alpha.w_off, st, train_gemm_for(m)));
            NRF_TRY(run_grad_b(c, g_alpha, 1, bias_of(alpha), st));
            // d / d h through alpha_linear is the rank-1 product g_alpha (x) w_alpha: formed inside the sum (no [c, 256] array of its own), with h_{D-1}'s ReLU mask --
            // the first pts_linears stage below finds it applied.  Split-precision modes: both in the write-out of the back-propagation product through feature_linear
            if (const int arith = run_backprop_fuses_mask(m, c) ? train_gemm_for(m) : 0)
                NRF_TRY(gemm_nt_split(arith, c, feat.in, g_feat, none, feat.d_wt, feat.out, gh, W, nullptr, 0, H[D - 1], W, st, nullptr, 0, g_alpha.p + g_alpha.off, g_alpha.stride,
                                      m->d_params + alpha.w_off, nullptr, 0, bits_of(H[D - 1]), bits_ld));
            else {
            NRF_TRY(run_backprop_fast(c, g_feat, m, feat, G[1], W, st));
            hipLaunchKernelGGL(k_sum_rank1, dim3((unsigned)ceil_div(c * Wd, 256)), dim3(256), 0, st, c, Wd, (const float *)G[1], W, g_alpha.p + g_alpha.off, g_alpha.stride,
                               (const float *)(m->d_params + alpha.w_off), gh, W, (const float *)H[D - 1], W);
            NRF_LAUNCH_CHECK();
            }
            gh_masked = true;
        } else {
            const LinearLayer &outl = m->layers[D];                                                                    // output_linear(cat[h, input_pts])  :121-124
            const Seg g_o{gc, gos, 0, outl.out};
            NRF_TRY(run_grad_wb_fast(c, g_o, hlast, xin, outl.out, Wd + in, g_params + outl.w_off, bias_of(outl), st, train_gemm_for(m)));
            NRF_TRY(run_backprop_fast(c, g_o, m, outl, G[1], W, st));
            NRF_TRY(run_sum_cols(c, Wd, G[1], W, 0, nullptr, 0, 0, gh, W, st));
            NRF_TRY(add_gx(G[1], W, Wd));
        }
        // ---- pts_linears, last first ----
        float *gcur = gh;
        const bool fuse = run_backprop_fuses_mask(m, c);          // bf16x3 products: the next stage's ReLU mask in the back-propagation product's epilogue
        bool premasked = gh_masked;
        for (int l = D - 1; l >= 0; l--) {
            const LinearLayer &L = m->layers[l];
            if (!premasked) NRF_TRY(run_relu_mask(c, Wd, gcur, W, H[l], W, st));
            premasked = false;
            const Seg g{gcur, W, 0, Wd};
            const bool cat = (l > 0) && (l - 1 == d.skip);                                                             // this layer's input is cat[input_pts, h_{l-1}]
            const Seg a = (l == 0) ? xin : (cat ? xin : Seg{H[l - 1], W, 0, Wd});
            const Seg b = cat ? Seg{H[l - 1], W, 0, Wd} : none;
            NRF_TRY(run_grad_wb_fast(c, g, a, b, Wd, L.in, g_params + L.w_off, bias_of(L), st, train_gemm_for(m)));
            if (l == 0 && !g_x) break;
            float *dst = (gcur == G[1]) ? G[2] : G[1];
            if (cat && fuse && !g_x) {
                // the skip layer's input is cat[input_pts, h]: nobody asks for d / d input_pts, so only the h columns are formed -- rows [in, in + Wd) of W^T as a product of
                // their own, which lands at column 0 with h_{l-1}'s ReLU mask applied by its write-out (the whole product + a masked copy of 256 of its 319 columns before)
                NRF_TRY(gemm_nt_split(train_gemm_for(m), c, Wd, g, none, L.d_wt + (size_t)in * L.out, L.out, dst, W, nullptr, 0, H[l - 1], W, st, nullptr, 0, nullptr, 0, nullptr,
                                      nullptr, 0, bits_of(H[l - 1]), bits_ld));
                premasked = true;
                gcur = dst;
                continue;
            }
            const bool mask_next = fuse && l > 0 && !cat;          // dst = d / d H[l - 1] (with the skip concat the h part sits at a column offset: masked by its own pass)
            NRF_TRY(run_backprop_fast(c, g, m, L, dst, W, st, mask_next ? H[l - 1] : nullptr, W, nullptr, 0, mask_next ? bits_of(H[l - 1]) : nullptr, bits_ld));
            premasked = mask_next;
            if (l == 0) NRF_TRY(add_gx(dst, W, 0));
            else if (cat) {
                NRF_TRY(add_gx(dst, W, 0));
                float *nxt = (dst == G[1]) ? G[2] : G[1];
                // the h part of the gradient, moved to column 0 of another buffer (never in place: rows overlap)
                nxt = (nxt == gcur) ? G[3] : nxt;
                NRF_TRY(run_sum_cols(c, Wd, dst, W, in, nullptr, 0, 0, nxt, W, st, H[l - 1], W));          // (+ h_{l-1}'s ReLU mask)
                premasked = true;
                gcur = nxt;
                continue;
            }
            gcur = dst;
        }
        if (g_x) {
            if (gx_acc) NRF_TRY(run_sum_cols(c, in, gx_acc, W, 0, nullptr, 0, 0, g_x + p0 * gxs, gxs, st));
        }
    }
    return NRF_OK;
}

// ---------------------------------------------------------------------------------------------------
// handle construction
// ---------------------------------------------------------------------------------------------------
static int add_layer(nrf_mlp *m, const std::vector<float> &hp, size_t &off, int in, int out, bool bias)
{
    LinearLayer L;
    L.in = in; L.out = out; L.w_off = off;
    std::vector<float> wt((size_t)in * out);
    for (int o = 0; o < out; o++)
        for (int k = 0; k < in; k++) wt[(size_t)k * out + o] = hp[off + (size_t)o * in + k];
    off += (size_t)in * out;
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&L.d_wt), wt.size() * sizeof(float)));
    NRF_HIP(hipMemcpy(L.d_wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
    if (bias) {
        NRF_HIP(hipMalloc(reinterpret_cast<void **>(&L.d_bias), (size_t)out * sizeof(float)));
        NRF_HIP(hipMemcpy(L.d_bias, hp.data() + off, (size_t)out * sizeof(float), hipMemcpyHostToDevice));
        off += out;
    }
    m->layers.push_back(L);
    if (in > m->max_width) m->max_width = in;
    if (out > m->max_width) m->max_width = out;
    return NRF_OK;
}

int host_pack_threads()
{
    static const int n = [] {
        if (const char *e = getenv("NRF_PACK_THREADS")) { const int v = atoi(e); if (v >= 1) return v > 64 ? 64 : v; }
        const unsigned hw = std::thread::hardware_concurrency();
        return hw >= 8 ? 8 : (hw >= 2 ? (int)hw : 1);          // (16 / 32 threads: no faster, profiles/round6/r6w_pack_threads.log)
    }();
    return n;
}

static int fetch_params(const float *params, int on_device, int64_t n, hipStream_t st, std::vector<float> &host, nrf_mlp *m)
{
    host.resize((size_t)n);
    if (on_device) {
        NRF_HIP(hipMemcpyAsync(host.data(), params, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        NRF_HIP(hipStreamSynchronize(st));
    } else memcpy(host.data(), params, (size_t)n * 4);
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_params), (size_t)n * 4));
    NRF_HIP(hipMemcpy(m->d_params, host.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    m->n_params = n;
    return NRF_OK;
}

// ---------------------------------------------------------------------------------------------------
// weight maps: every derived image of a NeRFSmall handle as a gather from the parameter blob (mlp.h)
// ---------------------------------------------------------------------------------------------------
// The host packers (mlp_small_mfma.hip, sigma_small_f32.hip, mlp_small_bwd_mfma.hip) stay the single statement of the fragment layouts.  They are run on four PROBE
// blobs whose entries spell their own index: probes 0 / 1 carry the index in fp16-exact values 1 + a / 1024 (hi halves and fp32 copies read it back, lo halves are 0
// there); probes 2 / 3 are 0.5 + b 2^-24, whose hi half is 0.5 and whose lo half is b 2^-24 exactly.  The decoded (source, kind) table is then checked against the
// host-packed image of the REAL parameters byte for byte; any mismatch leaves the handle on the host repack.
static const int WM_RADIX_HI = 1024, WM_RADIX_LO = 1023;

static std::vector<float> weight_map_probe(int64_t n, int which)
{
    std::vector<float> v((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (which == 0) v[i] = 1.0f + (float)(i % WM_RADIX_HI) / 1024.0f;
        else if (which == 1) v[i] = 1.0f + (float)(i / WM_RADIX_HI) / 1024.0f;
        else if (which == 2) v[i] = 0.5f + (float)(i % WM_RADIX_LO + 1) * 0x1p-24f;
        else v[i] = 0.5f + (float)(i / WM_RADIX_LO + 1) * 0x1p-24f;
    }
    return v;
}

static inline float wm_value(const uint8_t *img, int64_t e, int elem)
{
    if (elem == 4) { float f; memcpy(&f, img + 4 * e, 4); return f; }
    _Float16 h; memcpy(&h, img + 2 * e, 2); return (float)h;
}

struct HostMap { std::vector<int32_t> src; std::vector<uint8_t> kind; int elem = 2; };

static bool decode_weight_map(const std::vector<uint8_t> probe_img[4], int elem, int64_t n_params, HostMap &hm)
{
    const int64_t n = (int64_t)probe_img[0].size() / elem;
    for (int q = 1; q < 4; q++) if ((int64_t)probe_img[q].size() != n * elem) return false;
    hm.elem = elem; hm.src.assign((size_t)n, -1); hm.kind.assign((size_t)n, WM_ZERO);
    for (int64_t e = 0; e < n; e++) {
        const float a = wm_value(probe_img[0].data(), e, elem), b = wm_value(probe_img[1].data(), e, elem);
        const float c = wm_value(probe_img[2].data(), e, elem), d = wm_value(probe_img[3].data(), e, elem);
        int64_t i = -1; uint8_t k = WM_ZERO;
        if (a != 0.0f) {
            i = (int64_t)lrintf((b - 1.0f) * 1024.0f) * WM_RADIX_HI + lrintf((a - 1.0f) * 1024.0f);
            k = elem == 4 ? WM_F32 : WM_F16_HI;
        } else if (c != 0.0f) {
            if (elem == 4) return false;
            i = ((int64_t)lrintf(d * 0x1p24f) - 1) * WM_RADIX_LO + (lrintf(c * 0x1p24f) - 1);
            k = WM_F16_LO;
        }
        if (k != WM_ZERO && (i < 0 || i >= n_params)) return false;
        hm.src[(size_t)e] = (int32_t)i; hm.kind[(size_t)e] = k;
    }
    return true;
}

static bool weight_map_reproduces(const HostMap &hm, const std::vector<float> &hp, const std::vector<uint8_t> &img)
{
    const int64_t n = (int64_t)hm.src.size();
    if ((int64_t)img.size() != n * hm.elem) return false;
    for (int64_t e = 0; e < n; e++) {
        const float v = hm.kind[(size_t)e] == WM_ZERO ? 0.0f : hp[(size_t)hm.src[(size_t)e]];
        if (hm.elem == 4) {
            if (memcmp(&v, img.data() + 4 * e, 4) != 0) return false;
        } else {
            _Float16 h = (_Float16)v;
            if (hm.kind[(size_t)e] == WM_F16_LO) h = (_Float16)(v - (float)h);
            if (hm.kind[(size_t)e] == WM_ZERO) h = (_Float16)0.0f;
            if (memcmp(&h, img.data() + 2 * e, 2) != 0) return false;
        }
    }
    return true;
}

static int upload_weight_map(nrf_mlp *m, const HostMap &hm, void *d_out)
{
    WeightMap wm;
    wm.n = (int64_t)hm.src.size(); wm.elem = hm.elem; wm.d_out = d_out;
    if (wm.n == 0) return NRF_OK;
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&wm.d_src), (size_t)wm.n * 4));
    NRF_HIP(hipMemcpy(wm.d_src, hm.src.data(), (size_t)wm.n * 4, hipMemcpyHostToDevice));
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&wm.d_kind), (size_t)wm.n));
    NRF_HIP(hipMemcpy(wm.d_kind, hm.kind.data(), (size_t)wm.n, hipMemcpyHostToDevice));
    m->maps.push_back(wm);
    return NRF_OK;
}

static void drop_weight_maps(nrf_mlp *m)
{
    for (auto &w : m->maps) { if (w.d_src) (void)hipFree(w.d_src); if (w.d_kind) (void)hipFree(w.d_kind); }
    m->maps.clear();
}

// NeRFSmall, after the images exist: NRF_OK with m->maps filled, or with m->maps empty when a layout does not decode (the host repack then stays in charge)
static int build_weight_maps(nrf_mlp *m, const std::vector<float> &hp)
{
    if (m->family != MLP_SMALL || m->n_params >= (int64_t)WM_RADIX_LO * WM_RADIX_LO) return NRF_OK;
    for (auto &L : m->layers) if (L.d_bias) return NRF_OK;
    if (const char *e = getenv("NRF_MLP_HOST_REPACK")) if (atoi(e) != 0) return NRF_OK;
    const auto &d = m->small;
    enum { I_F16, I_SPLIT, I_BWD, I_SIG_HEAD, I_SIG_TAIL, N_IMG };
    auto images = [&](const std::vector<float> &p, std::vector<uint8_t> *out) {
        const bool a = mlp_small_images_host(d, p, out[I_F16], out[I_SPLIT]);
        const bool b = mlp_small_bwd_image_host(m, p, out[I_BWD]);
        const bool c = mlp_small_sigma_image_host(d, p, out[I_SIG_HEAD], out[I_SIG_TAIL]);
        if (!a) { out[I_F16].clear(); out[I_SPLIT].clear(); }
        if (!b) out[I_BWD].clear();
        if (!c) { out[I_SIG_HEAD].clear(); out[I_SIG_TAIL].clear(); }
    };
    std::vector<uint8_t> real[N_IMG], probe[4][N_IMG];
    images(hp, real);
    for (int q = 0; q < 4; q++) images(weight_map_probe(m->n_params, q), probe[q]);
    const int elem[N_IMG] = { 2, 2, 2, 4, 2 };
    void *dst[N_IMG] = { m->d_packed_f16, m->d_packed_split, m->d_packed_bwd, m->d_packed_sigma_f32,
                         m->d_packed_sigma_f32 ? static_cast<char *>(m->d_packed_sigma_f32) + real[I_SIG_HEAD].size() : nullptr };
    const size_t have[N_IMG] = { m->packed_f16_bytes, m->packed_split_bytes, m->packed_bwd_bytes, m->packed_sigma_f32_bytes, m->packed_sigma_f32_bytes };
    for (int im = 0; im < N_IMG; im++) {
        if (real[im].empty()) { if (dst[im] && im != I_SIG_TAIL && have[im]) { drop_weight_maps(m); return NRF_OK; } continue; }
        const size_t want = (im == I_SIG_HEAD || im == I_SIG_TAIL) ? real[I_SIG_HEAD].size() + real[I_SIG_TAIL].size() : real[im].size();
        const std::vector<uint8_t> four[4] = { probe[0][im], probe[1][im], probe[2][im], probe[3][im] };
        HostMap hm;
        if (!dst[im] || have[im] != want || !decode_weight_map(four, elem[im], m->n_params, hm) || !weight_map_reproduces(hm, hp, real[im])) { drop_weight_maps(m); return NRF_OK; }
        NRF_TRY(upload_weight_map(m, hm, dst[im]));
        m->maps.back().scaled = im == I_SPLIT || im == I_SIG_TAIL;      // the split-precision operands: gathered from the range-scaled blob (mlp_small_rescale)
    }
    for (auto &L : m->layers) {                       // W^T [in][out] of the generic fp32 forward / backward
        HostMap hm; hm.elem = 4;
        hm.src.resize((size_t)L.in * L.out); hm.kind.assign(hm.src.size(), WM_F32);
        for (int o = 0; o < L.out; o++)
            for (int k = 0; k < L.in; k++) hm.src[(size_t)k * L.out + o] = (int32_t)(L.w_off + (size_t)o * L.in + k);
        NRF_TRY(upload_weight_map(m, hm, L.d_wt));
    }
    return NRF_OK;
}

// ---- classic NeRF (8 x 256): the same, with the merged views layer as a DERIVED block behind the blob ----
// The packers' images are gathers of [blob | merged (views_linears_0 o feature_linear) | merged_b]; the merged block is a product, re-derived on the device by
// k_nerf_merged_f64 with the host's double sums in the host's order (nerf_merged_views_host: f ascending, one accumulator per entry): the same bits.  The handle's
// d_params buffer is re-allocated with room for the block behind the blob, so the maps' source indices run over one array.
constexpr int NERF_MERGED_ROWS = 128, NERF_MERGED_W = 256, NERF_DERIVED = NERF_MERGED_ROWS * NERF_MERGED_W + NERF_MERGED_ROWS;

__global__ void __launch_bounds__(256) k_nerf_merged_f64(const float *__restrict__ wv, int wv_stride, const float *__restrict__ wf, const float *__restrict__ bf,
                                                         const float *__restrict__ bv, float *__restrict__ merged, float *__restrict__ merged_b)
{
    const int r = blockIdx.x, k = threadIdx.x;
    double acc = 0.0;
    for (int f = 0; f < NERF_MERGED_W; f++) acc += (double)wv[(size_t)r * wv_stride + f] * (double)wf[(size_t)f * NERF_MERGED_W + k];
    merged[(size_t)r * NERF_MERGED_W + k] = (float)acc;
    if (k == 0) {
        double b = (double)bv[r];
        for (int f = 0; f < NERF_MERGED_W; f++) b += (double)wv[(size_t)r * wv_stride + f] * (double)bf[f];
        merged_b[r] = (float)b;
    }
}

static int nerf_derive_on_device(const nrf_mlp *m, hipStream_t st)
{
    const auto &Lv = m->layers[8], &Lf = m->layers[9];          // views_linears_0, feature_linear (nrf_mlp_nerf_create's order)
    const float *wv = m->d_params + Lv.w_off, *bv = wv + (size_t)Lv.in * Lv.out, *wf = m->d_params + Lf.w_off, *bf = wf + (size_t)Lf.in * Lf.out;
    float *merged = m->d_params + m->n_params;
    k_nerf_merged_f64<<<dim3(NERF_MERGED_ROWS), dim3(NERF_MERGED_W), 0, st>>>(wv, Lv.in, wf, bf, bv, merged, merged + (size_t)NERF_MERGED_ROWS * NERF_MERGED_W);
    NRF_HIP(hipGetLastError());
    return NRF_OK;
}

static std::vector<uint8_t> wm_bytes(const void *p, size_t n) { const uint8_t *b = static_cast<const uint8_t *>(p); return std::vector<uint8_t>(b, b + n); }

// classic NeRF, after the images exist: NRF_OK with m->maps filled (and d_params re-homed with the derived block), or with m->maps empty (host repack stays in charge)
static int build_weight_maps_nerf(nrf_mlp *m, const std::vector<float> &hp)
{
    const auto &d = m->nerf;
    if (m->family != MLP_NERF || !m->d_packed_f16 || !m->d_packed_split || !m->d_packed_sigma_f32 || m->layers.size() != 12) return NRF_OK;
    if (const char *e = getenv("NRF_MLP_HOST_REPACK")) if (atoi(e) != 0) return NRF_OK;
    const int64_t n = m->n_params, next = n + NERF_DERIVED;
    if (next >= (int64_t)WM_RADIX_LO * WM_RADIX_LO) return NRF_OK;
    struct Images { std::vector<_Float16> img, img2; std::vector<float> bias, sig; size_t f16_at = 0, f16_floats = 0; bool ok = false; };
    auto images = [&](const float *ext, Images &o) {
        o.ok = nerf_f16_images_host(d, ext, ext + n, ext + n + (size_t)NERF_MERGED_ROWS * NERF_MERGED_W, o.img, o.img2, o.bias) &&
               nerf_sigma_image_host(d, ext, ext + n, ext + n + (size_t)NERF_MERGED_ROWS * NERF_MERGED_W, o.sig, o.f16_at, o.f16_floats);
    };
    // the real extended blob: the host's merged block
    std::vector<float> ext((size_t)next);
    memcpy(ext.data(), hp.data(), (size_t)n * 4);
    {
        std::vector<float> merged, merged_b;
        const size_t ov = m->layers[8].w_off, of = m->layers[9].w_off;
        nerf_merged_views_host(hp.data() + ov, m->layers[8].in, hp.data() + of, hp.data() + of + (size_t)256 * 256, hp.data() + ov + (size_t)m->layers[8].in * 128, NERF_MERGED_ROWS, NERF_MERGED_W,
                               merged, merged_b);
        memcpy(ext.data() + n, merged.data(), merged.size() * 4);
        memcpy(ext.data() + n + merged.size(), merged_b.data(), merged_b.size() * 4);
    }
    Images real, probe[4];
    images(ext.data(), real);
    if (!real.ok) return NRF_OK;
    for (int q = 0; q < 4; q++) { const std::vector<float> pv = weight_map_probe(next, q); images(pv.data(), probe[q]); if (!probe[q].ok) return NRF_OK; }
    // regions: (bytes of the real image, bytes of the four probes, element size, destination)
    struct Region { std::vector<uint8_t> real, p[4]; int elem; void *dst; };
    std::vector<Region> regs;
    auto add = [&](auto bytes_of /* (const Images &) -> vector<uint8_t> */, int elem, void *dst) {
        Region r; r.elem = elem; r.dst = dst; r.real = bytes_of(real);
        for (int q = 0; q < 4; q++) r.p[q] = bytes_of(probe[q]);
        regs.push_back(std::move(r));
    };
    char *f16 = static_cast<char *>(m->d_packed_f16), *spl = static_cast<char *>(m->d_packed_split), *sig = static_cast<char *>(m->d_packed_sigma_f32);
    const size_t img_b = real.img.size() * 2, img2_b = real.img2.size() * 2, bias_b = real.bias.size() * 4;
    if (m->packed_f16_bytes != img_b + bias_b || m->packed_split_bytes != img2_b + bias_b || m->packed_sigma_f32_bytes != real.sig.size() * 4) return NRF_OK;
    add([](const Images &o) { return wm_bytes(o.img.data(), o.img.size() * 2); }, 2, f16);
    add([](const Images &o) { return wm_bytes(o.bias.data(), o.bias.size() * 4); }, 4, f16 + img_b);
    add([](const Images &o) { return wm_bytes(o.img2.data(), o.img2.size() * 2); }, 2, spl);
    add([](const Images &o) { return wm_bytes(o.bias.data(), o.bias.size() * 4); }, 4, spl + img2_b);
    const size_t a0 = real.f16_at, a1 = real.f16_at + real.f16_floats;
    for (int q = 0; q < 4; q++) if (probe[q].f16_at != a0 || probe[q].f16_floats != real.f16_floats || probe[q].sig.size() != real.sig.size()) return NRF_OK;
    add([&](const Images &o) { return wm_bytes(o.sig.data(), a0 * 4); }, 4, sig);
    add([&](const Images &o) { return wm_bytes(o.sig.data() + a0, (a1 - a0) * 4); }, 2, sig + a0 * 4);
    add([&](const Images &o) { return wm_bytes(o.sig.data() + a1, (o.sig.size() - a1) * 4); }, 4, sig + a1 * 4);
    std::vector<HostMap> hms(regs.size());
    for (size_t i = 0; i < regs.size(); i++) {
        const std::vector<uint8_t> four[4] = { regs[i].p[0], regs[i].p[1], regs[i].p[2], regs[i].p[3] };
        if (!decode_weight_map(four, regs[i].elem, next, hms[i]) || !weight_map_reproduces(hms[i], ext, regs[i].real)) return NRF_OK;
    }
    // re-home the blob with room for the derived block, derive it on the device and check it against the host's bits
    float *big = nullptr;
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&big), (size_t)next * 4));
    NRF_HIP(hipMemcpy(big, m->d_params, (size_t)n * 4, hipMemcpyDeviceToDevice));
    (void)hipFree(m->d_params);
    m->d_params = big;
    NRF_TRY(nerf_derive_on_device(m, nullptr));
    std::vector<float> dev_derived((size_t)NERF_DERIVED);
    NRF_HIP(hipMemcpy(dev_derived.data(), m->d_params + n, dev_derived.size() * 4, hipMemcpyDeviceToHost));
    if (memcmp(dev_derived.data(), ext.data() + n, dev_derived.size() * 4) != 0) return NRF_OK;          // (maps stay empty: host repack)
    for (size_t i = 0; i < regs.size(); i++) NRF_TRY(upload_weight_map(m, hms[i], regs[i].dst));
    return NRF_OK;
}

// wt[k][o] = w[o][k]   (W [out][in] of the parameter blob -> W^T [in][out])
__global__ void k_transpose_wt(int in, int out, const float *__restrict__ w, float *__restrict__ wt)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)in * out) return;
    const int k = (int)(e / out), o = (int)(e - (int64_t)k * out);
    wt[e] = w[(size_t)o * in + k];
}

__global__ void k_apply_weight_map(int64_t n, const int32_t *__restrict__ src, const uint8_t *__restrict__ kind, const float *__restrict__ params, void *__restrict__ out, int elem)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int k = kind[e];
    const float v = k == WM_ZERO ? 0.0f : params[src[e]];
    if (elem == 4) { static_cast<float *>(out)[e] = v; return; }
    __half h = __float2half_rn(v);
    if (k == WM_F16_LO) h = __float2half_rn(v - __half2float(h));
    static_cast<__half *>(out)[e] = h;
}


// ---------------------------------------------------------------------------------------------------
// range-safe split precision: the scale groups of a NeRFSmall blob (mlp.h, SMALL_MAX_GROUPS)
// ---------------------------------------------------------------------------------------------------
// One workgroup.  Per group: sum of squares and absolute maximum of its weights (strided partials combined in thread order: the same blob always gives the same scales),
// then thread 0 walks the two nets with an RMS model of the activations -- a layer with RMS row norm r turns inputs of RMS a into pre-activations of RMS a r, a ReLU keeps
// 1 / sqrt 2 of that -- and picks the CUMULATIVE power of two that puts every layer's output near RMS 8: 2^13 of head-room below the fp16 maximum, and every value down
// to 1 / 64 of the RMS still has a normal (lo) half.  The weights follow: a group's own exponent is the difference of consecutive cumulative ones, capped so that its
// largest scaled weight stays below 2^15.  Nothing here is exact arithmetic and nothing needs to be: ANY powers of two give the same fp32 result up to the (hi, lo)
// roundings they are chosen to make harmless; an activation that still leaves the fp16 range is caught by the render path's non-finite word (render.hip).
// Dead band: a layer whose weights and predicted activations already sit in the comfortable range keeps the exponent 0 -- a checkpoint of ordinary magnitudes renders
// exactly as it did before there was any scaling; a layer outside the band is moved to the target.
__global__ void __launch_bounds__(256) k_small_scales(nrf_mlp_small_desc d, const float *__restrict__ w, float in_rms_hint, const float *__restrict__ in_rms_src,
                                                      float *__restrict__ gscale, float *__restrict__ scales)
{
    float in_rms = in_rms_hint;
    if (in_rms_src) { const float v = *in_rms_src; if (v > 0.0f && isfinite(v)) in_rms = v; }
    __shared__ float s_a[256], s_b[256];
    __shared__ float g_sq[SMALL_MAX_GROUPS], g_mx[SMALL_MAX_GROUPS];
    const int t = threadIdx.x;
    const int NL = d.num_layers, NLC = d.num_layers_color, V = d.input_ch_views;
    // every thread's v -> thread 0's result, in a FIXED order (a butterfly inside each wave, then the four waves' results in wave order): the same blob always gives
    // the same scales.  (Thread 0 adding the 256 partials one by one -- sixteen times -- was 130 us of a 6 ms training step.)
    auto combine = [&](float v, bool is_max) -> float {
        float r = v;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const float x = __shfl_xor(r, o); r = is_max ? fmaxf(r, x) : r + x; }
        if ((t & 63) == 0) s_a[t >> 6] = r;
        __syncthreads();
        float res = 0.0f;
        if (t == 0) res = is_max ? fmaxf(fmaxf(s_a[0], s_a[1]), fmaxf(s_a[2], s_a[3])) : (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        __syncthreads();
        return res;
    };
    (void)s_b;
    size_t off = 0;
    for (int L = 0; L < NL + NLC; L++) {
        const bool col = L >= NL;
        const int l = col ? L - NL : L;
        const int in = col ? (l == 0 ? V + d.geo_feat_dim : d.hidden_dim_color) : (l == 0 ? d.input_ch : d.hidden_dim);
        const int out = col ? (l == NLC - 1 ? 3 : d.hidden_dim_color) : (l == NL - 1 ? 1 + d.geo_feat_dim : d.hidden_dim);
        float sq0 = 0.0f, mx0 = 0.0f, sq1 = 0.0f, mx1 = 0.0f;
        for (int i = t; i < in * out; i += 256) {
            const float v = w[off + i];
            if (col && l == 0 && (i % in) >= V) { sq1 += v * v; mx1 = fmaxf(mx1, fabsf(v)); }
            else { sq0 += v * v; mx0 = fmaxf(mx0, fabsf(v)); }
        }
        const int g = col ? (l == 0 ? NL : NL + 1 + l) : l;
        const float a = combine(sq0, false), b = combine(mx0, true);
        if (t == 0) { g_sq[g] = a; g_mx[g] = b; }
        if (col && l == 0) {
            const float a1 = combine(sq1, false), b1 = combine(mx1, true);
            if (t == 0) { g_sq[NL + 1] = a1; g_mx[NL + 1] = b1; }
        }
        off += (size_t)in * out;
    }
    if (t != 0) return;
    const float TARGET = 8.0f, RELU_KEEPS = 0.70710678f;
    auto cap = [&](int e, int g, int shift) {                    // largest scaled weight of group g (exponent e + shift) below 2^15
        if (!(g_mx[g] > 0.0f) || !isfinite(g_mx[g])) return e;
        const int hi = 14 - ilogbf(g_mx[g]) - shift;
        return e < hi ? e : hi;
    };
    auto want_cum = [&](float an, int fallback) {                // cumulative exponent that brings an activation RMS `an` to TARGET
        if (!(an > 0.0f) || !isfinite(an)) return fallback;
        const float e = roundf(log2f(TARGET / an));
        return (int)fminf(fmaxf(e, -100.0f), 100.0f);
    };
    // exponent 0 is fine for group g (its weights enter the image times 2^shift) when the layer's output, at the cumulative scale 2^S it would then carry, has an RMS in
    // [2^-4, 2^9] and the group's largest weight lies in [2^-4, 2^10]
    auto comfortable = [&](float an, int S_, int g, int shift) {
        if (!(an > 0.0f) || !isfinite(an) || !(g_mx[g] > 0.0f) || !isfinite(g_mx[g])) return false;
        const float act = ldexpf(an, S_), wmax = ldexpf(g_mx[g], shift);
        return act >= 0.0625f && act <= 512.0f && wmax >= 0.0625f && wmax <= 1024.0f;
    };
    // ... and a floor under the weights: an exponent chosen for the activations alone can leave a layer's weights tiny (a large input RMS in front of a small target: the
    // last layer behind hidden layers that sit high in the band) -- then their (lo) halves are subnormal again.  The largest scaled weight of the group is kept at or
    // above 2^-4 as long as the layer's output RMS stays below 2^11.
    auto weight_floor = [&](int e, float an, int S_, int g, int shift) {
        if (!(g_mx[g] > 0.0f) || !isfinite(g_mx[g]) || !(an > 0.0f) || !isfinite(an)) return e;
        const int lo = -4 - ilogbf(g_mx[g]) - shift;
        if (e >= lo) return e;
        const int room = (int)floorf(log2f(2048.0f / an)) - S_;          // largest exponent that keeps the output RMS below 2^11
        const int e2 = lo < room ? lo : room;
        return e2 > e ? e2 : e;
    };
    int ge[SMALL_MAX_GROUPS];
    for (int g = 0; g < SMALL_MAX_GROUPS; g++) ge[g] = 0;
    float a = in_rms > 0.0f ? in_rms : 0.25f;
    int S = 0, S_hidden = 0;
    for (int l = 0; l < NL; l++) {
        const int out = l == NL - 1 ? 1 + d.geo_feat_dim : d.hidden_dim;
        const float an = a * sqrtf(g_sq[l] / (float)out) * (l < NL - 1 ? RELU_KEEPS : 1.0f);
        int e = comfortable(an, S, l, 0) ? 0 : want_cum(an, S) - S;
        e = weight_floor(e, an, S, l, 0);
        e = cap(e, l, 0);
        ge[l] = e; S += e; a = an;
        if (l == NL - 2) S_hidden = S;
    }
    const int S_sigma = S;
    int Sc = 0;
    {
        const int out = NLC == 1 ? 3 : d.hidden_dim_color;
        const float zz = sqrtf(g_sq[NL] / (float)out * 0.25f + g_sq[NL + 1] / (float)out * a * a);          // view-direction features: RMS ~ 0.5
        const float an = zz * (NLC > 1 ? RELU_KEEPS : 1.0f);
        int e = (comfortable(an, 0, NL, 0) && comfortable(an, 0, NL + 1, -S_sigma)) ? 0 : want_cum(an, 0);
        e = weight_floor(e, an, 0, NL, 0);
        e = cap(e, NL, 0);
        e = cap(e, NL + 1, -S_sigma);
        ge[NL] = e; ge[NL + 1] = e - S_sigma; Sc = e; a = an;
    }
    for (int l = 1; l < NLC; l++) {
        const int g = NL + 1 + l, out = l == NLC - 1 ? 3 : d.hidden_dim_color;
        const float an = a * sqrtf(g_sq[g] / (float)out) * (l < NLC - 1 ? RELU_KEEPS : 1.0f);
        int e = comfortable(an, Sc, g, 0) ? 0 : want_cum(an, Sc) - Sc;
        e = weight_floor(e, an, Sc, g, 0);
        e = cap(e, g, 0);
        ge[g] = e; Sc += e; a = an;
    }
    bool sane = true;
    for (int g = 0; g < NL + NLC + 1; g++) if (!isfinite(g_sq[g])) sane = false;
    for (int g = 0; g < SMALL_MAX_GROUPS; g++) gscale[g] = sane ? ldexpf(1.0f, ge[g] < -120 ? -120 : (ge[g] > 120 ? 120 : ge[g])) : 1.0f;
    for (int i = 0; i < SMALL_SCALE_COUNT; i++) scales[i] = 1.0f;
    if (sane) {
        scales[SMALL_SCALE_INV_SIGMA] = ldexpf(1.0f, -S_sigma);
        scales[SMALL_SCALE_INV_RGB] = ldexpf(1.0f, -Sc);
        scales[SMALL_SCALE_HIDDEN] = ldexpf(1.0f, S_hidden);
    }
}

__global__ void k_scale_blob(int64_t n, const float *__restrict__ w, const uint8_t *__restrict__ group, const float *__restrict__ gscale, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = w[i] * gscale[group[i]];
}

// scales of the CURRENT blob, the scaled blob, and the images that gather from it -- all in stream order, nothing waits
int mlp_small_rescale(nrf_mlp *m, hipStream_t st)
{
    if (!m || m->family != MLP_SMALL || m->maps.empty() || !m->d_group) return NRF_OK;
    if (!m->split_scaling) {
        const float ones[SMALL_MAX_GROUPS > SMALL_SCALE_COUNT ? SMALL_MAX_GROUPS : SMALL_SCALE_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        NRF_HIP(hipMemcpyAsync(m->d_params_scaled, m->d_params, (size_t)m->n_params * 4, hipMemcpyDeviceToDevice, st));
        NRF_HIP(hipMemcpyAsync(m->d_gscale, ones, SMALL_MAX_GROUPS * sizeof(float), hipMemcpyHostToDevice, st));
        NRF_HIP(hipMemcpyAsync(m->d_scales, ones, SMALL_SCALE_COUNT * sizeof(float), hipMemcpyHostToDevice, st));
        NRF_HIP(hipStreamSynchronize(st));            // (`ones` is a stack array)
    } else {
        k_small_scales<<<dim3(1), dim3(256), 0, st>>>(m->small, m->d_params, m->in_rms_hint, m->d_in_rms_src, m->d_gscale, m->d_scales);
        NRF_HIP(hipGetLastError());
        k_scale_blob<<<dim3((unsigned)((m->n_params + 255) / 256)), dim3(256), 0, st>>>(m->n_params, m->d_params, m->d_group, m->d_gscale, m->d_params_scaled);
        NRF_HIP(hipGetLastError());
    }
    for (auto &w : m->maps) {
        if (!w.scaled) continue;
        k_apply_weight_map<<<dim3((unsigned)((w.n + 255) / 256)), dim3(256), 0, st>>>(w.n, w.d_src, w.d_kind, m->d_params_scaled, w.d_out, w.elem);
        NRF_HIP(hipGetLastError());
    }
    return NRF_OK;
}

// the scale machinery of a freshly built NeRFSmall handle: group table, buffers, identity scales (kernels read d_scales whether or not scaling is in force)
static int small_scale_setup(nrf_mlp *m)
{
    const auto &d = m->small;
    std::vector<float> ones(SMALL_MAX_GROUPS > SMALL_SCALE_COUNT ? SMALL_MAX_GROUPS : SMALL_SCALE_COUNT, 1.0f);
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_scales), SMALL_SCALE_COUNT * sizeof(float)));
    NRF_HIP(hipMemcpy(m->d_scales, ones.data(), SMALL_SCALE_COUNT * sizeof(float), hipMemcpyHostToDevice));
    if (m->maps.empty() || d.use_pred_normal || d.num_layers + d.num_layers_color + 1 > SMALL_MAX_GROUPS) return NRF_OK;
    std::vector<uint8_t> group((size_t)m->n_params, 0);
    size_t off = 0;
    for (int l = 0; l < d.num_layers; l++) {
        const size_t ne = (size_t)(l == 0 ? d.input_ch : d.hidden_dim) * (l == d.num_layers - 1 ? 1 + d.geo_feat_dim : d.hidden_dim);
        for (size_t i = 0; i < ne; i++) group[off + i] = (uint8_t)l;
        off += ne;
    }
    for (int l = 0; l < d.num_layers_color; l++) {
        const int in = l == 0 ? d.input_ch_views + d.geo_feat_dim : d.hidden_dim_color, out = l == d.num_layers_color - 1 ? 3 : d.hidden_dim_color;
        for (size_t i = 0; i < (size_t)in * out; i++)
            group[off + i] = (uint8_t)(l == 0 ? ((int)(i % in) >= d.input_ch_views ? d.num_layers + 1 : d.num_layers) : d.num_layers + 1 + l);
        off += (size_t)in * out;
    }
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_group), group.size()));
    NRF_HIP(hipMemcpy(m->d_group, group.data(), group.size(), hipMemcpyHostToDevice));
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_gscale), SMALL_MAX_GROUPS * sizeof(float)));
    NRF_HIP(hipMemcpy(m->d_gscale, ones.data(), SMALL_MAX_GROUPS * sizeof(float), hipMemcpyHostToDevice));
    NRF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_params_scaled), (size_t)m->n_params * 4));
    if (const char *e = getenv("NRF_SPLIT_UNSCALED")) m->split_scaling = atoi(e) == 0;          // A/B switch of the default
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int64_t nrf_mlp_small_param_count(const nrf_mlp_small_desc *d)
{
    if (!d) return 0;
    int64_t n = 0;
    for (int l = 0; l < d->num_layers; l++) n += (int64_t)((l == 0) ? d->input_ch : d->hidden_dim) * ((l == d->num_layers - 1) ? (1 + d->geo_feat_dim) : d->hidden_dim);
    for (int l = 0; l < d->num_layers_color; l++)
        n += (int64_t)((l == 0) ? d->input_ch_views + d->geo_feat_dim : d->hidden_dim_color) * ((l == d->num_layers_color - 1) ? 3 : d->hidden_dim_color);
    if (d->use_pred_normal)                                               // NeRF.cpp:343-347
        for (int l = 0; l < d->num_layers_normals; l++)
            n += (int64_t)((l == 0) ? 1 + d->geo_feat_dim + d->input_ch : d->hidden_dim_normals) * ((l == d->num_layers_normals - 1) ? 3 : d->hidden_dim_normals);
    return n;
}

int64_t nrf_mlp_nerf_param_count(const nrf_mlp_nerf_desc *d)
{
    if (!d) return 0;
    const int w = d->width;
    int64_t n = (int64_t)d->input_ch * w + w;
    for (int i = 0; i < d->depth - 1; i++) n += (int64_t)((i == d->skip) ? (w + d->input_ch) : w) * w + w;
    if (d->use_viewdirs) n += (int64_t)(d->input_ch_views + w) * (w / 2) + w / 2 + (int64_t)w * w + w + w + 1 + (int64_t)(w / 2) * 3 + 3;
    else n += (int64_t)(w + d->input_ch) * d->output_ch + d->output_ch;
    return n;
}

int nrf_mlp_small_create(const nrf_mlp_small_desc *d, const float *params, int params_on_device, void *stream, nrf_mlp **out)
{
    NRF_CHECK_ARG(d && params && out, "nrf_mlp_small_create: null pointer");
    NRF_CHECK_ARG(d->input_ch > 0 && d->input_ch_views >= 0 && d->num_layers >= 1 && d->hidden_dim > 0 && d->geo_feat_dim >= 0 &&
                  d->num_layers_color >= 1 && d->hidden_dim_color > 0, "nrf_mlp_small_create: bad dimensions");
    nrf_mlp *m = new nrf_mlp();
    NRF_CHECK_ARG(!d->use_pred_normal || (d->num_layers_normals >= 1 && d->hidden_dim_normals > 0), "nrf_mlp_small_create: use_pred_normal needs num_layers_normals >= 1 and hidden_dim_normals > 0");
    m->family = MLP_SMALL; m->small = *d;
    m->in_dims = d->input_ch + d->input_ch_views; m->out_dims = d->use_pred_normal ? 7 : 4;
    std::vector<float> hp;
    int s = fetch_params(params, params_on_device, nrf_mlp_small_param_count(d), as_stream(stream), hp, m);
    size_t off = 0;
    for (int l = 0; l < d->num_layers && s == NRF_OK; l++)
        s = add_layer(m, hp, off, (l == 0) ? d->input_ch : d->hidden_dim, (l == d->num_layers - 1) ? (1 + d->geo_feat_dim) : d->hidden_dim, false);
    for (int l = 0; l < d->num_layers_color && s == NRF_OK; l++)
        s = add_layer(m, hp, off, (l == 0) ? d->input_ch_views + d->geo_feat_dim : d->hidden_dim_color, (l == d->num_layers_color - 1) ? 3 : d->hidden_dim_color, false);
    for (int l = 0; d->use_pred_normal && l < d->num_layers_normals && s == NRF_OK; l++)
        s = add_layer(m, hp, off, (l == 0) ? 1 + d->geo_feat_dim + d->input_ch : d->hidden_dim_normals, (l == d->num_layers_normals - 1) ? 3 : d->hidden_dim_normals, false);
    // the matrix-core images describe the two-net model: a handle with the normals head keeps to the fp32 layer kernels
    if (s == NRF_OK && !d->use_pred_normal) s = mlp_small_pack_f16(m, hp);
    if (s == NRF_OK && !d->use_pred_normal) s = mlp_small_pack_sigma_f32(m, hp);
    if (s == NRF_OK && !d->use_pred_normal) s = build_weight_maps(m, hp);
    if (s == NRF_OK) s = small_scale_setup(m);
    if (s == NRF_OK) s = mlp_small_rescale(m, as_stream(stream));
    if (s == NRF_OK) s = hipStreamSynchronize(as_stream(stream)) == hipSuccess ? NRF_OK : NRF_ERR_HIP;
    if (s != NRF_OK) { nrf_mlp_destroy(m); return s; }
    *out = m;
    return NRF_OK;
}

int nrf_mlp_nerf_create(const nrf_mlp_nerf_desc *d, const float *params, int params_on_device, void *stream, nrf_mlp **out)
{
    NRF_CHECK_ARG(d && params && out, "nrf_mlp_nerf_create: null pointer");
    NRF_CHECK_ARG(d->depth >= 2 && d->width >= 2 && d->input_ch > 0 && d->skip >= -1 && d->skip < d->depth - 1, "nrf_mlp_nerf_create: bad dimensions");
    NRF_CHECK_ARG(d->use_viewdirs ? d->input_ch_views > 0 : d->output_ch > 0, "nrf_mlp_nerf_create: bad head dimensions");
    nrf_mlp *m = new nrf_mlp();
    m->family = MLP_NERF; m->nerf = *d;
    m->in_dims = d->input_ch + (d->use_viewdirs ? d->input_ch_views : 0);
    m->out_dims = d->use_viewdirs ? 4 : d->output_ch;
    std::vector<float> hp;
    int s = fetch_params(params, params_on_device, nrf_mlp_nerf_param_count(d), as_stream(stream), hp, m);
    size_t off = 0;
    const int w = d->width;
    if (s == NRF_OK) s = add_layer(m, hp, off, d->input_ch, w, true);
    for (int i = 0; i < d->depth - 1 && s == NRF_OK; i++) s = add_layer(m, hp, off, (i == d->skip) ? (w + d->input_ch) : w, w, true);
    if (d->use_viewdirs) {
        // blob order: views_linears_0, feature_linear, alpha_linear, rgb_linear (NeRF.cpp:78-88)
        if (s == NRF_OK) s = add_layer(m, hp, off, d->input_ch_views + w, w / 2, true);
        if (s == NRF_OK) s = add_layer(m, hp, off, w, w, true);
        if (s == NRF_OK) s = add_layer(m, hp, off, w, 1, true);
        if (s == NRF_OK) s = add_layer(m, hp, off, w / 2, 3, true);
    } else if (s == NRF_OK) s = add_layer(m, hp, off, w + d->input_ch, d->output_ch, true);
    if (w + d->input_ch_views > m->max_width) m->max_width = w + d->input_ch_views;
    if (s == NRF_OK) s = mlp_nerf_pack_f16(m, hp);
    if (s == NRF_OK) s = mlp_nerf_pack_sigma_f32(m, hp);
    if (s == NRF_OK) s = build_weight_maps_nerf(m, hp);
    if (s != NRF_OK) { nrf_mlp_destroy(m); return s; }
    *out = m;
    return NRF_OK;
}

// The LeRF head's device packers (mlp_lerf_pack_f16_device, mlp_lerf_pack_sigma_f32_device) against the host packers' images, which the handle holds at this point:
// the three images are read back, rebuilt on the device from m->d_params and read back again; equal byte for byte (and the same Gram scale) -> nrf_mlp_set_params keeps
// this handle's parameter uploads on the device.  Anything else -> the host images are restored and the host packers stay in charge.  NRF_MLP_HOST_REPACK=1 skips it.
static int lerf_device_pack_verify(nrf_mlp *m, hipStream_t st)
{
    m->lerf_device_pack = false;
    if (const char *e = getenv("NRF_MLP_HOST_REPACK")) if (atoi(e) != 0) return NRF_OK;
    if (!m->d_packed_f16 || !m->d_packed_split || !m->d_packed_sigma_f32) return NRF_OK;
    void *dev[3] = { m->d_packed_f16, m->d_packed_split, m->d_packed_sigma_f32 };
    const size_t bytes[3] = { m->packed_f16_bytes, m->packed_split_bytes, m->packed_sigma_f32_bytes };
    std::vector<uint8_t> host[3], mine[3];
    NRF_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 3; i++) { host[i].resize(bytes[i]); NRF_HIP(hipMemcpy(host[i].data(), dev[i], bytes[i], hipMemcpyDeviceToHost)); }
    const float scale = m->lerf_gram_scale;
    bool same = mlp_lerf_pack_f16_device(m, st) == NRF_OK && mlp_lerf_pack_sigma_f32_device(m, st) == NRF_OK;
    if (same) {
        NRF_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < 3 && same; i++) {
            mine[i].resize(bytes[i]);
            NRF_HIP(hipMemcpy(mine[i].data(), dev[i], bytes[i], hipMemcpyDeviceToHost));
            same = memcmp(mine[i].data(), host[i].data(), bytes[i]) == 0;
        }
        same = same && m->lerf_gram_scale == scale;
    }
    if (!same) {
        for (int i = 0; i < 3; i++) NRF_HIP(hipMemcpy(dev[i], host[i].data(), bytes[i], hipMemcpyHostToDevice));
        m->lerf_gram_scale = scale;
    }
    m->lerf_device_pack = same;
    return NRF_OK;
}

/* LeRFImpl ctor (LeRF.cpp:3-26): sigma net in->H..->1+geo, LE net (geo+in)->H..->embed, both `num_layers` deep, bias-free.
 * Described with nrf_mlp_small_desc: input_ch, num_layers, hidden_dim, geo_feat_dim, hidden_dim_color = lang_embed_dim. */
NRF_API int64_t nrf_mlp_lerf_param_count(const nrf_mlp_small_desc *d)
{
    if (!d) return 0;
    int64_t n = 0;
    for (int l = 0; l < d->num_layers; l++) n += (int64_t)((l == 0) ? d->input_ch : d->hidden_dim) * ((l == d->num_layers - 1) ? (1 + d->geo_feat_dim) : d->hidden_dim);
    for (int l = 0; l < d->num_layers; l++) n += (int64_t)((l == 0) ? d->geo_feat_dim + d->input_ch : d->hidden_dim) * ((l == d->num_layers - 1) ? d->hidden_dim_color : d->hidden_dim);
    return n;
}

NRF_API int nrf_mlp_lerf_create(const nrf_mlp_small_desc *d, const float *params, int params_on_device, void *stream, nrf_mlp **out)
{
    NRF_CHECK_ARG(d && params && out, "nrf_mlp_lerf_create: null pointer");
    NRF_CHECK_ARG(d->input_ch > 0 && d->num_layers >= 1 && d->hidden_dim > 0 && d->geo_feat_dim >= 0 && d->hidden_dim_color > 0, "nrf_mlp_lerf_create: bad dimensions");
    nrf_mlp *m = new nrf_mlp();
    m->family = MLP_LERF; m->small = *d;
    m->in_dims = d->input_ch; m->out_dims = d->hidden_dim_color + 1;
    std::vector<float> hp;
    int s = fetch_params(params, params_on_device, nrf_mlp_lerf_param_count(d), as_stream(stream), hp, m);
    size_t off = 0;
    for (int l = 0; l < d->num_layers && s == NRF_OK; l++)
        s = add_layer(m, hp, off, (l == 0) ? d->input_ch : d->hidden_dim, (l == d->num_layers - 1) ? (1 + d->geo_feat_dim) : d->hidden_dim, false);
    for (int l = 0; l < d->num_layers && s == NRF_OK; l++)
        s = add_layer(m, hp, off, (l == 0) ? d->geo_feat_dim + d->input_ch : d->hidden_dim, (l == d->num_layers - 1) ? d->hidden_dim_color : d->hidden_dim, false);
    if (s == NRF_OK) s = mlp_lerf_pack_f16(m, hp);
    if (s == NRF_OK) s = mlp_lerf_pack_sigma_f32(m, hp);
    if (s == NRF_OK) s = lerf_device_pack_verify(m, as_stream(stream));
    if (s != NRF_OK) { nrf_mlp_destroy(m); return s; }
    *out = m;
    return NRF_OK;
}

/* Training: replace the parameter blob (same layout) and refresh every derived operand (transposed fp32 layers, matrix-core images). */
int nrf_mlp_set_params(nrf_mlp *m, const float *params, int params_on_device, void *stream)
{
    NRF_CHECK_ARG(m && params, "nrf_mlp_set_params: null pointer");
    hipStream_t st = as_stream(stream);
    if (!m->maps.empty()) {
        // NeRFSmall: blob and every derived image refreshed on the device, in stream order, nothing waits.  Work issued earlier on `stream` (and on the library's
        // lanes, which are joined to it before a render call returns) still reads the old images and finishes first; another stream of the caller's must be ordered
        // by the caller, as for any in-place update.
        NRF_HIP(hipMemcpyAsync(m->d_params, params, (size_t)m->n_params * 4, params_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        if (m->family == MLP_NERF) {
            // classic network: W^T and bias of every layer (the fp32 kernels' operands), then the merged views layer behind the blob: the maps below gather from both
            for (auto &L : m->layers) {
                const int64_t ne = (int64_t)L.in * L.out;
                k_transpose_wt<<<dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st>>>(L.in, L.out, m->d_params + L.w_off, L.d_wt);
                NRF_HIP(hipGetLastError());
                if (L.d_bias) NRF_HIP(hipMemcpyAsync(L.d_bias, m->d_params + L.w_off + (size_t)L.in * L.out, (size_t)L.out * 4, hipMemcpyDeviceToDevice, st));
            }
            NRF_TRY(nerf_derive_on_device(m, st));
        }
        for (auto &w : m->maps) {
            if (w.scaled && m->d_group) continue;          // the split-precision operands: below, from the range-scaled blob
            k_apply_weight_map<<<dim3((unsigned)((w.n + 255) / 256)), dim3(256), 0, st>>>(w.n, w.d_src, w.d_kind, m->d_params, w.d_out, w.elem);
            NRF_HIP(hipGetLastError());
        }
        return mlp_small_rescale(m, st);
    }
    if (m->family == MLP_LERF && m->lerf_device_pack && params_on_device) {
        // the LeRF head: blob, transposed layers and the matrix-core images all on the device, in stream order (one 4-byte read-back inside: the Gram matrix's scale)
        NRF_HIP(hipMemcpyAsync(m->d_params, params, (size_t)m->n_params * 4, hipMemcpyDeviceToDevice, st));
        for (auto &L : m->layers) {
            const int64_t ne = (int64_t)L.in * L.out;
            k_transpose_wt<<<dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st>>>(L.in, L.out, m->d_params + L.w_off, L.d_wt);
            NRF_HIP(hipGetLastError());
        }
        NRF_TRY(mlp_lerf_pack_f16_device(m, st));
        return mlp_lerf_pack_sigma_f32_device(m, st);
    }
    std::vector<float> hp((size_t)m->n_params);
    if (params_on_device) {
        NRF_HIP(hipMemcpyAsync(hp.data(), params, hp.size() * 4, hipMemcpyDeviceToHost, st));
        NRF_HIP(hipStreamSynchronize(st));
    } else {
        memcpy(hp.data(), params, hp.size() * 4);
        NRF_HIP(hipStreamSynchronize(st));             // the derived images are overwritten in place: nothing may still be reading them
    }
    NRF_HIP(hipMemcpyAsync(m->d_params, hp.data(), hp.size() * 4, hipMemcpyHostToDevice, st));
    // W^T [in][out] and the bias of every layer (the generic fp32 kernels' operands) from the uploaded blob, on the device: a training loop comes here every step, and the
    // host transposes + two small uploads per layer were most of what was left of this call (classic 8x256: 12 layers)
    for (auto &L : m->layers) {
        const int64_t ne = (int64_t)L.in * L.out;
        k_transpose_wt<<<dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st>>>(L.in, L.out, m->d_params + L.w_off, L.d_wt);
        NRF_HIP(hipGetLastError());
        if (L.d_bias) NRF_HIP(hipMemcpyAsync(L.d_bias, m->d_params + L.w_off + (size_t)L.in * L.out, (size_t)L.out * 4, hipMemcpyDeviceToDevice, st));
    }
    NRF_HIP(hipStreamSynchronize(st));
    if (m->family == MLP_SMALL) { if (m->small.use_pred_normal) return NRF_OK; NRF_TRY(mlp_small_pack_f16(m, hp)); return mlp_small_pack_sigma_f32(m, hp); }
    if (m->family == MLP_NERF) { NRF_TRY(mlp_nerf_pack_f16(m, hp)); return mlp_nerf_pack_sigma_f32(m, hp); }
    if (m->family == MLP_LERF) { NRF_TRY(mlp_lerf_pack_f16(m, hp)); return mlp_lerf_pack_sigma_f32(m, hp); }
    return NRF_OK;
}

int nrf_mlp_device_repack_images(const nrf_mlp *m) { return m ? (m->family == MLP_LERF ? (m->lerf_device_pack ? 3 : 0) : (int)m->maps.size()) : 0; }

int nrf_mlp_set_input_rms_hint(nrf_mlp *m, float rms, void *stream)
{
    NRF_CHECK_ARG(m && rms > 0.0f && rms < 1e30f, "nrf_mlp_set_input_rms_hint: bad argument");
    m->in_rms_hint = rms;
    return mlp_small_rescale(m, as_stream(stream));
}

int nrf_mlp_set_split_scaling(nrf_mlp *m, int on, void *stream)
{
    NRF_CHECK_ARG(m, "nrf_mlp_set_split_scaling: null pointer");
    m->split_scaling = on != 0;
    return mlp_small_rescale(m, as_stream(stream));
}

int nrf_mlp_get_split_scales(const nrf_mlp *m, float *group_scales_out, float *kernel_scales_out, void *stream)
{
    NRF_CHECK_ARG(m && group_scales_out && kernel_scales_out, "nrf_mlp_get_split_scales: null pointer");
    for (int i = 0; i < SMALL_MAX_GROUPS; i++) group_scales_out[i] = 1.0f;
    for (int i = 0; i < SMALL_SCALE_COUNT; i++) kernel_scales_out[i] = 1.0f;
    if (m->family != MLP_SMALL || !m->d_scales) return NRF_OK;
    hipStream_t st = as_stream(stream);
    if (m->d_gscale) NRF_HIP(hipMemcpyAsync(group_scales_out, m->d_gscale, SMALL_MAX_GROUPS * sizeof(float), hipMemcpyDeviceToHost, st));
    NRF_HIP(hipMemcpyAsync(kernel_scales_out, m->d_scales, SMALL_SCALE_COUNT * sizeof(float), hipMemcpyDeviceToHost, st));
    NRF_HIP(hipStreamSynchronize(st));
    return NRF_OK;
}

size_t nrf_mlp_backward_workspace_bytes(const nrf_mlp *m, int64_t p)
{
    if (!m) return 0;
    return m->family == MLP_NERF ? mlp_nerf_backward_workspace_bytes(m, p) : mlp_backward_workspace_bytes(m, p);
}

int nrf_mlp_backward(const nrf_mlp *m, const float *d_x, const float *d_g_out, int64_t p, float *d_g_params, float *d_g_x, void *d_workspace,
                     size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(m && d_x && d_g_out && d_g_params && d_workspace && p >= 0, "nrf_mlp_backward: bad argument");
    if (p == 0) return NRF_OK;
    if (m->family == MLP_NERF)          // NeRFImpl (NeRF.cpp:92-126); d_g_x [p, input_ch] = d loss / d input_pts
        return mlp_nerf_backward(m, d_x, m->in_dims, d_g_out, m->out_dims, p, d_g_params, d_g_x, m->nerf.input_ch, d_workspace, workspace_bytes, as_stream(stream));
    return mlp_small_backward(m, d_x, m->in_dims, d_g_out, m->out_dims, p, d_g_params, d_g_x, m->small.input_ch, d_workspace, workspace_bytes, as_stream(stream));
}

size_t nrf_mlp_backward_f16_workspace_bytes(const nrf_mlp *m, int64_t p) { return (m && m->family == MLP_SMALL) ? mlp_small_backward_mfma_workspace_bytes(m, p) : 0; }

int nrf_mlp_backward_f16(const nrf_mlp *m, const float *d_x, const float *d_g_out, int64_t p, float *d_g_params, float *d_g_x, void *d_workspace,
                         size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(m && d_x && d_g_out && d_g_params && d_workspace && p >= 0, "nrf_mlp_backward_f16: bad argument");
    if (p == 0) return NRF_OK;
    return mlp_small_backward_mfma(m, d_x, m->in_dims, d_g_out, m->out_dims, p, d_g_params, d_g_x, m->small.input_ch, d_workspace, workspace_bytes, as_stream(stream));
}

int nrf_mlp_backward_f16_lm(const nrf_mlp *m, const void *d_feats_lm, const void *d_dirs_f16, int s, const float *d_g_out, int64_t p, float *d_g_params, float *d_g_x,
                            void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_dirs_f16 && d_g_out && d_g_params && d_workspace && p >= 0 && s >= 1, "nrf_mlp_backward_f16_lm: bad argument");
    if (p == 0) return NRF_OK;
    if (m->family != MLP_SMALL) { set_error("nrf_mlp_backward_f16_lm: built for the NeRFSmall family"); return NRF_ERR_UNSUPPORTED; }
    return mlp_small_backward_mfma_lm(m, reinterpret_cast<const __half2 *>(d_feats_lm), reinterpret_cast<const __half *>(d_dirs_f16), s, d_g_out, m->out_dims, p, d_g_params, d_g_x,
                                      m->small.input_ch, d_workspace, workspace_bytes, as_stream(stream));
}

// ... reading the features through a column map: point q -> column d_src[q] of a level-major table [16][pstride] (a renderer's feature-reusing fine pass keeps the
// coarse columns and the new samples' columns apart: nrf_renderer_last_features)
int nrf_mlp_backward_f16_lm_src(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const int32_t *d_src, const void *d_dirs_f16, int s, const float *d_g_out, int64_t p,
                                float *d_g_params, float *d_g_x, void *d_workspace, size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_src && pstride >= 1 && d_dirs_f16 && d_g_out && d_g_params && d_workspace && p >= 0 && s >= 1, "nrf_mlp_backward_f16_lm_src: bad argument");
    if (p == 0) return NRF_OK;
    if (m->family != MLP_SMALL) { set_error("nrf_mlp_backward_f16_lm_src: built for the NeRFSmall family"); return NRF_ERR_UNSUPPORTED; }
    return mlp_small_backward_mfma_lm(m, reinterpret_cast<const __half2 *>(d_feats_lm), reinterpret_cast<const __half *>(d_dirs_f16), s, d_g_out, m->out_dims, p, d_g_params, d_g_x,
                                      m->small.input_ch, d_workspace, workspace_bytes, as_stream(stream), pstride, d_src);
}

void nrf_mlp_destroy(nrf_mlp *m)
{
    if (!m) return;
    for (auto &L : m->layers) {
        if (L.d_wt) (void)hipFree(L.d_wt);
        if (L.d_bias) (void)hipFree(L.d_bias);
    }
    drop_weight_maps(m);
    if (m->d_params) (void)hipFree(m->d_params);
    if (m->d_packed_f16) (void)hipFree(m->d_packed_f16);
    if (m->d_packed_split) (void)hipFree(m->d_packed_split);
    if (m->d_packed_bwd) (void)hipFree(m->d_packed_bwd);
    if (m->d_packed_sigma_f32) (void)hipFree(m->d_packed_sigma_f32);
    if (m->d_lerf_gram) (void)hipFree(m->d_lerf_gram);
    if (m->d_params_scaled) (void)hipFree(m->d_params_scaled);
    if (m->d_group) (void)hipFree(m->d_group);
    if (m->d_gscale) (void)hipFree(m->d_gscale);
    if (m->d_scales) (void)hipFree(m->d_scales);
    delete m;
}

int nrf_mlp_output_dims(const nrf_mlp *m) { return m ? m->out_dims : 0; }

int nrf_mlp_forward(const nrf_mlp *m, const float *d_x, int64_t p, int precision, float *d_out, void *stream)
{
    NRF_CHECK_ARG(m && d_x && d_out && p >= 0, "nrf_mlp_forward: bad argument");
    if (p == 0) return NRF_OK;
    hipStream_t st = as_stream(stream);
    const size_t wsb = mlp_workspace_bytes(m, p, precision);
    void *ws = nullptr;
    NRF_HIP(scratch_take(&ws, wsb, st));          // stream-ordered scratch for the standalone entry point
    const int s = mlp_forward(m, d_x, m->in_dims, p, precision, d_out, m->out_dims, ws, wsb, st);
    NRF_HIP(scratch_give(ws, st));
    return s;
}

// ... without waiting: the two words are copied into h_flags2 (two uint32, best pinned) in `stream`'s order; the caller reads them once the stream (or an event recorded
// behind this call) has passed.  nrf_mlp_backward_f16_flags_device: where they live on the device (what nrf_adam_step_guarded takes).
int nrf_mlp_backward_f16_flags_async(const void *d_workspace, uint32_t *h_flags2, void *stream)
{
    NRF_CHECK_ARG(d_workspace && h_flags2, "nrf_mlp_backward_f16_flags_async: null pointer");
    NRF_HIP(hipMemcpyAsync(h_flags2, static_cast<const uint32_t *>(d_workspace) + 1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, as_stream(stream)));
    return NRF_OK;
}
const uint32_t *nrf_mlp_backward_f16_flags_device(const void *d_workspace) { return d_workspace ? static_cast<const uint32_t *>(d_workspace) + 1 : nullptr; }

int nrf_mlp_backward_f16_flags(const void *d_workspace, uint32_t *flags_out, void *stream)
{
    NRF_CHECK_ARG(d_workspace && flags_out, "nrf_mlp_backward_f16_flags: null pointer");
    uint32_t w[4] = {0, 0, 0, 0};
    NRF_HIP(hipMemcpyAsync(w, d_workspace, sizeof(w), hipMemcpyDeviceToHost, as_stream(stream)));
    NRF_HIP(hipStreamSynchronize(as_stream(stream)));
    flags_out[0] = w[1]; flags_out[1] = w[2];
    return NRF_OK;
}

}  // extern "C"
