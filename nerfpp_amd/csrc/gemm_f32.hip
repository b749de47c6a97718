// gemm_f32.hip -- the plain fp32 layer products of the TRAINING paths (the recomputed forward and the backward of the classic NeRFImpl and of the LeRF head; the fp32
// backward of NeRFSmall) as library GEMMs: rocBLAS sgemm, which runs them on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why a library here and hand-written kernels everywhere else: these are textbook row-major GEMMs with nothing to fuse (Y = X W^T, G_in = G W, dW += G^T X over 10^4..10^6
// points), carry no bit-exactness requirement (a gradient is compared with autograd at 1e-4), and are not on the render path.  The parity mode's forward (NRF_PREC_F32 ==
// the oracle's FMA chains bit for bit) never comes here: it keeps mlp.hip's k_linear.  rocBLAS is resolved with dlopen at first use -- the copy the process already maps
// (LibTorch's) if there is one -- so libnerfpp_hip.so carries no link-time dependency on it; without it, or with NRF_FP32_GEMM=0, the hand-written fp32 kernels of mlp.hip run.
#include "mlp.h"

#include <dlfcn.h>
#include <rocblas/rocblas.h>

#include <mutex>

namespace nrf {

struct RocBlas {
    void *handle = nullptr;
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocblas_sgemm) sgemm = nullptr;
    decltype(&rocblas_sgemm_strided_batched) sgemm_sb = nullptr;
    decltype(&rocblas_set_pointer_mode) set_pointer_mode = nullptr;
    decltype(&rocblas_set_atomics_mode) set_atomics_mode = nullptr;
};

static RocBlas g_rb;
static std::once_flag g_rb_once;
static std::mutex g_rb_mu;
// one handle per (host thread, device): a handle carries its stream (rocblas_set_stream), so two host threads that drive one GPU on their own streams -- data-parallel
// replicas rehearsed as threads, a host that renders on one thread and trains on another -- must not share one (their products would be enqueued on each other's streams).
// Handles live as long as the process (a host thread that ends leaves its handles behind: tens of them at most).
static thread_local rocblas_handle g_rb_handle[64] = {};

static void rb_load()
{
    if (const char *e = getenv("NRF_FP32_GEMM")) if (atoi(e) == 0) return;
    const char *names[] = {"librocblas.so.5", "librocblas.so"};
    void *h = nullptr;
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // the copy the host process already uses
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librocblas.so.5", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    RocBlas r;
    r.handle = h;
    r.create = reinterpret_cast<decltype(r.create)>(dlsym(h, "rocblas_create_handle"));
    r.set_stream = reinterpret_cast<decltype(r.set_stream)>(dlsym(h, "rocblas_set_stream"));
    r.sgemm = reinterpret_cast<decltype(r.sgemm)>(dlsym(h, "rocblas_sgemm"));
    r.sgemm_sb = reinterpret_cast<decltype(r.sgemm_sb)>(dlsym(h, "rocblas_sgemm_strided_batched"));
    r.set_pointer_mode = reinterpret_cast<decltype(r.set_pointer_mode)>(dlsym(h, "rocblas_set_pointer_mode"));
    r.set_atomics_mode = reinterpret_cast<decltype(r.set_atomics_mode)>(dlsym(h, "rocblas_set_atomics_mode"));
    if (r.create && r.set_stream && r.sgemm) g_rb = r;
}

// the device's handle bound to `st`, or nullptr (no rocBLAS: the callers fall back to the hand-written kernels)
static rocblas_handle rb_handle(hipStream_t st)
{
    std::call_once(g_rb_once, rb_load);
    if (!g_rb.handle) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_rb_mu);
    if (!g_rb_handle[dev]) {
        rocblas_handle h = nullptr;
        if (g_rb.create(&h) != rocblas_status_success) return nullptr;
        if (g_rb.set_pointer_mode) (void)g_rb.set_pointer_mode(h, rocblas_pointer_mode_host);
        g_rb_handle[dev] = h;
    }
    if (g_rb.set_stream(g_rb_handle[dev], st) != rocblas_status_success) return nullptr;
    return g_rb_handle[dev];
}

int fp32_gemm_available() { std::call_once(g_rb_once, rb_load); return g_rb.handle != nullptr; }

// Row-major views: a row-major [rows][ld] array IS the column-major matrix (cols_of_the_view x rows) with leading dimension ld.
// y[pt][y_off + o] (+)= sum_k seg[pt][k] W[o][w_col0 + k]          W: the blob's [out][in] block
static bool gemm_fwd_seg(rocblas_handle h, int64_t npts, Seg x, const float *w_blob, int in, int w_col0, int out, float beta, float *y, int y_stride, int y_off)
{
    if (x.n == 0) return true;
    const float alpha = 1.0f;
    // Y_cm (out x pts, ld y_stride) = W_cm^T (W_cm = in x out, ld in; rows w_col0..) . X_cm (x.n x pts, ld x.stride)
    return g_rb.sgemm(h, rocblas_operation_transpose, rocblas_operation_none, out, (rocblas_int)npts, x.n, &alpha, w_blob + w_col0, in, x.p + x.off, x.stride, &beta, y + y_off,
                      y_stride) == rocblas_status_success;
}

__global__ void k_bias_relu(int64_t total, int out, float *__restrict__ y, int y_stride, int y_off, const float *__restrict__ bias, int relu)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t pt = e / out; const int o = (int)(e - pt * out);
    float v = y[pt * y_stride + y_off + o];
    if (bias) v = v + bias[o];
    if (relu) v = v < 0.0f ? 0.0f : v;
    y[pt * y_stride + y_off + o] = v;
}

// Linear(+bias)(+ReLU) on cat[a, b]: the library product, then one elementwise pass where there is a bias or a ReLU.  Falls back to mlp.hip's kernel.
int run_linear_fast(int64_t npts, Seg a, Seg b, const nrf_mlp *m, const LinearLayer &L, int relu, float *y, int y_stride, int y_off, hipStream_t st, uint64_t *relu_bits,
                    int relu_bits_ld)
{
    if (const int arith = npts >= 256 ? train_gemm_for(m) : 0)          // split-precision matrix-core product, bias + ReLU in its epilogue (gemm_bf16x3.hip)
        return gemm_nt_split(arith, npts, L.out, a, b, m->d_params + L.w_off, L.in, y + y_off, y_stride, L.d_bias, relu, nullptr, 0, st, nullptr, 0, nullptr, 0, nullptr,
                             relu ? relu_bits : nullptr, relu_bits_ld);
    if (relu_bits) { set_error("internal: run_linear_fast: mask bits exist in the split-precision modes only"); return NRF_ERR_INVALID_ARG; }
    rocblas_handle h = (npts >= 256) ? rb_handle(st) : nullptr;
    if (!h || npts > 0x7fffffff) return run_linear(npts, a, b, L, relu, y, y_stride, y_off, st);
    const float *wb = m->d_params + L.w_off;
    if (!gemm_fwd_seg(h, npts, a, wb, L.in, 0, L.out, 0.0f, y, y_stride, y_off) || !gemm_fwd_seg(h, npts, b, wb, L.in, a.n, L.out, 1.0f, y, y_stride, y_off)) {
        set_error("rocblas_sgemm failed (forward layer %d -> %d over %lld points)", L.in, L.out, (long long)npts);
        return NRF_ERR_HIP;
    }
    if (L.d_bias || relu) {
        hipLaunchKernelGGL(k_bias_relu, dim3((unsigned)ceil_div(npts * L.out, 256)), dim3(256), 0, st, npts * L.out, L.out, y, y_stride, y_off, (const float *)L.d_bias, relu);
        NRF_LAUNCH_CHECK();
    }
    return NRF_OK;
}

// dw[(o) * in + col0 + i] += sum_b part[b][o][i]
__global__ void k_sum_partials(int batches, int out, int n, int in, int col0, const float *__restrict__ part, float *__restrict__ dw)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= out * n) return;
    float acc = 0.0f;
    for (int bch = 0; bch < batches; bch++) acc += part[(size_t)bch * out * n + e];
    const int o = e / n, i = e - o * n;
    dw[(size_t)o * in + col0 + i] += acc;
}

// dw[o][i] += sum_pt g[pt][o] cat[a, b][pt][i]: accumulated by the GEMM itself (beta = 1), no atomics.  A weight gradient is a SMALL matrix (out x in) summed over very
// MANY points: as one GEMM it is a handful of output tiles with a huge K (12 workgroups for 768 x 256: 29 TFLOP/s).  Above 16 k points the points are cut into 32 slices
// computed as one strided-batched GEMM into partial matrices, which one small kernel then adds to dw.
int run_grad_w_fast(int64_t npts, Seg g, Seg a, Seg b, int out, int in, float *dw, hipStream_t st, int arith)
{
    if (arith != 0 && npts >= 4096 && out < 32) {           // a head of a few rows: a pass over x per four rows (gemm_tn_thin)
        if (a.n > 0) NRF_TRY(gemm_tn_thin(npts, g, a, out, in, 0, dw, st));
        if (b.n > 0) NRF_TRY(gemm_tn_thin(npts, g, b, out, in, a.n, dw, st));
        return NRF_OK;
    }
    if (arith != 0 && npts >= 4096 && out >= 32) {          // the split-precision modes: the hand-written bf16x3 TN product (gemm_bf16x3.hip), one call per column segment
        if (a.n > 0) NRF_TRY(gemm_tn_bf16x3(npts, g, a, out, in, 0, dw, st));
        if (b.n > 0) NRF_TRY(gemm_tn_bf16x3(npts, g, b, out, in, a.n, dw, st));
        return NRF_OK;
    }
    rocblas_handle h = (npts >= 256) ? rb_handle(st) : nullptr;
    if (!h || npts > 0x7fffffff) return run_grad_w(npts, g, a, b, out, in, dw, st);
    const float one = 1.0f, zero = 0.0f;
    constexpr int SLICES = 32;
    const bool split = g_rb.sgemm_sb && npts >= 16384 && (int64_t)out * in <= 1024 * 1024;
    float *part = nullptr;
    if (split) {
        const int nmax = a.n > b.n ? a.n : b.n;
        if (scratch_take(reinterpret_cast<void **>(&part), (size_t)SLICES * out * nmax * sizeof(float), st) != hipSuccess) { set_error("run_grad_w_fast: scratch allocation failed"); return NRF_ERR_HIP; }
    }
    auto seg = [&](Seg x, int col0) {
        if (x.n == 0) return true;
        if (split) {
            const int64_t per = npts / SLICES, done = per * SLICES;
            // part[b] (x.n x out, compact) = X_b (x.n x per) . G_b^T
            if (g_rb.sgemm_sb(h, rocblas_operation_none, rocblas_operation_transpose, x.n, out, (rocblas_int)per, &one, x.p + x.off, x.stride, (rocblas_stride)(per * x.stride), g.p + g.off,
                              g.stride, (rocblas_stride)(per * g.stride), &zero, part, x.n, (rocblas_stride)((int64_t)out * x.n), SLICES) != rocblas_status_success) return false;
            hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ceil_div((int64_t)out * x.n, 256)), dim3(256), 0, st, SLICES, out, x.n, in, col0, (const float *)part, dw);
            if (done == npts) return true;
            return g_rb.sgemm(h, rocblas_operation_none, rocblas_operation_transpose, x.n, out, (rocblas_int)(npts - done), &one, x.p + done * x.stride + x.off, x.stride,
                              g.p + done * g.stride + g.off, g.stride, &one, dw + col0, in) == rocblas_status_success;
        }
        // dW_cm (in x out, ld in; rows col0..) += X_cm (x.n x pts, ld x.stride) . G_cm^T (G_cm = out x pts, ld g.stride)
        return g_rb.sgemm(h, rocblas_operation_none, rocblas_operation_transpose, x.n, out, (rocblas_int)npts, &one, x.p + x.off, x.stride, g.p + g.off, g.stride, &one, dw + col0,
                          in) == rocblas_status_success;
    };
    const bool ok = seg(a, 0) && seg(b, a.n);
    if (part) (void)scratch_give(part, st);
    if (!ok) { set_error("rocblas_sgemm failed (weight gradient %d x %d over %lld points)", out, in, (long long)npts); return NRF_ERR_HIP; }
    return NRF_OK;
}

// The back-propagation through a layer with FEW outputs (the LeRF sigma net's last layer: 33), y[p][n] = sum_{o < K} g[p][o] W[o][n] (. mask): as a matrix-core
// product its K is one ragged tile and a half (0.68 ms per 786 432 points through the generic tile kernel, most of it clamped loads of columns that do not exist);
// as plain fp32 FMAs it is 64 K per thread and 64-point tile, hidden behind the 1 KB row stores it exists to produce.  W [K][N] and the tile's g rows sit in LDS; a
// wave owns rows (one instruction = one whole 1 KB row), a lane four columns; the g value of a (row, o) is an LDS broadcast.  Deterministic, exact fp32.
__global__ void __launch_bounds__(256) k_backprop_thin(int64_t P, int K, int N, const float *__restrict__ g, int ldg, const float *__restrict__ W, float *__restrict__ y, int ldy,
                                                       const float *__restrict__ mask, int ldm)
{
    extern __shared__ __attribute__((aligned(16))) float bt_smem[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    float *sw = bt_smem, *sg = bt_smem + (size_t)K * N;
    const int t = threadIdx.x, c4 = (t & 63) * 4, pg = t >> 6;
    for (int i = t; i < K * N; i += 256) sw[i] = W[i];
    const int64_t tiles = (P + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t p0 = tile * 64;
        __syncthreads();                                                   // (first pass: W is in; later: the previous tile's g rows are read)
        for (int i = t; i < 64 * K; i += 256) {
            const int pr = i / K, k = i - pr * K;
            const int64_t p = p0 + pr;
            sg[i] = p < P ? g[p * ldg + k] : 0.0f;
        }
        __syncthreads();
        for (int nb = 0; nb < N; nb += 256) {
            const int n = nb + c4;
            if (n >= N) continue;
            f2 acc[16][2];
#pragma unroll
            for (int j = 0; j < 16; j++) { acc[j][0] = f2{0.0f, 0.0f}; acc[j][1] = f2{0.0f, 0.0f}; }
            for (int k = 0; k < K; k++) {
                const float4 w = *reinterpret_cast<const float4 *>(sw + (size_t)k * N + n);
                const f2 w0 = {w.x, w.y}, w1 = {w.z, w.w};
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const float gv = sg[(pg + 4 * j) * K + k];
                    const f2 g2 = {gv, gv};
                    acc[j][0] = __builtin_elementwise_fma(g2, w0, acc[j][0]);
                    acc[j][1] = __builtin_elementwise_fma(g2, w1, acc[j][1]);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int64_t p = p0 + pg + 4 * j;
                if (p >= P) continue;
                float4 v = {acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
                if (mask) {
                    const float4 m = *reinterpret_cast<const float4 *>(mask + p * ldm + n);
                    v.x = m.x > 0.0f ? v.x : 0.0f; v.y = m.y > 0.0f ? v.y : 0.0f; v.z = m.z > 0.0f ? v.z : 0.0f; v.w = m.w > 0.0f ? v.w : 0.0f;
                }
                *reinterpret_cast<float4 *>(y + p * ldy + n) = v;
            }
        }
    }
}

static bool rows16(const float *p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; }

// y[pt][k] = sum_o g[pt][o] W[o][k]
bool run_backprop_fuses_mask(const nrf_mlp *m, int64_t npts) { return npts >= 256 && train_gemm_for(m) != 0; }

bool run_backprop_uses_mask_bits(const nrf_mlp *m, int64_t npts, int width) { return npts >= 256 && train_gemm_for(m) != 0 && gemm_nt_bits_ok(npts, width); }

int run_backprop_fast(int64_t npts, Seg g, const nrf_mlp *m, const LinearLayer &L, float *y, int y_stride, hipStream_t st, const float *mask_act, int mask_stride, const float *add,
                      int add_stride, const uint64_t *mask_bits, int mask_bits_ld)
{
    const int arith = npts >= 256 ? train_gemm_for(m) : 0;
    static const bool thin_on = [] { const char *e = getenv("NRF_BACKPROP_THIN"); return !e || atoi(e) != 0; }();          // 0: the tile kernel for every shape (A/B)
    const size_t thin_lds = ((size_t)L.out * L.in + (size_t)64 * L.out) * sizeof(float);
    if (arith && !add && thin_on && L.out <= 48 && (L.in & 3) == 0 && thin_lds <= 64 * 1024 && rows16(y, y_stride) && (!mask_act || rows16(mask_act, mask_stride))) {
        const int64_t tiles = ceil_div(npts, (int64_t)64);
        hipLaunchKernelGGL(k_backprop_thin, dim3((unsigned)(tiles < 768 ? tiles : 768)), dim3(256), thin_lds, st, npts, L.out, L.in, g.p + g.off, g.stride,
                           (const float *)(m->d_params + L.w_off), y, y_stride, mask_act, mask_stride);
        NRF_LAUNCH_CHECK();
        return NRF_OK;
    }
    if (arith)          // G . W with W^T [in][out] as the K-contiguous second operand (L.d_wt, refreshed by every nrf_mlp_set_params)
        return gemm_nt_split(arith, npts, L.in, g, Seg{nullptr, 0, 0, 0}, L.d_wt, L.out, y, y_stride, nullptr, 0, mask_act, mask_stride, st, add, add_stride, nullptr, 0, nullptr,
                             nullptr, 0, mask_bits, mask_bits_ld);
    if (mask_act || add) { set_error("internal: run_backprop_fast: the fused ReLU mask exists in bf16x3 mode only"); return NRF_ERR_INVALID_ARG; }
    rocblas_handle h = (npts >= 256) ? rb_handle(st) : nullptr;
    if (!h || npts > 0x7fffffff) return run_backprop(npts, g, m, L, y, y_stride, st);
    const float one = 1.0f, zero = 0.0f;
    // Y_cm (in x pts, ld y_stride) = W_cm (in x out, ld in) . G_cm (out x pts, ld g.stride)
    if (g_rb.sgemm(h, rocblas_operation_none, rocblas_operation_none, L.in, (rocblas_int)npts, L.out, &one, m->d_params + L.w_off, L.in, g.p + g.off, g.stride, &zero, y, y_stride) !=
        rocblas_status_success) {
        set_error("rocblas_sgemm failed (backprop %d <- %d over %lld points)", L.in, L.out, (long long)npts);
        return NRF_ERR_HIP;
    }
    return NRF_OK;
}

// C[b][r][c] summed over b into C: dst[r * ldc + c] = beta * dst[..] + sum_b part[b][r][c]
__global__ void k_sum_partials_rm(int batches, int rows, int cols, const float *__restrict__ part, float beta, float *__restrict__ dst, int ldc)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cols) return;
    float acc = 0.0f;
    for (int b = 0; b < batches; b++) acc += part[(size_t)b * rows * cols + e];
    const int r = e / cols, c = e - r * cols;
    float *d = dst + (size_t)r * ldc + c;
    *d = (beta == 0.0f ? 0.0f : beta * *d) + acc;
}

// A generic ROW-MAJOR product on the fp32 matrix cores (the Gram form of the LeRF training step, lerf_train.hip):  C [M x N] (ldc) = alpha op(A) op(B) + beta C,
// A: transA ? [K x M] : [M x K] (lda), B: transB ? [N x K] : [K x N] (ldb).  A small C summed over a very long K (a 256 x 256 matrix over 10^5 points) is cut into 32
// K-slices computed as one strided-batched GEMM and added by a small kernel, as run_grad_w_fast does.  Returns NRF_ERR_UNSUPPORTED without rocBLAS (the caller has a
// path without it).
int gemm_rm(hipStream_t st, bool transA, bool transB, int64_t M, int64_t N, int64_t K, float alpha, const float *A, int lda, const float *B, int ldb, float beta, float *C, int ldc)
{
    rocblas_handle h = rb_handle(st);
    if (!h || M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return NRF_ERR_UNSUPPORTED;
    if (M == 0 || N == 0) return NRF_OK;
    // row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T, and a row-major array IS its transpose in column-major terms
    const rocblas_operation o1 = transB ? rocblas_operation_transpose : rocblas_operation_none, o2 = transA ? rocblas_operation_transpose : rocblas_operation_none;
    constexpr int SLICES = 32;
    if (g_rb.sgemm_sb && K >= 16384 && M * N <= 1024 * 1024 && transA && !transB && (K % SLICES) == 0) {
        // A: [K x M], B: [K x N]: slice b takes rows [b K / 32, (b + 1) K / 32) of both
        float *part = nullptr;
        if (scratch_take(reinterpret_cast<void **>(&part), (size_t)SLICES * M * N * sizeof(float), st) != hipSuccess) { set_error("gemm_rm: scratch allocation failed"); return NRF_ERR_HIP; }
        const float zero = 0.0f;
        const int64_t per = K / SLICES;
        const bool ok = g_rb.sgemm_sb(h, o1, o2, (rocblas_int)N, (rocblas_int)M, (rocblas_int)per, &alpha, B, ldb, (rocblas_stride)(per * ldb), A, lda, (rocblas_stride)(per * lda), &zero, part,
                                      (rocblas_int)N, (rocblas_stride)(M * N), SLICES) == rocblas_status_success;
        if (ok) hipLaunchKernelGGL(k_sum_partials_rm, dim3((unsigned)ceil_div(M * N, 256)), dim3(256), 0, st, SLICES, (int)M, (int)N, (const float *)part, beta, C, ldc);
        (void)scratch_give(part, st);
        if (!ok) { set_error("rocblas_sgemm_strided_batched failed (%lld x %lld over %lld)", (long long)M, (long long)N, (long long)K); return NRF_ERR_HIP; }
        return NRF_OK;
    }
    if (g_rb.sgemm(h, o1, o2, (rocblas_int)N, (rocblas_int)M, (rocblas_int)K, &alpha, B, ldb, A, lda, &beta, C, ldc) != rocblas_status_success) {
        set_error("rocblas_sgemm failed (%lld x %lld x %lld)", (long long)M, (long long)N, (long long)K);
        return NRF_ERR_HIP;
    }
    return NRF_OK;
}

}  // namespace nrf

// 1 when the training paths' fp32 layer products run as rocBLAS GEMMs on the fp32 matrix cores, 0 when they run mlp.hip's hand-written FMA kernels (rocBLAS absent or NRF_FP32_GEMM=0)
extern "C" NRF_API int nrf_fp32_gemm_available(void) { return nrf::fp32_gemm_available(); }
