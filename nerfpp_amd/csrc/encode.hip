// encode.hip -- position / direction encoders behind the BaseEmbedder plugin surface (BaseEmbedder.h:6-15):
//   sinusoidal PE            EmbedderImpl          NeRF.cpp:4-39
//   spherical harmonics      SHEncoderImpl         NeRF.cpp:131-201      (variant NRF_SH_LIBTORCH)
//                            CuSHKernel            CuSHEncoder.cu:4-107  (variant NRF_SH_CUDA)
//   multiresolution hash     HashEmbedderImpl      NeRF.cpp:208-318      (mode NRF_HASH_NGP)
//                            CuHashEmbedder kernel CuHashEmbedder.cu:8-102 + CuHashEmbedder.cpp:85-103 (mode NRF_HASH_CU)
//
// Generic row-major kernels (any caller, fp32 [p, D] outputs with a row stride so the renderer can write the
// concatenated MLP input in place).  The level-major fp16 fast path used by the fused renderer lives in
// hash_fast.hip.  Built with -ffp-contract=off; sin/cos from include/nrf_math.h: every value equals the oracle's bit for bit.
#include "encode.h"
#include "hash_fast.h"

namespace nrf {

// ------------------------------------------------------------------------------------------------
// E1 sinusoidal PE.  One thread per (row, frequency slot); slot 0 copies x.
// out row = [x, sin(x f0), cos(x f0), sin(x f1), ...]; f_i = 2^i exactly (powf(2, (n-1)/(n-1)*i), NeRF.cpp:15).
// `rep`: row i encodes x[i / rep] (view directions are shared by the `rep` samples of a ray, NeRFRenderer.h:179).
// ------------------------------------------------------------------------------------------------
__global__ void k_pe(int64_t rows, int nfreq, int rep, const float *__restrict__ x, int x_stride, float *__restrict__ out, int out_stride)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int slots = nfreq + 1;
    if (gid >= rows * slots) return;
    const int64_t row = gid / slots;
    const int slot = (int)(gid - row * slots);
    const float *xp = x + (row / rep) * x_stride;
    float *o = out + row * out_stride;
    if (slot == 0) {
        o[0] = xp[0]; o[1] = xp[1]; o[2] = xp[2];
        return;
    }
    const int f = slot - 1;
    const float freq = __builtin_ldexpf(1.0f, f);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = xp[a] * freq;
        float sn, cs;
        nrf_sincosf(v, &sn, &cs);
        o[3 + f * 6 + a] = sn;
        o[3 + f * 6 + 3 + a] = cs;
    }
}

__global__ void k_sh(int64_t rows, int degree, int variant, int rep, const float *__restrict__ dirs, int dir_stride,
                     float *__restrict__ out, int out_stride)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float *dp = dirs + (i / rep) * dir_stride;
    float r[64];
    if (variant == NRF_SH_LIBTORCH) sh_libtorch(dp[0], dp[1], dp[2], degree, r);
    else sh_cuda(dp[0], dp[1], dp[2], degree, r);
    float *o = out + i * out_stride;
    const int od = degree * degree;
    for (int k = 0; k < od; k++) o[k] = r[k];
}

// ------------------------------------------------------------------------------------------------
// H1  LibTorch hash grid, generic row-major kernel.  One thread per (point, level).
// ------------------------------------------------------------------------------------------------
template <int F>
__global__ void k_hash_ngp(HashParams hp, PointSource ps, int64_t p, float *__restrict__ out, int out_stride, uint8_t *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= p) return;
    const F3 pt = load_point(ps, i);
    const float x[3] = {pt.x, pt.y, pt.z};
    float w[3];
    int32_t idx[3];
    bool kp = true;
    const float res = hp.level_scale[l];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]);
        kp = kp && (x[a] == c);
        const float grid = (hp.bbox.mx[a] - hp.bbox.mn[a]) / res;
        const float fl = floorf((c - hp.bbox.mn[a]) / grid);
        idx[a] = (int32_t)fl;
        const float vmin = fl * grid + hp.bbox.mn[a];       // (float)int64 idx * grid + min
        const float vmax = vmin + grid;
        w[a] = (x[a] - vmin) / (vmax - vmin);               // UNCLAMPED x (NeRF.cpp:311)
    }
    if (l == 0 && keep) keep[i] = kp ? 1 : 0;
    const float *tl = reinterpret_cast<const float *>(hp.table) + (int64_t)l * ((int64_t)1 << hp.log2_t) * F;
    const uint32_t hmask = (1u << hp.log2_t) - 1u;
    // low T bits of the int64 hash (NeRF.cpp:230-237) == the same expression in uint32 arithmetic
    const float *e[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint32_t cx = (uint32_t)idx[0] + ((c >> 2) & 1), cy = (uint32_t)idx[1] + ((c >> 1) & 1), cz = (uint32_t)idx[2] + (c & 1);
        const uint32_t h = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & hmask;
        e[c] = tl + (int64_t)h * F;
    }
    const float omx = 1.0f - w[0], omy = 1.0f - w[1], omz = 1.0f - w[2];
    float *o = out + i * out_stride + l * F;
#pragma unroll
    for (int f = 0; f < F; f++) {
        const float c00 = e[0][f] * omx + e[4][f] * w[0];
        const float c01 = e[1][f] * omx + e[5][f] * w[0];
        const float c10 = e[2][f] * omx + e[6][f] * w[0];
        const float c11 = e[3][f] * omx + e[7][f] * w[0];
        const float c0 = c00 * omy + c10 * w[1];
        const float c1 = c01 * omy + c11 * w[1];
        o[f] = c0 * omz + c1 * w[2];
    }
}

// ------------------------------------------------------------------------------------------------
// H2  CUDA-semantics hash grid, generic row-major kernel.  One thread per (point, level).
// ------------------------------------------------------------------------------------------------
template <int F>
__global__ void k_hash_cu(HashParams hp, PointSource ps, int64_t p, float *__restrict__ out, int out_stride, uint8_t *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= p) return;
    const F3 pt = load_point(ps, i);
    const float x[3] = {pt.x, pt.y, pt.z};
    bool kp = true;
    float fr[3];
    uint32_t pos[3];
    const float mul = hp.level_scale[l];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]);      // CuHashEmbedder.cpp:92-94
        kp = kp && (x[a] == c);
        float q = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]) * mul;  // .cu:44-46
        q = q + hp.bias[l * 3 + a];                                              // .cu:61-63
        const float fl = floorf(q);
        pos[a] = (uint32_t)fl;                                                   // .cu:66-68
        fr[a] = q - fl;                                                          // .cu:79-81
    }
    if (l == 0 && keep) keep[i] = kp ? 1 : 0;
    const uint32_t pa = hp.primes[l * 3 + 0], pb = hp.primes[l * 3 + 1], pc = hp.primes[l * 3 + 2];
    const uint32_t lsz = hp.local_size[l];
    const __half *fp = reinterpret_cast<const __half *>(hp.table) + hp.local_idx[l];   // the overlap quirk (.cu:54)
    float acc[F];
    cu_blend<F>(fp, pos, fr, pa, pb, pc, lsz, acc);
    float *o = out + i * out_stride + l * F;
#pragma unroll
    for (int f = 0; f < F; f++) o[f] = __half2float(__float2half_rn(acc[f]));     // one fp16 rounding (.cu:95), returned as fp32 (.cu:274)
}

// The same lookup, written level-major in fp16: feats[level][point][F] (2F bytes per thread, contiguous across the points of a wave).  For F = 8 that is one
// 16-byte store per thread and exactly the operand fragment of a matrix-core consumer: k-step s, lane half h = level 2s + h (mlp_lerf_mfma.hip).
// The values are the row-major kernel's (one fp16 rounding of the fp32 blend, .cu:95), so consumers see identical inputs.
// LPT levels per thread (blockIdx.y indexes groups of LPT levels): the point, its clamp and its three box-coordinate divisions are level-independent and formed once
template <int F, int LPT>
__global__ void k_hash_cu_lmf(HashParams hp, PointSource ps, int64_t p, __half *__restrict__ feats, int64_t pstride, uint8_t *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l0 = blockIdx.y * LPT;
    if (i >= p) return;
    const F3 pt = load_point(ps, i);
    const float x[3] = {pt.x, pt.y, pt.z};
    bool kp = true;
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]);
        kp = kp && (x[a] == c);
        t[a] = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]);
    }
    if (l0 == 0 && keep) keep[i] = kp ? 1 : 0;
#pragma unroll
    for (int j = 0; j < LPT; j++) {
        const int l = l0 + j;
        float fr[3];
        uint32_t pos[3];
        const float mul = hp.level_scale[l];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float q = t[a] * mul;
            q = q + hp.bias[l * 3 + a];
            const float fl = floorf(q);
            pos[a] = (uint32_t)fl;
            fr[a] = q - fl;
        }
        const uint32_t pa = hp.primes[l * 3 + 0], pb = hp.primes[l * 3 + 1], pc = hp.primes[l * 3 + 2];
        const __half *fp = reinterpret_cast<const __half *>(hp.table) + hp.local_idx[l];
        float acc[F];
        cu_blend<F>(fp, pos, fr, pa, pb, pc, hp.local_size[l], acc);
        __half o[F];
#pragma unroll
        for (int f = 0; f < F; f++) o[f] = __float2half_rn(acc[f]);
        __half *dst = feats + ((int64_t)l * pstride + i) * F;
        if constexpr (F == 8) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(o);
        else if constexpr (F == 4) *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(o);
        else if constexpr (F == 2) *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(o);
        else dst[0] = o[0];
    }
}

// RMS of `n` floats, deterministic: block b of RMS_BLOCKS sums elements b * 256 + t, + RMS_BLOCKS * 256, ... per thread, the block's 256 partials and then the blocks'
// are added in index order
constexpr int RMS_BLOCKS = 256;
template <typename T>
__global__ void __launch_bounds__(256) k_sumsq_partial(int64_t n, const T *__restrict__ in, float *__restrict__ part)
{
    __shared__ float s[256];
    float a = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)RMS_BLOCKS * 256) { const float v = (float)in[i]; a += v * v; }
    s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) { float r = 0.0f; for (int i = 0; i < 256; i++) r += s[i]; part[blockIdx.x] = r; }
}
__global__ void k_rms_final(int64_t n, const float *__restrict__ part, float *__restrict__ rms)
{
    float r = 0.0f;
    for (int i = 0; i < RMS_BLOCKS; i++) r += part[i];
    *rms = sqrtf(r / (float)n);
}

// *d_table_rms = RMS of the table as the kernels see it (the fp16 table of the CuHashEmbedder mode, the fp32 one of the LibTorch twin), in `st`'s order; a no-op while the
// table has not changed since the last call
int hash_table_rms_update(nrf_hash *h, hipStream_t st)
{
    if (!h || !h->table_set || !h->d_table_rms) return NRF_OK;
    if (h->rms_version == h->table_version) return NRF_OK;
    const int64_t elems = nrf_hash_table_elems(h);
    if (h->desc.mode == NRF_HASH_NGP) hipLaunchKernelGGL(k_sumsq_partial<float>, dim3(RMS_BLOCKS), dim3(256), 0, st, elems, reinterpret_cast<const float *>(h->d_table), h->d_rms_part);
    else hipLaunchKernelGGL(k_sumsq_partial<__half>, dim3(RMS_BLOCKS), dim3(256), 0, st, elems, reinterpret_cast<const __half *>(h->d_table), h->d_rms_part);
    NRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rms_final, dim3(1), dim3(1), 0, st, elems, h->d_rms_part, h->d_table_rms);
    NRF_LAUNCH_CHECK();
    h->rms_version = h->table_version;
    return NRF_OK;
}

__global__ void k_f32_to_f16(int64_t n, const float *__restrict__ in, __half *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __float2half_rn(in[i]);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int launch_pe(const float *x, int x_stride, int64_t rows, int nfreq, int rep, float *out, int out_stride, hipStream_t st)
{
    if (rows == 0) return NRF_OK;
    const int64_t total = rows * (nfreq + 1);
    hipLaunchKernelGGL(k_pe, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, rows, nfreq, rep, x, x_stride, out, out_stride);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_sh(const float *dirs, int dir_stride, int64_t rows, int degree, int variant, int rep, float *out, int out_stride, hipStream_t st)
{
    if (rows == 0) return NRF_OK;
    hipLaunchKernelGGL(k_sh, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, st, rows, degree, variant, rep, dirs, dir_stride, out, out_stride);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

template <int F>
static int launch_hash_f(const nrf_hash *h, const PointSource &ps, int64_t p, float *out, int out_stride, uint8_t *keep, hipStream_t st)
{
    const dim3 grid((unsigned)ceil_div(p, 256), (unsigned)h->desc.n_levels);
    if (h->desc.mode == NRF_HASH_NGP) hipLaunchKernelGGL(k_hash_ngp<F>, grid, dim3(256), 0, st, h->params, ps, p, out, out_stride, keep);
    else hipLaunchKernelGGL(k_hash_cu<F>, grid, dim3(256), 0, st, h->params, ps, p, out, out_stride, keep);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_hash(const nrf_hash *h, const PointSource &ps, int64_t p, float *out, int out_stride, uint8_t *keep, hipStream_t st)
{
    if (p == 0) return NRF_OK;
    if (!h->table_set) { set_error("hash grid: table not uploaded (nrf_hash_set_table)"); return NRF_ERR_INVALID_ARG; }
    if (h->desc.mode == NRF_HASH_CU && !h->primes_set) { set_error("hash grid (CU mode): primes not set (nrf_hash_set_primes)"); return NRF_ERR_INVALID_ARG; }
    ProfScope prof(NRF_PROF_HASH, st);
    switch (h->desc.n_features) {
        case 1: return launch_hash_f<1>(h, ps, p, out, out_stride, keep, st);
        case 2: return launch_hash_f<2>(h, ps, p, out, out_stride, keep, st);
        case 4: return launch_hash_f<4>(h, ps, p, out, out_stride, keep, st);
        case 8: return launch_hash_f<8>(h, ps, p, out, out_stride, keep, st);
        default: set_error("hash grid: n_features %d not built (1, 2, 4, 8)", h->desc.n_features); return NRF_ERR_UNSUPPORTED;
    }
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_pe_encode(const float *d_x, int64_t p, int nfreq, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_x && d_out && p >= 0 && nfreq >= 1 && nfreq <= 32, "nrf_pe_encode: bad argument");
    return launch_pe(d_x, 3, p, nfreq, 1, d_out, 3 + 6 * nfreq, as_stream(stream));
}

int nrf_sh_encode(const float *d_dirs, int64_t p, int degree, int variant, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_dirs && d_out && p >= 0, "nrf_sh_encode: bad argument");
    NRF_CHECK_ARG(variant == NRF_SH_LIBTORCH || variant == NRF_SH_CUDA, "nrf_sh_encode: unknown variant %d", variant);
    NRF_CHECK_ARG(degree >= 1 && degree <= (variant == NRF_SH_LIBTORCH ? 5 : 8), "nrf_sh_encode: degree %d out of range for variant %d", degree, variant);
    return launch_sh(d_dirs, 3, p, degree, variant, 1, d_out, degree * degree, as_stream(stream));
}

int nrf_hash_create(const nrf_hash_desc *desc, nrf_hash **out)
{
    NRF_CHECK_ARG(desc && out, "nrf_hash_create: null pointer");
    NRF_CHECK_ARG(desc->mode == NRF_HASH_NGP || desc->mode == NRF_HASH_CU, "nrf_hash_create: unknown mode %d", desc->mode);
    NRF_CHECK_ARG(desc->n_levels >= 2 && desc->n_levels <= NRF_MAX_LEVELS, "nrf_hash_create: n_levels %d outside [2,%d]", desc->n_levels, NRF_MAX_LEVELS);
    NRF_CHECK_ARG(desc->log2_hashmap_size >= 4 && desc->log2_hashmap_size <= 24, "nrf_hash_create: log2_hashmap_size %d outside [4,24]", desc->log2_hashmap_size);
    NRF_CHECK_ARG(desc->n_features == 1 || desc->n_features == 2 || desc->n_features == 4 || desc->n_features == 8, "nrf_hash_create: n_features %d not in {1,2,4,8}", desc->n_features);
    NRF_CHECK_ARG(desc->base_resolution >= 1 && desc->finest_resolution >= desc->base_resolution, "nrf_hash_create: bad resolutions");
    for (int a = 0; a < 3; a++) NRF_CHECK_ARG(desc->bbox[3 + a] > desc->bbox[a], "nrf_hash_create: empty bounding box on axis %d", a);
    nrf_hash *h = new nrf_hash();
    h->desc = *desc;
    const int L = desc->n_levels;
    HashParams &hp = h->params;
    memset(&hp, 0, sizeof(hp));
    for (int a = 0; a < 3; a++) { hp.bbox.mn[a] = desc->bbox[a]; hp.bbox.mx[a] = desc->bbox[3 + a]; }
    hp.n_levels = L; hp.log2_t = desc->log2_hashmap_size;
    for (int l = 0; l < NRF_MAX_LEVELS; l++) hp.dense_off[l] = -1;
    if (desc->mode == NRF_HASH_NGP) {
        // NeRF.cpp:251: b = float(exp((ln finest - ln base)/(L-1)));  :309: res_l = floor(float(base * pow(b, l)))
        const float b = (float)exp((log((double)desc->finest_resolution) - log((double)desc->base_resolution)) / (double)(L - 1));
        for (int l = 0; l < L; l++) hp.level_scale[l] = floorf((float)((double)desc->base_resolution * pow((double)b, (double)l)));
        // NeRF.cpp:262: grid_size = (max - min) / resolution, one correctly rounded fp32 division per level and axis: formed here once, read by the
        // level-major encode as a scalar (hash_fast.hip; the generic kernels divide on the device, same IEEE result)
        for (int l = 0; l < L; l++)
            for (int a = 0; a < 3; a++) hp.bias[l * 3 + a] = (hp.bbox.mx[a] - hp.bbox.mn[a]) / hp.level_scale[l];
    } else {
        // CuHashEmbedder.cu:40: mul_l = exp2f((log2f(finest) - log2f(base)) * l / (L-1) + log2f(base)), fp32 libm on the host
        for (int l = 0; l < L; l++)
            hp.level_scale[l] = exp2f((log2f((float)desc->finest_resolution) - log2f((float)desc->base_resolution)) * (float)l / (float)(L - 1) + log2f((float)desc->base_resolution));
        // CuHashEmbedder.cpp:62-68: local_size = (2^T >> 4) << 4; local_idx = cumsum - local_size
        const int32_t ls = (int32_t)((((int64_t)1 << desc->log2_hashmap_size) >> 4) << 4);
        for (int l = 0; l < L; l++) { hp.local_size[l] = (uint32_t)ls; hp.local_idx[l] = (int32_t)((int64_t)l * ls); }
    }
    const int64_t elems = (int64_t)L * ((int64_t)1 << desc->log2_hashmap_size) * desc->n_features;
    const size_t bytes = (size_t)elems * (desc->mode == NRF_HASH_NGP ? 4 : 2);
    hipError_t e = hipMalloc(&h->d_table, bytes);
    if (e != hipSuccess) { set_error("nrf_hash_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); delete h; return NRF_ERR_HIP; }
    hp.table = h->d_table;
    *out = h;
    return NRF_OK;
}

void nrf_hash_destroy(nrf_hash *h)
{
    if (!h) return;
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_fast) (void)hipFree(h->d_fast);
    if (h->d_table_rms) (void)hipFree(h->d_table_rms);
    if (h->d_rms_part) (void)hipFree(h->d_rms_part);
    delete h;
}

int nrf_hash_memory_bytes(const nrf_hash *h, int64_t *table_bytes, int64_t *baked_bytes)
{
    NRF_CHECK_ARG(h, "nrf_hash_memory_bytes: null pointer");
    if (table_bytes) *table_bytes = nrf_hash_table_elems(h) * (h->desc.mode == NRF_HASH_CU ? 2 : 4);
    if (baked_bytes) *baked_bytes = (int64_t)h->fast_bytes;
    return NRF_OK;
}

int nrf_hash_output_dims(const nrf_hash *h) { return h ? h->desc.n_levels * h->desc.n_features : 0; }
int64_t nrf_hash_table_elems(const nrf_hash *h) { return h ? (int64_t)h->desc.n_levels * ((int64_t)1 << h->desc.log2_hashmap_size) * h->desc.n_features : 0; }

int nrf_hash_set_table(nrf_hash *h, const float *src, int src_on_device, void *stream)
{
    NRF_CHECK_ARG(h && src, "nrf_hash_set_table: null pointer");
    const int64_t elems = nrf_hash_table_elems(h);
    hipStream_t st = as_stream(stream);
    if (!h->d_table_rms) {
        NRF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_table_rms), sizeof(float)));
        NRF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_rms_part), RMS_BLOCKS * sizeof(float)));
    }
    if (h->desc.mode == NRF_HASH_NGP) {
        NRF_HIP(hipMemcpyAsync(h->d_table, src, (size_t)elems * 4, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    } else {
        // fp32 master -> fp16 once (the reference re-casts on every forward, CuHashEmbedder.cu:257)
        const float *d_src = src;
        float *tmp = nullptr;
        if (!src_on_device) {
            NRF_HIP(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)elems * 4));
            NRF_HIP(hipMemcpyAsync(tmp, src, (size_t)elems * 4, hipMemcpyHostToDevice, st));
            d_src = tmp;
        }
        hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)ceil_div(elems, 256)), dim3(256), 0, st, elems, d_src, reinterpret_cast<__half *>(h->d_table));
        NRF_LAUNCH_CHECK();
        if (tmp) { NRF_HIP(hipStreamSynchronize(st)); NRF_HIP(hipFree(tmp)); }
    }
    h->table_set = true;
    h->fast_valid = false;
    h->table_version++;          // (the table's RMS is re-derived by hash_table_rms_update when a consumer asks: a LeRF language grid of 134 MB never does)
    if (hash_fast_supported(h)) NRF_TRY(hash_fast_prepare(h, h->dense_budget, st));
    return NRF_OK;
}


int nrf_hash_set_primes(nrf_hash *h, const int32_t *primes, const float *biases)
{
    NRF_CHECK_ARG(h && primes, "nrf_hash_set_primes: null pointer");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_CU, "nrf_hash_set_primes: only the CuHashEmbedder mode has per-level primes");
    // The reference draws every multiplier as a prime in [2^28, 2^30) (CuHashEmbedder.cpp:28-49).  A zero removes its axis from the hash (all-zero: every
    // corner of every voxel lands on row 0 of its level -- what an uninitialised `_primes` buffer would silently do); an even one loses the axis' low bit.
    for (int i = 0; i < h->desc.n_levels * 3; i++)
        NRF_CHECK_ARG(primes[i] != 0 && (primes[i] & 1), "nrf_hash_set_primes: multiplier %d of level %d is %d: zero / even multipliers degenerate the hash "
                      "(the reference draws primes in [2^28, 2^30), CuHashEmbedder.cpp:28-49) -- was the `_primes` buffer initialised or loaded?", i % 3, i / 3, primes[i]);
    for (int i = 0; i < h->desc.n_levels * 3; i++) {
        h->params.primes[i] = (uint32_t)primes[i];
        h->params.bias[i] = biases ? biases[i] : 0.0f;
    }
    h->primes_set = true;
    h->fast_valid = false;
    // model-load time: the table upload may still be in flight on a caller's (non-blocking) stream, and the bake below runs on the default one
    NRF_HIP(hipDeviceSynchronize());
    if (hash_fast_supported(h)) NRF_TRY(hash_fast_prepare(h, h->dense_budget, nullptr));
    return NRF_OK;
}

int nrf_hash_set_dense_budget(nrf_hash *h, int64_t budget_bytes, void *stream)
{
    NRF_CHECK_ARG(h && budget_bytes >= 0, "nrf_hash_set_dense_budget: bad argument");
    h->dense_budget = (size_t)budget_bytes;
    h->fast_valid = false;
    if (!hash_fast_supported(h)) return NRF_OK;           // takes effect at the next table / primes upload
    return hash_fast_prepare(h, h->dense_budget, as_stream(stream));
}

int64_t nrf_hash_get_dense_budget(const nrf_hash *h) { return h ? (int64_t)h->dense_budget : 0; }

int nrf_hash_get_level_scales(const nrf_hash *h, float *scales_out)
{
    NRF_CHECK_ARG(h && scales_out, "nrf_hash_get_level_scales: null pointer");
    for (int l = 0; l < h->desc.n_levels; l++) scales_out[l] = h->params.level_scale[l];
    return NRF_OK;
}

int nrf_hash_set_level_scales(nrf_hash *h, const float *scales, void *stream)
{
    NRF_CHECK_ARG(h && scales, "nrf_hash_set_level_scales: null pointer");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_CU, "nrf_hash_set_level_scales: the CuHashEmbedder mode computes mul_l with exp2f / log2f on the device (CuHashEmbedder.cu:40); the HashEmbedder's resolutions are floor()ed integers");
    for (int l = 0; l < h->desc.n_levels; l++) NRF_CHECK_ARG(scales[l] >= 1.0f && scales[l] < 65536.0f, "nrf_hash_set_level_scales: scale %d = %g outside [1, 65536)", l, (double)scales[l]);
    for (int l = 0; l < h->desc.n_levels; l++) h->params.level_scale[l] = scales[l];
    h->fast_valid = false;                                   // the dense image's extents follow floor(mul_l)
    if (!hash_fast_supported(h)) return NRF_OK;
    NRF_HIP(hipDeviceSynchronize());
    return hash_fast_prepare(h, h->dense_budget, as_stream(stream));
}

int nrf_hash_encode_lm_f16(const nrf_hash *h, const float *d_x, int64_t p, void *d_feats, uint8_t *d_keep_mask, void *stream)
{
    return nrf_hash_encode_lm_f16_strided(h, d_x, p, d_feats, p, d_keep_mask, stream);
}

int nrf_hash_encode_lm_f16_strided(const nrf_hash *h, const float *d_x, int64_t p, void *d_feats, int64_t pstride, uint8_t *d_keep_mask, void *stream)
{
    NRF_CHECK_ARG(h && d_x && d_feats && p >= 0 && pstride >= p, "nrf_hash_encode_lm_f16: bad argument");
    if (h->desc.mode != NRF_HASH_CU) { set_error("nrf_hash_encode_lm_f16: built for the CuHashEmbedder (its features ARE fp16, CuHashEmbedder.cu:95); the HashEmbedder's are fp32"); return NRF_ERR_UNSUPPORTED; }
    if (!h->table_set || !h->primes_set) { set_error("nrf_hash_encode_lm_f16: table / primes not set"); return NRF_ERR_INVALID_ARG; }
    NRF_CHECK_ARG((reinterpret_cast<uintptr_t>(d_feats) & (size_t)(h->desc.n_features * 2 - 1)) == 0, "nrf_hash_encode_lm_f16: feature buffer must be aligned to one point's %d bytes", h->desc.n_features * 2);
    if (p == 0) return NRF_OK;
    hipStream_t st = as_stream(stream);
    PointSource ps{d_x, nullptr, nullptr, 0, 1};
    // F = 2: the renderer's own encode kernel (four coarse levels per thread, the dense image where one is baked, the hashed table otherwise): same values
    if (h->desc.n_features == 2 && h->desc.n_levels >= 8 && hash_fast_supported(h) && (reinterpret_cast<uintptr_t>(d_feats) & 3) == 0)
        return launch_hash_lm(h, ps, p, reinterpret_cast<__half2 *>(d_feats), pstride, d_keep_mask, HASH_LM_DEFAULT_VARIANT, st);
    ProfScope prof(NRF_PROF_HASH, st);
    // levels per thread: ONE.  This kernel's lookups are eight 16-byte gathers per level out of the hashed table (F = 8: 128 B per point and level); it lives on the number
    // of gathers in flight, and sharing the point preparation between levels costs more than it saves -- LeRF frame, same call, ms of hash encode: 1 per thread 40.4-41.1,
    // 2: 43.8-43.9, 4: 50.2-50.3, 8: 59.5-59.6 (docs/history/profiles/round3/r6h_lerf_hash_levels_per_thread_ab.log).  (The F = 2 fast path is the opposite case: hash_fast.hip.)
#ifndef NRF_HASH_LMF_LPT
#define NRF_HASH_LMF_LPT 1
#endif
    const int lpt = (h->desc.n_levels % NRF_HASH_LMF_LPT) == 0 ? NRF_HASH_LMF_LPT : 1;
    const dim3 grid((unsigned)ceil_div(p, 256), (unsigned)(h->desc.n_levels / lpt));
    __half *f = reinterpret_cast<__half *>(d_feats);
#define NRF_LMF(F_) do { if (lpt == 1) hipLaunchKernelGGL((k_hash_cu_lmf<F_, 1>), grid, dim3(256), 0, st, h->params, ps, p, f, pstride, d_keep_mask); \
                         else hipLaunchKernelGGL((k_hash_cu_lmf<F_, NRF_HASH_LMF_LPT>), grid, dim3(256), 0, st, h->params, ps, p, f, pstride, d_keep_mask); } while (0)
    switch (h->desc.n_features) {
        case 1: NRF_LMF(1); break;
        case 2: NRF_LMF(2); break;
        case 4: NRF_LMF(4); break;
        case 8: NRF_LMF(8); break;
        default: set_error("nrf_hash_encode_lm_f16: n_features %d not built (1, 2, 4, 8)", h->desc.n_features); return NRF_ERR_UNSUPPORTED;
    }
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_hash_encode(const nrf_hash *h, const float *d_x, int64_t p, float *d_out, uint8_t *d_keep_mask, void *stream)
{
    NRF_CHECK_ARG(h && d_x && d_out && p >= 0, "nrf_hash_encode: bad argument");
    PointSource ps{d_x, nullptr, nullptr, 0, 1};
    return launch_hash(h, ps, p, d_out, nrf_hash_output_dims(h), d_keep_mask, as_stream(stream));
}

}  // extern "C"
