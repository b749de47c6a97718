// hash_fast.hip -- the renderer's fast hash-grid encode for the CuHashEmbedder semantics (CuHashEmbedder.cu:8-102),
// F = 2, fp16 table.  Same arithmetic, bit for bit, as k_hash_cu in encode.hip; what changes is WHERE things live:
//
//   * output is LEVEL-MAJOR fp16: feats[level][point] as one half2 (4 B) per (point, level).  A wavefront's store is one
//     contiguous 256-B segment (the reference kernel's out_feat[p][level*2+k] is a 4-B store at a 64-B stride), and the
//     fused MLP kernel reads its layer-0 MFMA operand straight from this layout (4 coalesced dword loads per lane).
//   * levels are pinned to XCDs.  Workgroups are dealt round-robin over the 8 XCDs, so block b runs on the XCD
//     labelled b % 8; giving all blocks with the same label the same ceil(L/8) levels keeps those levels' slice of the
//     table (2 x 2 MiB, overlapping to 3 MiB through the reference's level-offset quirk) resident in that XCD's
//     4 MiB L2 instead of all 32 MiB competing for every L2.  Placement is a speed assumption only.
//   * points are formed on the fly as o + d*z from the packed ray batch (NeRFRenderer.h:419): no pts buffer.
#include "encode.h"
#include "hash_fast.h"

namespace nrf {

struct PointPrep {
    float q[3];     // (clamp(x) - min) / (max - min), level independent (CuHashEmbedder.cu:44-46 before * mul)
    bool keep;
};

__device__ __forceinline__ PointPrep prep_point(const HashParams &hp, const F3 &pt)
{
    PointPrep r;
    const float x[3] = {pt.x, pt.y, pt.z};
    r.keep = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]);
        r.keep = r.keep && (x[a] == c);
        r.q[a] = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]);
    }
    return r;
}

// One (point, level): 8 half2 gathers issued back to back, then the fp32 blend in the reference's order.
__device__ __forceinline__ __half2 encode_level(const HashParams &hp, const PointPrep &pp, int l)
{
    float fr[3];
    uint32_t pos[3];
    const float mul = hp.level_scale[l];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float q = pp.q[a] * mul;
        q = q + hp.bias[l * 3 + a];
        const float fl = floorf(q);
        pos[a] = (uint32_t)fl;
        fr[a] = q - fl;
    }
    const uint32_t pa = hp.primes[l * 3 + 0], pb = hp.primes[l * 3 + 1], pc = hp.primes[l * 3 + 2];
    const uint32_t lsz = hp.local_size[l];
    const __half *fp = reinterpret_cast<const __half *>(hp.table) + hp.local_idx[l];
    float acc[2];
    cu_blend<2>(fp, pos, fr, pa, pb, pc, lsz, acc);
    return __halves2half2(__float2half_rn(acc[0]), __float2half_rn(acc[1]));
}

template <int PPT>
__global__ void __launch_bounds__(256)
k_hash_cu_lm(HashParams hp, PointSource ps, int64_t p, __half2 *__restrict__ feats, int64_t pstride, uint8_t *__restrict__ keep, int lpg, int xcd_map,
             int level0)
{
    int level;
    int64_t tile;
    if (xcd_map) {
        const int g = blockIdx.x & 7;
        const int64_t j = blockIdx.x >> 3;
        const int sub = (int)(j % lpg);
        tile = j / lpg;
        // xcd_map 1: XCD g owns levels [g*lpg, (g+1)*lpg);  2: mirrored pairing (cheap coarse level with an expensive fine one)
        if (xcd_map == 1) level = g * lpg + sub;
        else { const int k = sub * 8 + g; level = (sub & 1) ? (hp.n_levels - 1 - (k - 8 * sub) - 8 * (sub >> 1)) : (k - 8 * sub) + 8 * (sub >> 1); }
        if (level >= hp.n_levels || level < 0) return;
    } else {
        level = level0 + blockIdx.y;
        tile = blockIdx.x;
    }
    PointPrep pp[PPT];
    int64_t idx[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) {
        idx[q] = (tile * PPT + q) * 256 + threadIdx.x;
        const int64_t i = idx[q] < p ? idx[q] : p - 1;
        pp[q] = prep_point(hp, load_point(ps, i));
    }
    __half2 out[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) out[q] = encode_level(hp, pp[q], level);
#pragma unroll
    for (int q = 0; q < PPT; q++) {
        if (idx[q] < p) {
            feats[(int64_t)level * pstride + idx[q]] = out[q];
            if (level == 0 && keep) keep[idx[q]] = pp[q].keep ? 1 : 0;
        }
    }
}

// per-ray direction features as fp16 rows [n, V] (the MLP kernel's colour-net operand): SH of the ray's view direction
__global__ void k_dirs_f16(int64_t n, int degree, int variant, const float *__restrict__ rays, int stride, __half *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *dp = rays + i * stride + 8;
    float r[64];
    if (variant == NRF_SH_CUDA) sh_cuda(dp[0], dp[1], dp[2], degree, r);
    else sh_libtorch(dp[0], dp[1], dp[2], degree, r);
    const int od = degree * degree;
    for (int k = 0; k < od; k++) out[i * od + k] = __float2half_rn(r[k]);
}

int hash_fast_supported(const nrf_hash *h)
{
    return h && h->desc.mode == NRF_HASH_CU && h->desc.n_features == 2 && h->table_set && h->primes_set;
}

int launch_hash_lm(const nrf_hash *h, const PointSource &ps, int64_t p, __half2 *feats, int64_t pstride, uint8_t *keep, int variant, hipStream_t st,
                   int level_lo, int level_hi)
{
    if (p == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_HASH, st);
    const int L = h->desc.n_levels;
    const int lpg = (L + 7) / 8;
    const int ppt = (variant & 1) ? 2 : 1;
    const int xcd = (variant >> 1) & 3;
    if (level_hi < 0) level_hi = L;
    const int64_t ntiles = ceil_div(p, 256 * ppt);
    dim3 grid = xcd ? dim3((unsigned)(ntiles * lpg * 8)) : dim3((unsigned)ntiles, (unsigned)(level_hi - level_lo));
    if (ppt == 1) hipLaunchKernelGGL(k_hash_cu_lm<1>, grid, dim3(256), 0, st, h->params, ps, p, feats, pstride, keep, lpg, xcd, level_lo);
    else hipLaunchKernelGGL(k_hash_cu_lm<2>, grid, dim3(256), 0, st, h->params, ps, p, feats, pstride, keep, lpg, xcd, level_lo);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_dirs_f16(const float *rays, int stride, int64_t n, int degree, int variant, __half *out, hipStream_t st)
{
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_dirs_f16, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, degree, variant, rays, stride, out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

// Debug / tuning entry (not part of the public header): level-major encode of explicit points with a kernel variant.
extern "C" NRF_API int nrf_dbg_hash_lm(const nrf_hash *h, const float *d_x, int64_t p, int variant, int level_lo, int level_hi, void *d_feats, uint8_t *d_keep, void *stream)
{
    NRF_CHECK_ARG(h && d_x && d_feats && p >= 0, "nrf_dbg_hash_lm: bad argument");
    NRF_CHECK_ARG(hash_fast_supported(h), "nrf_dbg_hash_lm: needs a CuHashEmbedder-mode grid with F = 2, table and primes set");
    PointSource ps{d_x, nullptr, nullptr, 0, 1};
    return launch_hash_lm(h, ps, p, reinterpret_cast<__half2 *>(d_feats), p, d_keep, variant, as_stream(stream), level_lo, level_hi);
}
