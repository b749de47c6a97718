// encode.h -- shared encoder state and device helpers.
#pragma once
#include "common.h"

#include <cmath>

#define NRF_MAX_LEVELS 32

namespace nrf {

// Passed by value to the kernels (kernarg segment -> scalar loads; everything is wave-uniform per level).
struct HashParams {
    Bbox bbox;
    int n_levels;
    int log2_t;
    const void *table;                      // NGP: fp32 [L][2^T][F];  CU: fp16 [L*2^T, F]
    float level_scale[NRF_MAX_LEVELS];      // NGP: floor()ed resolution;  CU: un-floored scale mul_l
    uint32_t primes[NRF_MAX_LEVELS * 3];    // CU
    float bias[NRF_MAX_LEVELS * 3];         // CU: the level's offset;  NGP: the level's cell size (max - min) / res per axis
    int32_t local_idx[NRF_MAX_LEVELS];      // CU: level base offset in ELEMENTS (the reference's overlap quirk)
    uint32_t local_size[NRF_MAX_LEVELS];    // CU
    // baked dense image of the coarse levels (hash_fast.hip): entry offset into `dense` (-1: level stays hashed) and block grid
    const void *dense;
    int64_t dense_off[NRF_MAX_LEVELS];
    int32_t dense_nby[NRF_MAX_LEVELS], dense_nbz[NRF_MAX_LEVELS];   // (x,y) tiles per row; z extent
};

}  // namespace nrf

struct nrf_hash {
    nrf_hash_desc desc;
    nrf::HashParams params;
    void *d_table = nullptr;
    bool table_set = false;
    bool primes_set = false;
    // fast-path image of the table (hash_fast.hip), rebuilt lazily after the table or primes change
    void *d_fast = nullptr;
    size_t fast_bytes = 0;
    bool fast_valid = false;
    int dense_levels = 0;
    // RMS of the table's entries (device float; a deterministic two-stage sum: the same table always gives the same bits), re-derived on demand
    // (hash_table_rms_update) when the table was uploaded since, and a counter of the uploads: what a renderer's split-precision network scales its first layer by
    // (mlp.h, k_small_scales)
    float *d_table_rms = nullptr;
    float *d_rms_part = nullptr;
    uint64_t table_version = 0, rms_version = ~(uint64_t)0;
    size_t dense_budget = (size_t)24 << 30;   // bytes of dense image to bake (levels 0.. while they fit): 4.4 GB at 16..512; at 16..1024 the finest level (35 GB) stays hashed -- HBM is 288 GB
};

namespace nrf {

int launch_pe(const float *x, int x_stride, int64_t rows, int nfreq, int rep, float *out, int out_stride, hipStream_t st);
int launch_sh(const float *dirs, int dir_stride, int64_t rows, int degree, int variant, int rep, float *out, int out_stride, hipStream_t st);
int launch_hash(const nrf_hash *h, const PointSource &ps, int64_t p, float *out, int out_stride, uint8_t *keep, hipStream_t st);
int hash_table_rms_update(nrf_hash *h, hipStream_t st);          // *h->d_table_rms brought up to date with the table (no-op when it is)

// CuHashEmbedder.cu:70-100: 8 hashed corners, trilinear weights as three-factor products, fp32 sum of
// products in the order 000,001,010,011,100,101,110,111 (bit 2 = x, bit 1 = y, bit 0 = z).
// GATHER: 0 = plain global loads; 1..3 = buffer loads through a descriptor of the table (voffset = byte offset of the row) with
// cache policy aux 0 / nt / sc1 (tuning experiments, hash_fast.hip)
template <int F, int GATHER = 0>
__device__ __forceinline__ void cu_blend(const __half *fp, const uint32_t pos[3], const float fr[3], uint32_t pa, uint32_t pb, uint32_t pc,
                                         uint32_t lsz, float acc[F], __amdgpu_buffer_rsrc_t rsrc = __amdgpu_buffer_rsrc_t(), uint32_t base_bytes = 0)
{
    const float a = fr[0], b = fr[1], c = fr[2];
    const float oma = 1.0f - a, omb = 1.0f - b, omc = 1.0f - c;
    const uint32_t hx0 = pos[0] * pa, hx1 = (pos[0] + 1u) * pa;
    const uint32_t hy0 = pos[1] * pb, hy1 = (pos[1] + 1u) * pb;
    const uint32_t hz0 = pos[2] * pc, hz1 = (pos[2] + 1u) * pc;
    uint32_t ps[8];
    float ws[8];
    // `% local_size` (CuHashEmbedder.cu:70-77): local_size = (2^T >> 4) << 4 is a power of two for every T >= 4, where the
    // modulo is a mask; the general remainder (a ~25-instruction sequence, 8 times per point and level) is kept for other sizes.
    const bool pow2 = (lsz & (lsz - 1u)) == 0u;          // wave-uniform (lsz comes from the kernel arguments)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t hx = (k & 4) ? hx1 : hx0, hy = (k & 2) ? hy1 : hy0, hz = (k & 1) ? hz1 : hz0;
        const uint32_t hv = hx ^ hy ^ hz;
        ps[k] = pow2 ? (hv & (lsz - 1u)) : (hv % lsz);
        const float wx = (k & 4) ? a : oma, wy = (k & 2) ? b : omb, wz = (k & 1) ? c : omc;
        ws[k] = wx * wy * wz;
    }
    if constexpr (F == 2) {
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if constexpr (GATHER == 0) v[k] = __half22float2(*reinterpret_cast<const __half2 *>(fp + (size_t)ps[k] * 2));
            else {
                constexpr int aux = GATHER == 1 ? 0 : GATHER == 2 ? 2 : 16;
                const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, ps[k] * 4u, base_bytes, aux);
                __half2 hv; __builtin_memcpy(&hv, &w, 4);
                v[k] = __half22float2(hv);
            }
        }
        float a0 = ws[0] * v[0].x, a1 = ws[0] * v[0].y;
#pragma unroll
        for (int k = 1; k < 8; k++) { a0 = a0 + ws[k] * v[k].x; a1 = a1 + ws[k] * v[k].y; }
        acc[0] = a0; acc[1] = a1;
    } else {
#pragma unroll
        for (int f = 0; f < F; f++) {
            float s = ws[0] * __half2float(fp[(size_t)ps[0] * F + f]);
#pragma unroll
            for (int k = 1; k < 8; k++) s = s + ws[k] * __half2float(fp[(size_t)ps[k] * F + f]);
            acc[f] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// S2 LibTorch SH (degree <= 5).  Expression order follows NeRF.cpp:158-196 (tensor-scalar ops, left to right).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_libtorch(float x, float y, float z, int degree, float *r)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    r[0] = C0;
    if (degree <= 1) return;
    r[1] = -C1 * y; r[2] = C1 * z; r[3] = -C1 * x;
    if (degree <= 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    r[4] = 1.0925484305920792f * xy;
    r[5] = -1.0925484305920792f * yz;
    r[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    r[7] = -1.0925484305920792f * xz;
    r[8] = 0.5462742152960396f * (xx - yy);
    if (degree <= 3) return;
    r[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
    r[10] = 2.890611442640554f * xy * z;
    r[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
    r[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    r[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
    r[14] = 1.445305721320277f * z * (xx - yy);
    r[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    if (degree <= 4) return;
    r[16] = 2.5033429417967046f * xy * (xx - yy);
    r[17] = -1.7701307697799304f * yz * (3.0f * xx - yy);
    r[18] = 0.9461746957575601f * xy * (7.0f * zz - 1.0f);
    r[19] = -0.6690465435572892f * yz * (7.0f * zz - 3.0f);
    r[20] = 0.10578554691520431f * (zz * (35.0f * zz - 30.0f) + 3.0f);
    r[21] = -0.6690465435572892f * xz * (7.0f * zz - 3.0f);
    r[22] = 0.47308734787878004f * (xx - yy) * (7.0f * zz - 1.0f);
    r[23] = -1.7701307697799304f * xz * (xx - 3.0f * yy);
    r[24] = 0.6258357354491761f * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

// S1  CUDA SH basis for unit directions, degree <= 8 (CuSHEncoder.cu:15-104).  The expressions keep the
// kernel's evaluation order; r must hold degree^2 floats.
__device__ __forceinline__ void sh_cuda(float x, float y, float z, int degree, float *r)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
    const float x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
    r[0] = 0.28209479177387814f;
    if (degree <= 1) return;
    r[1] = -0.48860251190291987f * y;
    r[2] = 0.48860251190291987f * z;
    r[3] = -0.48860251190291987f * x;
    if (degree <= 2) return;
    r[4] = 1.0925484305920792f * xy;
    r[5] = -1.0925484305920792f * yz;
    r[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    r[7] = -1.0925484305920792f * xz;
    r[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if (degree <= 3) return;
    r[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    r[10] = 2.8906114426405538f * xy * z;
    r[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    r[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    r[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    r[14] = 1.4453057213202769f * z * (x2 - y2);
    r[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    if (degree <= 4) return;
    r[16] = 2.5033429417967046f * xy * (x2 - y2);
    r[17] = 1.7701307697799304f * yz * (-3.0f * x2 + y2);
    r[18] = 0.94617469575756008f * xy * (7.0f * z2 - 1.0f);
    r[19] = 0.66904654355728921f * yz * (3.0f - 7.0f * z2);
    r[20] = -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f;
    r[21] = 0.66904654355728921f * xz * (3.0f - 7.0f * z2);
    r[22] = 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f);
    r[23] = 1.7701307697799304f * xz * (-x2 + 3.0f * y2);
    r[24] = -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4;
    if (degree <= 5) return;
    r[25] = 0.65638205684017015f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4);
    r[26] = 8.3026492595241645f * xy * z * (x2 - y2);
    r[27] = -0.48923829943525038f * y * (3.0f * x2 - y2) * (9.0f * z2 - 1.0f);
    r[28] = 4.7935367849733241f * xy * z * (3.0f * z2 - 1.0f);
    r[29] = 0.45294665119569694f * y * (14.0f * z2 - 21.0f * z4 - 1.0f);
    r[30] = 0.1169503224534236f * z * (-70.0f * z2 + 63.0f * z4 + 15.0f);
    r[31] = 0.45294665119569694f * x * (14.0f * z2 - 21.0f * z4 - 1.0f);
    r[32] = 2.3967683924866621f * z * (x2 - y2) * (3.0f * z2 - 1.0f);
    r[33] = -0.48923829943525038f * x * (x2 - 3.0f * y2) * (9.0f * z2 - 1.0f);
    r[34] = 2.0756623148810411f * z * (-6.0f * x2 * y2 + x4 + y4);
    r[35] = 0.65638205684017015f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4);
    if (degree <= 6) return;
    r[36] = 1.3663682103838286f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
    r[37] = 2.3666191622317521f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4);
    r[38] = 2.0182596029148963f * xy * (x2 - y2) * (11.0f * z2 - 1.0f);
    r[39] = -0.92120525951492349f * yz * (3.0f * x2 - y2) * (11.0f * z2 - 3.0f);
    r[40] = 0.92120525951492349f * xy * (-18.0f * z2 + 33.0f * z4 + 1.0f);
    r[41] = 0.58262136251873131f * yz * (30.0f * z2 - 33.0f * z4 - 5.0f);
    r[42] = 6.6747662381009842f * z2 - 20.024298714302954f * z4 + 14.684485723822165f * z6 - 0.31784601133814211f;
    r[43] = 0.58262136251873131f * xz * (30.0f * z2 - 33.0f * z4 - 5.0f);
    r[44] = 0.46060262975746175f * (x2 - y2) * (11.0f * z2 * (3.0f * z2 - 1.0f) - 7.0f * z2 + 1.0f);
    r[45] = -0.92120525951492349f * xz * (x2 - 3.0f * y2) * (11.0f * z2 - 3.0f);
    r[46] = 0.50456490072872406f * (11.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4);
    r[47] = 2.3666191622317521f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4);
    r[48] = 10.247761577878714f * x2 * y4 - 10.247761577878714f * x4 * y2 + 0.6831841051919143f * x6 - 0.6831841051919143f * y6;
    if (degree <= 7) return;
    r[49] = 0.70716273252459627f * y * (-21.0f * x2 * y4 + 35.0f * x4 * y2 - 7.0f * x6 + y6);
    r[50] = 5.2919213236038001f * xy * z * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
    r[51] = -0.51891557872026028f * y * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + 5.0f * x4 + y4);
    r[52] = 4.1513246297620823f * xy * z * (x2 - y2) * (13.0f * z2 - 3.0f);
    r[53] = -0.15645893386229404f * y * (3.0f * x2 - y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f);
    r[54] = 0.44253269244498261f * xy * z * (-110.0f * z2 + 143.0f * z4 + 15.0f);
    r[55] = 0.090331607582517306f * y * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f);
    r[56] = 0.068284276912004949f * z * (315.0f * z2 - 693.0f * z4 + 429.0f * z6 - 35.0f);
    r[57] = 0.090331607582517306f * x * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f);
    r[58] = 0.07375544874083044f * z * (x2 - y2) * (143.0f * z2 * (3.0f * z2 - 1.0f) - 187.0f * z2 + 45.0f);
    r[59] = -0.15645893386229404f * x * (x2 - 3.0f * y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f);
    r[60] = 1.0378311574405206f * z * (13.0f * z2 - 3.0f) * (-6.0f * x2 * y2 + x4 + y4);
    r[61] = -0.51891557872026028f * x * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + x4 + 5.0f * y4);
    r[62] = 2.6459606618019f * z * (15.0f * x2 * y4 - 15.0f * x4 * y2 + x6 - y6);
    r[63] = 0.70716273252459627f * x * (-35.0f * x2 * y4 + 21.0f * x4 * y2 - x6 + 7.0f * y6);
}

}  // namespace nrf
