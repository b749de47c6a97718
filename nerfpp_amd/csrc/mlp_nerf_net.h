// mlp_nerf_net.h -- layer plan and fragment bookkeeping of the classic 8 x 256 NeRF on the matrix cores, shared by mlp_nerf_mfma.hip
// (NRF_PREC_F16_MFMA) and mlp_nerf_split_mfma.hip (NRF_PREC_F16_SPLIT).  See mlp_nerf_mfma.hip for the formulation.
#pragma once
#include "mlp.h"

namespace nrf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Measured on MI355X (bench --workload classic): 8 waves x 1 point tile = 1021 TFLOP/s (40.9 % of the 2.5 PF dense fp16 peak);
// 4 waves x 2 tiles (one wave per SIMD, half the LDS fragment reads) = 779 TFLOP/s: with a single wave per SIMD nothing
// covers the bias/ReLU/convert epilogue between tiles, with two the other wave's MFMAs do.
#ifndef NRF_NERF_NW
#define NRF_NERF_NW 8
#define NRF_NERF_NPT 1
#endif
constexpr int NW = NRF_NERF_NW;        // waves per workgroup
constexpr int NPT = NRF_NERF_NPT;      // 32-point tiles per wave (every weight fragment read from LDS feeds NPT MFMAs)
constexpr int NBLK = 32 * NPT * NW;    // points per workgroup iteration
constexpr int MAXF = 40;               // fragments (1 KB each) in the largest chunk
constexpr int NBIAS = 8 * 256 + 160 + 32;

__host__ __device__ inline int nerf_perm_row(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

struct NerfNet {
    static constexpr int NLAYER = 10;
    static constexpr int tiles(int l) { return l < 8 ? 8 : l == 8 ? 5 : 1; }
    static constexpr int ks_nat(int l) { return (l == 0 || l == 5) ? 4 : l == 8 ? 2 : 0; }
    static constexpr int ks_ch(int l) { return l == 0 ? 0 : l == 9 ? 8 : 16; }
    static constexpr bool nat_first(int l) { return l != 8; }
    static constexpr int ks(int l) { return ks_nat(l) + ks_ch(l); }
    static constexpr int chunks(int l) { return (tiles(l) + 1) / 2; }
    static constexpr int chunk_tiles(int l, int c) { return (2 * c + 2 <= tiles(l)) ? 2 : 1; }
    static constexpr int first_chunk(int l) { int n = 0; for (int i = 0; i < l; i++) n += chunks(i); return n; }
    static constexpr int total_chunks() { return first_chunk(NLAYER); }
    static constexpr int layer_of(int ci) { int l = 0; while (first_chunk(l + 1) <= ci) l++; return l; }
    static constexpr int chunk_frags(int ci) { const int l = layer_of(ci); return chunk_tiles(l, ci - first_chunk(l)) * ks(l); }
    static constexpr int chunk_off(int ci) { int n = 0; for (int i = 0; i < ci; i++) n += chunk_frags(i); return n; }
    static constexpr int total_frags() { return chunk_off(total_chunks()); }
    static constexpr int bias_off(int l) { int n = 0; for (int i = 0; i < l; i++) n += tiles(i) * 32; return n; }
};
static_assert(NerfNet::total_chunks() == 36, "chunk count");

// Input: fp32 rows [p, 90] = [PE(10)(x) | PE(4)(dir)] (the generic BaseNeRF::forward boundary), or -- FUSED, the renderer's
// fast path -- the packed rays, the depth table and per-RAY fp16 direction encodings: the kernel forms x = o + d*z
// (NeRFRenderer.h:419) and its 63 sinusoidal features (NeRF.cpp:33-37, same nrf_sincosf as the stand-alone encoder, so the
// operand fragments are bit-identical to the unfused path) in registers; no [P, 90] input is ever written.
struct NerfInput {
    const float *x; int x_stride;
    const float *rays; int ray_stride; const float *z; int s; const __half *dirs;    // dirs: [n, 32] fp16, PE(4) of the view direction, zero padded
    const __half *dirs_lo;                                                            // split precision: the rounding residuals of `dirs`, same layout
};

// split-precision image and launcher (mlp_nerf_split_mfma.hip)
int mlp_nerf_forward_split(const nrf_mlp *m, const NerfInput &in, bool fused, int64_t p, float *out, int os, hipStream_t st);

}  // namespace nrf
