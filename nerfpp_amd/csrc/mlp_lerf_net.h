// mlp_lerf_net.h -- layer plan, fragment bookkeeping and kernel arguments of the LeRF head on the matrix cores, shared by mlp_lerf_mfma.hip (fp16 operands)
// and mlp_lerf_split_mfma.hip (hi + lo fp16 operand pairs).  See mlp_lerf_mfma.hip for the formulation.
#pragma once
#include "mlp.h"

namespace nrf {
namespace lerf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 8;              // waves per workgroup
constexpr int NBLK = 32 * NW;      // points per workgroup iteration (one 32-point tile per wave)
constexpr int MAXF = 32;           // fragments (1 KB each) in the largest chunk
constexpr int IN = 128, HID = 256, GEO = 32, EMB = 768;

__host__ __device__ inline int perm_row(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// Layers: 0 sigma0 [nat 8] -> 8 tiles ReLU | 1 sigma1 [chained 16] -> 2 tiles (row 0 = sigma, rows 1..32 = geo) | 2 LE0 [chained 4 | nat 8]
// -> 8 tiles ReLU (cat[geo, in], LeRF.cpp) | 3 GRAM [chained 16] -> 8 tiles: t = (W^T W) a, ||LE1(a)||^2 = a . t | 4 LE1 [chained 16] -> 24 tiles
// (weighted-sum pass).  NL = 2: the sigma net alone (kernel A); NL = 5: everything (kernel B).  Both walk the same weight image.
//
// The norm: LE1 is bias-free (LeRF.cpp:21-24), so ||W a||^2 = a^T (W^T W) a.  The 256 x 256 Gram matrix is formed once per weight set (in double, at pack
// time) and costs 8 tiles x 16 k-steps per point tile instead of the 24 x 16 of a first full pass through the 256 -> 768 layer: 704 matrix instructions
// per 32 points instead of 960.
template <int NL>
struct Net {
    static constexpr int NLAYER = NL;
    static constexpr int tiles(int l) { return l == 0 ? 8 : l == 1 ? 2 : l == 2 ? 8 : l == 3 ? 8 : 24; }
    static constexpr int ks_nat(int l) { return (l == 0 || l == 2) ? 8 : 0; }
    static constexpr int ks_ch(int l) { return l == 0 ? 0 : l == 2 ? 4 : 16; }
    static constexpr bool nat_first(int l) { return l != 2; }
    static constexpr int ks(int l) { return ks_nat(l) + ks_ch(l); }
    static constexpr int chunk_tiles(int l, int) { return l == 1 ? 1 : 2; }            // sigma1: 1 + 1 keeps the chunk count even
    static constexpr int chunks(int l) { return l == 1 ? 2 : tiles(l) / 2; }
    static constexpr int first_chunk(int l) { int n = 0; for (int i = 0; i < l; i++) n += chunks(i); return n; }
    static constexpr int total_chunks() { return first_chunk(NLAYER); }
    static constexpr int layer_of(int ci) { int l = 0; while (first_chunk(l + 1) <= ci) l++; return l; }
    static constexpr int chunk_frags(int ci) { const int l = layer_of(ci); return chunk_tiles(l, ci - first_chunk(l)) * ks(l); }
    static constexpr int chunk_off(int ci) { int n = 0; for (int i = 0; i < ci; i++) n += chunk_frags(i); return n; }
};
static_assert(Net<2>::total_chunks() == 6 && Net<4>::total_chunks() == 14 && Net<5>::total_chunks() == 26, "chunk counts");
constexpr int IMAGE_FRAGS = 8 * 8 + 2 * 16 + 8 * 12 + 8 * 16 + 24 * 16;       // 704 KB: sigma0, sigma1, LE0, Gram (chained operands) + LE1 in NATURAL operand order (kernel C)
constexpr int LE1_FRAG0 = 8 * 8 + 2 * 16 + 8 * 12 + 8 * 16;
static_assert(Net<5>::chunk_off(Net<5>::first_chunk(4)) + 24 * 16 == IMAGE_FRAGS, "image size");

struct Args {
    const float *x; int x_stride;           // hash features [p, 128] fp32 ...
    const __half *x_lm; int64_t pstride;    // ... or level-major fp16 [16][pstride][8] (nrf_hash_encode_lm_f16): k-step s, lane half h = level 2s + h, one 16-byte load
    const float *weights;                   // [p] render weights (kernel B)
    const uint8_t *keep;                    // optional: sigma forced to 0 where false (kernel A)
    float *sigma;                           // [p] (kernel A)
    float *out;                             // [n, 768] accumulated (kernel B)
    int s;                                  // samples per ray, a multiple of 32
    float gram_scale;                       // the image's Gram matrix is (W^T W) / gram_scale (a power of two keeping it inside fp16 range); set by the launchers
    const int32_t *src;                     // optional (x_lm): point q reads feature column src[q] -- the merge map of a feature-reusing fine pass (nrf_fine_depths_merge)
    // Split precision, level-major input: the sigma net's second output (sigma, geo32 = LE0's chained operand: three (hi, lo) fragment pairs per point) handed
    // from kernel A to kernel B through memory, so that B starts at LE0 instead of re-evaluating the sigma net (224 of its 832 matrix instructions per tile).
    // Layout: plane (f, part), f = 0..2, part = hi / lo: half8 [geo_stride columns][2 lane halves]; kernel A writes column q, kernel B reads column src[q] (or q).
    void *geo;
    int64_t geo_stride;
};
constexpr int GEO_FRAGS = 3;                                        // fragments 0, 1: sigma + geo[0..30]; fragment 2: geo[31] (the fourth is all zero)
constexpr int64_t GEO_BYTES_PER_COLUMN = GEO_FRAGS * 2 * 2 * 16;    // 192

}  // namespace lerf

// split-precision passes (mlp_lerf_split_mfma.hip); same arguments as the fp16 launchers
int lerf_split_available(const nrf_mlp *m);
int lerf_split_sigma(const nrf_mlp *m, const lerf::Args &a, int64_t p, hipStream_t st);
int lerf_split_embedding_passes(const nrf_mlp *m, lerf::Args a, int64_t n, int s, float *d_out, hipStream_t st);

}  // namespace nrf
