// gemm_bf16x3.hip -- the layer products of the TRAINING paths (classic NeRF backward, LeRF head backward) on the bf16 / fp16 matrix cores in split precision (round 6;
// replaces the rocBLAS sgemm calls of gemm_f32.hip): the recomputed forward  Y = cat[a, b] W^T (+ bias)(ReLU), the back-propagation  G_in = G W (. mask)  -- both "NT"
// products whose operands are K-contiguous (k_gemm_nt_rows, k_gemm_nt) -- and the weight gradient  dW += G^T X  (k_gemm_tn: the contraction runs over the points).
//
// Arithmetic.  Every fp32 operand x is carried as x = hi + lo and a product is three matrix-core instructions into one fp32 accumulator: ah.bh + al.bh + ah.bl.
//   bf16x3  hi = bf16(x), lo = bf16(x - hi): 16 significant bits with fp32's own EXPONENT range -- nothing can leave it.  One product within 6e-6 of its largest entry.
//           Used for dW (a leaf of the step: nothing is computed from it, and a per-point scale could not be undone in a sum over points).
//   f16x3   hi + lo fp16 (22 significant bits) of POWER-OF-TWO SCALED operands: every row of A by its own largest entry, B by its largest entry, undone exactly in the
//           epilogue (round 5's unscaled fp16x3 products were "not range-safe" -- gradients of 1e-9 lie below fp16's normals -- and were removed).  One product within
//           5e-7 of its largest entry (rocBLAS sgemm: 8e-7).  The default of the forward / back-propagation products for the classic and the LeRF networks.
// The fp32 matrix instruction (v_mfma_f32_32x32x2_f32, what rocBLAS runs) retires 1/16 of the 16-bit rate: three products cost 3/16 of it.
//
// NT kernels: C [M x N] = A [M x K] . B [N x K]^T, row-major fp32 in memory, M = points (10^5..10^6), N, K <= a few hundred.  A may be the concatenation of TWO column
// segments (the skip concat cat[input_pts, h] of NeRFImpl, cat[geo, x] of the LeRF head): the K loop walks segment 0 then segment 1, B's columns follow.
//   * B (the weights: the same for all ~6 000 workgroups of a layer product) is split ONCE per product into the kernels' LDS image (k_gb_split_b; F16: after
//     k_gb_absmax has found its largest entry) and copied tile by tile, 16 bytes per thread, no arithmetic.
//   * k_gemm_nt_rows (K in {128, 160, 256}, 16-byte aligned rows, N > 128): 128 x 256 output tile per 512-thread workgroup; the workgroup's WHOLE A block (128 rows x K:
//     contiguous 128 KB) is requested in one burst -- with K tiles of 32 requested one by one every row is visited eight times, 128 bytes at a time, microseconds apart,
//     DRAM pages re-opened for each piece: ~2 TB/s, for rocBLAS alike -- and the C block leaves through LDS as whole rows for the same reason.  The row's largest entry
//     (F16) is 8 TK maxima and three lane exchanges away.
//   * k_gemm_nt (everything else: ragged K, unaligned segments, narrow N): K tiles of 32 through two LDS stages, two register sets in flight; F16 first scans the
//     workgroup's rows for their largest entries (their second read comes out of the caches); the 256-wide tile shares the row-staged epilogue.
//   * LDS image per operand and half: [k-step of 16][row][16 elements] -- a fragment read (lane = row r, half h -> 16 bytes at (ks, r, 8h)) of a 32-row tile covers 1 KB
//     contiguously: conflict-free ds_read_b128.
//   * epilogue: inverse scales, + bias[n], ReLU, and the ReLU mask of the NEXT backward stage (C = act > 0 ? C : 0), which removes the k_bias_relu / k_relu_mask passes.
//   * workgroups of one XCD work through a contiguous eighth of the tiles (they share that XCD's L2).
#include "mlp.h"

#include <atomic>

namespace nrf {

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 gb_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gb_f16x4 __attribute__((ext_vector_type(4)));
typedef float gb_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t gb_u32x4 __attribute__((ext_vector_type(4)));

// The two split arithmetics of these kernels (same LDS image, same instruction count):
//   F16 = false  hi + lo bf16 (16 significant bits, fp32's exponent range: nothing to scale)                                                -- mode 1, "bf16x3"
//   F16 = true   hi + lo fp16 (22 significant bits) of POWER-OF-TWO SCALED operands: every row of A by its own largest entry (brought to [2^13, 2^14); the row's
//                whole K is in registers before the first split), B by its largest entry (a small pre-pass, k_gb_absmax); the epilogue multiplies the two
//                inverse powers back (exact).  An entry far below its row's maximum loses RELATIVE precision (fp16 subnormals: absolute step 2^-24 of the scaled
//                row) but never more than 2^-38 of the row's maximum in absolute terms -- which is what a dot product feels.                  -- mode 2, "f16x3"
template <bool F16> struct GbT;
template <> struct GbT<false> {
    typedef __bf16 e; typedef gb_bf16x8 v8; typedef gb_bf16x4 v4;
    static __device__ __forceinline__ gb_f32x16 mfma(v8 a, v8 b, gb_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct GbT<true> {
    typedef _Float16 e; typedef gb_f16x8 v8; typedef gb_f16x4 v4;
    static __device__ __forceinline__ gb_f32x16 mfma(v8 a, v8 b, gb_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// power-of-two scale that brings a largest magnitude mx into [2^13, 2^14), and its inverse.  mx = 0 / subnormal / below 2^-114: scale 2^127, inverse flushed to 0 (the
// products are 0 or below fp32's normal range anyway); inf / NaN rows keep their inf / NaN through the fp16 conversion, as an fp32 product would.
__device__ __forceinline__ void gb_pow2_scale(float mx, float &scale, float &inv)
{
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    e = e < 13 ? 13 : e;
    scale = __uint_as_float((uint32_t)(267 - e) << 23);
    inv = __uint_as_float((uint32_t)(e - 13) << 23);
}
__device__ __forceinline__ float gb_absmax4(const float4 &v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

constexpr int GB_BM = 128, GB_BK = 32;

// the bit pattern of the largest |B[n][k]| (k < K) into *out (zeroed before the launch) with one atomic per workgroup: |x|'s bits order like the values
__global__ void __launch_bounds__(256) k_gb_absmax(int N, int K, const float *__restrict__ b, int ldb, uint32_t *__restrict__ out)
{
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float mx = 0.0f;
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        const float *row = b + (size_t)n * ldb;
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(row[k]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// WNW waves along n (64 columns each) x 2 waves along m: BN = 64 WNW, 128 WNW threads.  LDS per stage: A hi | A lo (8 KB each) | B hi | B lo (BN x 64 bytes each)
template <int WNW> struct GbCfg {
    static constexpr int BN = 64 * WNW, THREADS = 128 * WNW;
    static constexpr int A_HALF = 2 * GB_BM * 32, B_HALF = 2 * BN * 32;
    static constexpr int STAGE = 2 * A_HALF + 2 * B_HALF;
    static constexpr int QA = GB_BM * 8 / THREADS, QB = BN * 8 / THREADS;          // quads (four consecutive k of one row) per thread and K tile
    static constexpr int RSTEP = THREADS / 8;                                       // rows between a thread's quads
};

struct GemmNT {
    const float *a0; int lda0, k0;        // A columns [0, k0)
    const float *a1; int lda1, k1;        // A columns [k0, k0 + k1) (k1 == 0: none)
    float *c; int ldc;
    int64_t M; int N;
    const float *bias; int relu;
    const float *mask; int mask_ld;       // optional [M][mask_ld]: C = mask > 0 ? C : 0
    const float *add; int add_ld;         // optional [M][add_ld]: C = (C after bias / ReLU / mask) + add  (a gradient that reaches the same tensor along a second path)
    const float *r1s; int r1s_ld;         // optional rank-1 term BEFORE the ReLU / mask: C += r1s[m * r1s_ld] * r1w[n]  (the back-propagation through a one-row layer
    const float *r1w;                     //  that reads the same input: g_alpha (x) w_alpha of NeRFImpl's alpha_linear)
    // The ReLU mask as BITS (256-wide tiles, N <= 256): a forward product with ReLU leaves, per row, four 64-bit words -- bit `lane` of word j = (C[m][4 lane + j] > 0), the
    // write-out's own thread layout -- and the back-propagation product that needs that mask reads those 32 bytes instead of the activation's 1 KB row (a third of its
    // memory traffic).  Strides in 64-bit words.  bits_in replaces `mask` where the launch can use it (gemm_nt_bits_ok).
    uint64_t *bits_out; int bits_out_ld;
    const uint64_t *bits_in; int bits_in_ld;
    int va0, va1;                         // widest aligned vector load of each A segment: 4, 2 or 1 floats
    const uint32_t *bmax;                 // F16 arithmetic: bits of the largest |B| entry (k_gb_absmax)
    const unsigned char *bimg;            // B already split (k_gb_split_b): [hi | lo][K tile][k-step][npad rows][16 elements], i.e. the kernels' LDS image tile by tile
    int64_t bimg_half;                    // bytes of one half of it
    int npad;                             // rows of the image (N rounded up to the workgroup's tile width)
};

// B -> its split image, ONCE per product instead of once per workgroup and K tile (the weights are the same for all 6 000 workgroups of a layer product: re-splitting
// them was a quarter of the kernels' vector instructions).  Tile t < t0 covers columns [32 t, 32 t + 32) of segment 0 (zero beyond k0), tile t >= t0 columns
// k0 + 32 (t - t0) ... of segment 1 -- the K loop's own tiling; rows >= N are zero.  One thread per (tile, k-step, row): 16 elements, two 32-byte stores.
template <bool F16>
__global__ void __launch_bounds__(256) k_gb_split_b(int N, int npad, int k0, int k1, const float *__restrict__ b, int ldb, const uint32_t *__restrict__ bmax, unsigned char *__restrict__ img,
                                                    int64_t img_half)
{
    const int t0 = (k0 + GB_BK - 1) / GB_BK, t1 = (k1 + GB_BK - 1) / GB_BK;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)(t0 + t1) * 2 * npad) return;
    const int n = (int)(idx % npad);
    const int ks = (int)((idx / npad) & 1);
    const int tile = (int)(idx / npad / 2);
    const bool s1 = tile >= t0;
    const int kb = (s1 ? tile - t0 : tile) * GB_BK + ks * 16, kend = s1 ? k1 : k0, col0 = s1 ? k0 : 0;
    float scale = 1.0f, inv;
    if (F16) gb_pow2_scale(__uint_as_float(bmax[0]), scale, inv);
    typename GbT<F16>::e hi[16], lo[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        float x = (n < N && kb + j < kend) ? b[(size_t)n * ldb + col0 + kb + j] : 0.0f;
        if (F16) x *= scale;
        const typename GbT<F16>::e t = (typename GbT<F16>::e)x;
        hi[j] = t; lo[j] = (typename GbT<F16>::e)(x - (float)t);
    }
    typename GbT<F16>::e *dh = reinterpret_cast<typename GbT<F16>::e *>(img) + idx * 16, *dl = reinterpret_cast<typename GbT<F16>::e *>(img + img_half) + idx * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) { dh[j] = hi[j]; dl[j] = lo[j]; }
}

// NQ rows' quads (four consecutive k each) of one operand for one K tile.  Rows past the end are CLAMPED to the last row (valid memory; their products land in output
// rows / columns that the epilogue does not store), so an interior tile -- every tile but the last one of a segment -- is NQ unconditional vector loads in one basic
// block (the compiler keeps them in flight together); only a segment's last tile, where a quad may straddle `kend`, takes the guarded scalar path.
template <int NQ>
__device__ __forceinline__ void gb_load_rows(const float *base, int ld, const int64_t (&row)[NQ], int kb, int kend, bool tile_full, int vec, float4 (&dst)[NQ])
{
    if (tile_full) {
        if (vec == 4) {
#pragma unroll
            for (int i = 0; i < NQ; i++) dst[i] = *reinterpret_cast<const float4 *>(base + row[i] * ld + kb);
        } else if (vec == 2) {
            float2 p[NQ], q[NQ];
#pragma unroll
            for (int i = 0; i < NQ; i++) { p[i] = *reinterpret_cast<const float2 *>(base + row[i] * ld + kb); q[i] = *reinterpret_cast<const float2 *>(base + row[i] * ld + kb + 2); }
#pragma unroll
            for (int i = 0; i < NQ; i++) dst[i] = float4{p[i].x, p[i].y, q[i].x, q[i].y};
        } else {
            float e[NQ][4];
#pragma unroll
            for (int i = 0; i < NQ; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) e[i][j] = base[row[i] * ld + kb + j];
#pragma unroll
            for (int i = 0; i < NQ; i++) dst[i] = float4{e[i][0], e[i][1], e[i][2], e[i][3]};
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const float *rp = base + row[i] * ld;
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int k = kb + j; const int kc = k < kend ? k : kend - 1; const float v = rp[kc]; e[j] = k < kend ? v : 0.0f; }
        dst[i] = float4{e[0], e[1], e[2], e[3]};
    }
}

// x (times a power of two in the F16 arithmetic) -> (hi, lo) pairs of four values, written as two 8-byte stores
template <bool F16>
__device__ __forceinline__ void gb_split_store(const float4 &v, float scale, unsigned char *hi_img, unsigned char *lo_img, int byte_off)
{
    typename GbT<F16>::v4 h, l;
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (F16) x[j] *= scale;
        const typename GbT<F16>::e t = (typename GbT<F16>::e)x[j];
        h[j] = t; l[j] = (typename GbT<F16>::e)(x[j] - (float)t);
    }
    *reinterpret_cast<typename GbT<F16>::v4 *>(hi_img + byte_off) = h;
    *reinterpret_cast<typename GbT<F16>::v4 *>(lo_img + byte_off) = l;
}

// B tile of the workgroup's BN rows out of the split image: per [hi | lo] and k-step a contiguous BN x 32 bytes, which is also its LDS layout -- 16 bytes per thread
// (THREADS x 16 = BN x 32), no arithmetic
template <int BN>
__device__ __forceinline__ void gb_load_b(const unsigned char *bimg, int64_t bimg_half, int npad, int tile, int n0, int t, gb_u32x4 (&rb)[4])
{
#pragma unroll
    for (int c = 0; c < 4; c++)
        rb[c] = *reinterpret_cast<const gb_u32x4 *>(bimg + (c >> 1) * bimg_half + ((int64_t)(tile * 2 + (c & 1)) * npad + n0) * 32 + t * 16);
}
template <int BN>
__device__ __forceinline__ void gb_store_b(unsigned char *b_base, int b_half, int t, const gb_u32x4 (&rb)[4])
{
#pragma unroll
    for (int c = 0; c < 4; c++) *reinterpret_cast<gb_u32x4 *>(b_base + (c >> 1) * b_half + (c & 1) * BN * 32 + t * 16) = rb[c];
}

// Epilogue of the 256-wide tiles through LDS: the workgroup's C block goes out as WHOLE ROWS (1 KB contiguous per wave instruction, the whole 128 x 256 block contiguous
// when ldc == N) instead of 128-byte pieces of rows 1 KB apart (the same DRAM-page argument as for the A burst).  Row stride 260 floats: the two half-waves of an
// accumulator register write rows 4 apart, 4 x 1040 bytes = 64 bytes off in the banks.  The inverse scales (F16), the bias, the ReLU and the mask are applied at the
// write-out, once per row and four columns at a time, not per accumulator register.  All waves must be past their last LDS read of the K loop (a barrier) on entry.
// accumulators (NI row tiles of 32 from row `row0` of the block x the wave's 64 columns) -> the C block's LDS image
template <int NI>
__device__ __forceinline__ void gb_stage_c(const gb_f32x16 (&acc)[NI][2], unsigned char *smem, int row0, int wn, int r, int h)
{
    constexpr int CS = GbCfg<4>::BN + 4;
    float *ct = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int nl = wn * 64 + j * 32 + r;
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int q = 0; q < 16; q++) ct[(row0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * CS + nl] = acc[i][j][q];
    }
}

// The mask words of the rows a wave will write out (rows rw + NW i, i < 16: four 64-bit words each) with ONE load per lane -- lane 4 i + j takes word j of row i -- issued
// before the C block is staged, so that its latency is gone by the time the write-out loop wants the bits (a load per row INSIDE that loop cost the back-propagation
// products 0.3 ms of 0.8: eight dependent round trips per output tile).
template <int NW>
__device__ __forceinline__ uint64_t gb_preload_bits(const uint64_t *__restrict__ bits_in, int bits_in_ld, int64_t M, int64_t m0, int th)
{
    if (!bits_in) return 0;
    const int ln = th & 63, rw = th >> 6;
    const int64_t m = m0 + rw + NW * (ln >> 2);
    return m < M ? bits_in[m * bits_in_ld + (ln & 3)] : 0;
}

// the staged C block -> memory as whole rows, by `NW` waves (thread th of 64 NW): inverse scales, bias, ReLU, mask at the write-out
template <bool F16, int NW, int BM = GB_BM>
__device__ __forceinline__ void gb_readout_c(const unsigned char *smem, const float *rinv, float binv, float *__restrict__ c, int ldc, int64_t M, int N, const float *__restrict__ bias, int relu,
                                             const float *__restrict__ mask, int mask_ld, int64_t m0, int n0, int th, const float *__restrict__ add = nullptr, int add_ld = 0,
                                             const float *__restrict__ r1s = nullptr, int r1s_ld = 0, const float *__restrict__ r1w = nullptr, uint64_t *__restrict__ bits_out = nullptr,
                                             int bits_out_ld = 0, const uint64_t *__restrict__ bits_in = nullptr, uint64_t pre = 0)
{
    static_assert(BM / NW <= 16, "a wave's rows' mask words must fit its 64 lanes");
    constexpr int CS = GbCfg<4>::BN + 4;
    const float *ct = reinterpret_cast<const float *>(smem);
    const int c4 = (th & 63) * 4, rw = th >> 6;
    const bool vec_ok = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0) && (!mask || (((mask_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 15) == 0))) &&
                        (!add || (((add_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(add) & 15) == 0)));
    float bias4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) if (n0 + c4 + jj < N) bias4[jj] = bias[n0 + c4 + jj];
    }
    if (r1s) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) if (n0 + c4 + jj < N) w4[jj] = r1w[n0 + c4 + jj];
    }
#pragma unroll 4
    for (int i = 0; i < BM / NW; i++) {
        const int ml = rw + NW * i;
        const int64_t m = m0 + ml;
        const int n = n0 + c4;
        if (m >= M || n >= N) continue;
        float4 v = *reinterpret_cast<const float4 *>(ct + ml * CS + c4);
        if (F16) { const float rs = rinv[ml] * binv; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
        v.x += bias4[0]; v.y += bias4[1]; v.z += bias4[2]; v.w += bias4[3];
        if (r1s) { const float sv = r1s[m * r1s_ld]; v.x = v.x + sv * w4[0]; v.y = v.y + sv * w4[1]; v.z = v.z + sv * w4[2]; v.w = v.w + sv * w4[3]; }
        if (relu) { v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f; v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f; }
        if (bits_out) {          // (a wave = one row, a lane = four columns: four ballots are the row's mask; lanes of columns >= N are not here and leave zeros)
            const uint64_t b0 = __ballot(v.x > 0.0f), b1 = __ballot(v.y > 0.0f), b2 = __ballot(v.z > 0.0f), b3 = __ballot(v.w > 0.0f);
            const int ln = th & 63;
            if (ln < 4) bits_out[m * bits_out_ld + ln] = ln == 0 ? b0 : ln == 1 ? b1 : ln == 2 ? b2 : b3;
        }
        if (bits_in) {          // row i's four words sit in lanes 4 i .. 4 i + 3 of `pre` (gb_preload_bits): a lane exchange, no load inside this loop
            const int ln = th & 63;
            uint64_t w[4];
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
                w[jj] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(pre >> 32), 4 * i + jj) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pre, 4 * i + jj);
            v.x = ((w[0] >> ln) & 1) ? v.x : 0.0f; v.y = ((w[1] >> ln) & 1) ? v.y : 0.0f; v.z = ((w[2] >> ln) & 1) ? v.z : 0.0f; v.w = ((w[3] >> ln) & 1) ? v.w : 0.0f;
        }
        if (vec_ok && n + 4 <= N) {
            if (mask) {
                const float4 k = *reinterpret_cast<const float4 *>(mask + m * mask_ld + n);
                v.x = k.x > 0.0f ? v.x : 0.0f; v.y = k.y > 0.0f ? v.y : 0.0f; v.z = k.z > 0.0f ? v.z : 0.0f; v.w = k.w > 0.0f ? v.w : 0.0f;
            }
            if (add) { const float4 k = *reinterpret_cast<const float4 *>(add + m * add_ld + n); v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w; }
            *reinterpret_cast<float4 *>(c + m * ldc + n) = v;
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
            for (int jj = 0; jj < 4 && n + jj < N; jj++) {
                float x = e[jj];
                if (mask) x = mask[m * mask_ld + n + jj] > 0.0f ? x : 0.0f;
                if (add) x += add[m * add_ld + n + jj];
                c[m * ldc + n + jj] = x;
            }
        }
    }
}

template <bool F16>
__device__ __forceinline__ void gb_epilogue_rows(const gb_f32x16 (&acc)[2][2], unsigned char *smem, const float *rinv, float binv, float *__restrict__ c, int ldc, int64_t M, int N,
                                                 const float *__restrict__ bias, int relu, const float *__restrict__ mask, int mask_ld, int64_t m0, int n0, int t, int wm, int wn, int r, int h,
                                                 const float *__restrict__ add = nullptr, int add_ld = 0, const float *__restrict__ r1s = nullptr, int r1s_ld = 0,
                                                 const float *__restrict__ r1w = nullptr, uint64_t *__restrict__ bits_out = nullptr, int bits_out_ld = 0,
                                                 const uint64_t *__restrict__ bits_in = nullptr, int bits_in_ld = 0)
{
    const uint64_t pre = gb_preload_bits<8>(bits_in, bits_in_ld, M, m0, t);
    gb_stage_c<2>(acc, smem, wm * 64, wn, r, h);
    __syncthreads();
    gb_readout_c<F16, 8>(smem, rinv, binv, c, ldc, M, N, bias, relu, mask, mask_ld, m0, n0, t, add, add_ld, r1s, r1s_ld, r1w, bits_out, bits_out_ld, bits_in, pre);
}

constexpr int GB_ROWS_LDS_BASE = (GB_BM * (GbCfg<4>::BN + 4) * 4) > 2 * GbCfg<4>::STAGE ? (GB_BM * (GbCfg<4>::BN + 4) * 4) : 2 * GbCfg<4>::STAGE;     // the C tile (133 KB) or the two stages
constexpr int GB_ROWS_LDS = GB_ROWS_LDS_BASE + GB_BM * 4;                                                                                            // + the rows' inverse scales

// XCD: consecutive workgroup ids are dealt out round-robin to the 8 XCDs (each with its own L2): the launch is re-indexed so that XCD x works through a CONTIGUOUS
// eighth of the tiles -- the n-blocks of one m-block (which read the same A rows) and neighbouring m-blocks (which read the same B) then share an L2
template <int WNW, bool F16>
__global__ void __launch_bounds__(128 * WNW, WNW == 2 ? 2 : 1) k_gemm_nt(GemmNT g)
{
    using Cfg = GbCfg<WNW>;
    using T8 = typename GbT<F16>::v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nblocks_n = (g.N + Cfg::BN - 1) / Cfg::BN;
    int64_t bid = blockIdx.x;
    {
        const int64_t nb_all = gridDim.x, per = nb_all / 8;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);                       // (the remainder keeps its own index)
    }
    const int64_t mb = bid / nblocks_n;
    const int nb = (int)(bid - mb * nblocks_n);
    const int64_t m0 = mb * GB_BM;
    const int n0 = nb * Cfg::BN;
    const int wm = wave / WNW, wn = wave % WNW;
    // staging map: thread -> (row = rq + RSTEP i, four k at 4 kq)
    const int rq = t >> 3, kq = t & 7;
    const int t0 = (g.k0 + GB_BK - 1) / GB_BK, t1 = (g.k1 + GB_BK - 1) / GB_BK, T = t0 + t1;
    // two register sets: while tile t is multiplied out of LDS, tile t + 1's loads are landing and tile t + 2's are being issued (the kernel is latency-bound on its A
    // stream otherwise: one tile in flight per workgroup moved 1.9 TB/s, profiles/round6/r6m_gemm_probe.log)
    float4 ra0[Cfg::QA], ra1[Cfg::QA];
    gb_u32x4 rb0[4], rb1[4];                                                // B: the image's four 16-byte pieces of this thread per K tile ([hi | lo] x [k-step])
    int64_t arow[Cfg::QA];
#pragma unroll
    for (int i = 0; i < Cfg::QA; i++) { const int64_t m = m0 + rq + Cfg::RSTEP * i; arow[i] = m < g.M ? m : g.M - 1; }
    auto load_tile = [&](int tile, float4 (&ra)[Cfg::QA], gb_u32x4 (&rb)[4]) {
        const bool s1 = tile >= t0;
        const int tk = (s1 ? tile - t0 : tile) * GB_BK;                   // first column of the tile inside its segment
        const int kend = s1 ? g.k1 : g.k0;
        const bool full = tk + GB_BK <= kend;
        gb_load_rows<Cfg::QA>(s1 ? g.a1 : g.a0, s1 ? g.lda1 : g.lda0, arow, tk + 4 * kq, kend, full, s1 ? g.va1 : g.va0, ra);
        gb_load_b<Cfg::BN>(g.bimg, g.bimg_half, g.npad, tile, n0, t, rb);
    };
    // F16: the rows' scales need every row's largest entry BEFORE the first split -- one extra pass over the workgroup's A rows (their second read, by the K loop, comes
    // out of the caches); the inverse scales wait in LDS for the epilogue
    float as[Cfg::QA], binv = 1.0f;
    float *rinv = reinterpret_cast<float *>(gb_smem + (WNW == 4 ? GB_ROWS_LDS_BASE : 2 * Cfg::STAGE));          // (the 256-wide tile stages its C block through LDS: past it)
#pragma unroll
    for (int i = 0; i < Cfg::QA; i++) as[i] = 1.0f;
    if (F16) {
        { float bs; gb_pow2_scale(__uint_as_float(g.bmax[0]), bs, binv); }
        float mx[Cfg::QA];
#pragma unroll
        for (int i = 0; i < Cfg::QA; i++) mx[i] = 0.0f;
        for (int tile = 0; tile < T; tile++) {
            const bool s1 = tile >= t0;
            const int tk = (s1 ? tile - t0 : tile) * GB_BK;
            const int kend = s1 ? g.k1 : g.k0;
            gb_load_rows<Cfg::QA>(s1 ? g.a1 : g.a0, s1 ? g.lda1 : g.lda0, arow, tk + 4 * kq, kend, tk + GB_BK <= kend, s1 ? g.va1 : g.va0, ra0);
#pragma unroll
            for (int i = 0; i < Cfg::QA; i++) mx[i] = fmaxf(mx[i], gb_absmax4(ra0[i]));
        }
#pragma unroll
        for (int i = 0; i < Cfg::QA; i++) {
            float m = mx[i];
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));          // the row's eight threads (kq) are adjacent lanes
            float inv;
            gb_pow2_scale(m, as[i], inv);
            if (kq == 0) rinv[rq + Cfg::RSTEP * i] = inv;
        }
    }
    auto store_tile = [&](int stage, const float4 (&ra)[Cfg::QA], const gb_u32x4 (&rb)[4]) {
        unsigned char *base = gb_smem + stage * Cfg::STAGE;
        const int ks = kq >> 2;                                          // k-step of 16 inside the tile
#pragma unroll
        for (int i = 0; i < Cfg::QA; i++) gb_split_store<F16>(ra[i], as[i], base, base + Cfg::A_HALF, (ks * GB_BM + rq + Cfg::RSTEP * i) * 32 + (kq & 3) * 8);
        gb_store_b<Cfg::BN>(base + 2 * Cfg::A_HALF, Cfg::B_HALF, t, rb);
    };
    gb_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    auto multiply = [&](int stage) {
        const unsigned char *base = gb_smem + stage * Cfg::STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            T8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int offa = (ks * GB_BM + wm * 64 + i * 32 + r) * 32 + h * 16;
                ah[i] = *reinterpret_cast<const T8 *>(base + offa);
                al[i] = *reinterpret_cast<const T8 *>(base + Cfg::A_HALF + offa);
                const int offb = (ks * Cfg::BN + wn * 64 + i * 32 + r) * 32 + h * 16;
                bh[i] = *reinterpret_cast<const T8 *>(base + 2 * Cfg::A_HALF + offb);
                bl[i] = *reinterpret_cast<const T8 *>(base + 2 * Cfg::A_HALF + Cfg::B_HALF + offb);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    // small terms first, the leading product last
                    acc[i][j] = GbT<F16>::mfma(al[i], bh[j], acc[i][j]);
                    acc[i][j] = GbT<F16>::mfma(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = GbT<F16>::mfma(ah[i], bh[j], acc[i][j]);
                }
        }
    };
    // tile t lives in register set t & 1 and LDS stage t & 1
    if (T > 0) load_tile(0, ra0, rb0);
    if (T > 1) load_tile(1, ra1, rb1);
    if (T > 0) store_tile(0, ra0, rb0);
    __syncthreads();
    for (int tile = 0; tile < T; tile += 2) {
        // even tile: set 0 is free (stored), set 1 holds tile + 1
        if (tile + 2 < T) load_tile(tile + 2, ra0, rb0);
        multiply(0);
        if (tile + 1 < T) store_tile(1, ra1, rb1);
        __syncthreads();
        if (tile + 1 >= T) break;
        // odd tile: set 1 is free, set 0 holds tile + 2
        if (tile + 3 < T) load_tile(tile + 3, ra1, rb1);
        multiply(1);
        if (tile + 2 < T) store_tile(0, ra0, rb0);
        __syncthreads();
    }
    if (WNW == 4) {          // whole rows through LDS, as the burst kernel's
        gb_epilogue_rows<F16>(acc, gb_smem, rinv, binv, g.c, g.ldc, g.M, g.N, g.bias, g.relu, g.mask, g.mask_ld, m0, n0, t, wm, wn, r, h, g.add, g.add_ld, g.r1s, g.r1s_ld, g.r1w, g.bits_out, g.bits_out_ld, g.bits_in, g.bits_in_ld);
        return;
    }
    // epilogue: register q of lane (r, h) of tile (i, j) is C[m0 + 64 wm + 32 i + (q & 3) + 8 (q >> 2) + 4 h][n0 + 64 wn + 32 j + r]
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int ml = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                const int64_t m = m0 + ml;
                if (m >= g.M) continue;
                float v = (F16 ? acc[i][j][q] * rinv[ml] * binv : acc[i][j][q]) + bias;
                if (g.r1s) v = v + g.r1s[m * g.r1s_ld] * g.r1w[n];
                if (g.relu) v = v > 0.0f ? v : 0.0f;
                if (g.mask) v = g.mask[m * g.mask_ld + n] > 0.0f ? v : 0.0f;
                if (g.add) v += g.add[m * g.add_ld + n];
                g.c[m * g.ldc + n] = v;
            }
    }
}

#ifdef NRF_GB_TRACE
// diagnostic build only (tools/scratch/gemm_trace.py): clock stamps of waves 0 and 7 of every workgroup of k_gemm_nt_rows, summed per section
__device__ unsigned long long g_gb_trace[256 * 2 * 16];
#define NRF_GSTAMP(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tr[i] += t__ - tprev; tprev = t__; } while (0)
#else
#define NRF_GSTAMP(i) do { } while (0)
#endif

// The same product with the WHOLE A block of the workgroup (128 rows x K <= 256 columns: one contiguous 128 KB of a [M][K] array) requested in one burst at kernel
// start: with K tiles of 32 requested one by one every row is visited eight times, 128 bytes at a time, microseconds apart -- DRAM pages are re-opened for each piece and
// the A stream (what bounds this product: 64 flop per byte at N = K = 256) moved ~2 TB/s for this kernel and for rocBLAS's alike (profiles/round6/r6m_gemm_probe.log).
// TK = K / 32 tiles live in 8 TK registers per thread; B (the weights, L2-resident) is streamed per tile as before.  One segment, K a multiple of 32, 16-byte aligned rows.
// NRF_GB_ABL (diagnostic builds only, tools/scratch/gemm_ablate.sh; results are garbage): parts of this kernel left out to see what its time is made of.
//   1 no multiply (fragment reads + matrix instructions)   2 no epilogue   4 epilogue staged in LDS but not written out   8 A not split / stored to LDS
//   16 B tiles neither loaded nor stored   32 A not loaded   64 matrix instructions left out, fragment reads kept
#ifndef NRF_GB_ABL
#define NRF_GB_ABL 0
#endif
template <int TK, bool F16>
__global__ void __launch_bounds__(512, 1) k_gemm_nt_rows(GemmNT g)
{
    constexpr int ABL = NRF_GB_ABL;
    using Cfg = GbCfg<4>;
    using T8 = typename GbT<F16>::v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nblocks_n = (g.N + Cfg::BN - 1) / Cfg::BN;
    const int64_t total = ((g.M + GB_BM - 1) / GB_BM) * nblocks_n;
    const int wm = wave / 4, wn = wave % 4;
    const int rq = t >> 3, kq = t & 7;
    float4 ra[TK][Cfg::QA];
    gb_u32x4 rb0[4], rb1[4];
    const int t0 = g.k0 / GB_BK;
    // PERSISTENT: one workgroup per CU walks the output tiles bid, bid + gridDim.x, ...  With one tile per workgroup all 256 workgroups of a round read their A blocks,
    // then all compute, then all write: 17 us of memory phases + 14 us of compute per round, one after the other (0.65 ms per 786 432 x 256 x 256).  Here the NEXT tile's
    // A block is requested while the current one is multiplied -- into the very registers the current tile has just handed to LDS (tile k's four values per row are dead
    // once store_tile(k) has split them), so the prefetch costs no register -- and the C block of the current tile leaves while the next one's loads are in flight.
    // the A rows of an output tile as per-thread pointers (row rq + RSTEP i, columns 4 kq ..): computed once per output tile, the K tiles are immediate offsets
    const float *ap0[Cfg::QA], *ap1[Cfg::QA];
    auto point_a = [&](int64_t b) {          // (tile numbers fit 31 bits: a 32-bit division, not the 64-bit one's ~1 us per output tile)
        const int64_t m0b = (int64_t)((uint32_t)b / (uint32_t)nblocks_n) * GB_BM;
#pragma unroll
        for (int i = 0; i < Cfg::QA; i++) {
            const int64_t m = m0b + rq + Cfg::RSTEP * i;
            const int64_t mc = m < g.M ? m : g.M - 1;
            ap0[i] = g.a0 + mc * g.lda0 + 4 * kq;
            ap1[i] = g.a1 ? g.a1 + mc * g.lda1 + 4 * kq : ap0[i];
        }
    };
    auto issue_a = [&](int tile) {          // K tile `tile` of the pointed-at rows -> ra[tile][*] (segment 0's K tiles, then segment 1's)
#pragma unroll
        for (int i = 0; i < Cfg::QA; i++) {
            if (ABL & 32) ra[tile][i] = float4{1.0f + tile, 2.0f, 3.0f + i, 4.0f};
            else ra[tile][i] = *reinterpret_cast<const float4 *>(tile < t0 ? ap0[i] + tile * GB_BK : ap1[i] + (tile - t0) * GB_BK);
        }
    };
    int64_t bid = blockIdx.x;
    if (bid < total) {
        point_a(bid);
#pragma unroll
        for (int tile = 0; tile < TK; tile++) issue_a(tile);
    }
    float *rinv = reinterpret_cast<float *>(gb_smem + GB_ROWS_LDS_BASE);
#ifdef NRF_GB_TRACE
    unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_readcyclecounter();
    unsigned long long tprev = tstart;
#endif
    for (; bid < total; bid += gridDim.x) {
        const int64_t mb = (int64_t)((uint32_t)bid / (uint32_t)nblocks_n);
        const int nb = (int)(bid - mb * nblocks_n);
        const int64_t m0 = mb * GB_BM;
        const int n0 = nb * Cfg::BN;
        const int64_t nbid = bid + gridDim.x;
        const bool has_next = nbid < total;
        if (has_next) point_a(nbid);
        NRF_GSTAMP(0);
        // A NARROW product on this 256-wide tile (N = 33: the LeRF sigma net's last layer; N = 128): the column groups of 64 past N do nothing -- their waves skip the
        // fragment reads and matrix instructions, their B rows are neither loaded nor stored, their part of the C block is not staged (the write-out skips n >= N)
        const int nact = g.N - n0 >= Cfg::BN ? Cfg::BN : ((g.N - n0 + 63) & ~63);          // columns of this tile that exist, rounded up to a wave's 64
        const bool wave_on = wn * 64 < nact, b_on = (t >> 1) < nact;                        // (a thread's 16-byte B piece belongs to row t / 2 of a slab)
        // (B's columns run straight through both segments)
        auto load_b = [&](int tile, gb_u32x4 (&rb)[4]) { if (!(ABL & 16) && b_on) gb_load_b<Cfg::BN>(g.bimg, g.bimg_half, g.npad, tile, n0, t, rb); };
        // F16: the row's whole K sits in the eight threads (kq) of the row: its largest entry is 8 TK maxima and three lane exchanges away
        float as[Cfg::QA], binv = 1.0f;
#pragma unroll
        for (int i = 0; i < Cfg::QA; i++) as[i] = 1.0f;
        if (F16) {
            { float bs; gb_pow2_scale(__uint_as_float(g.bmax[0]), bs, binv); }
#pragma unroll
            for (int i = 0; i < Cfg::QA; i++) {
                float m = 0.0f;
#pragma unroll
                for (int tile = 0; tile < TK; tile++) m = fmaxf(m, gb_absmax4(ra[tile][i]));
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                float inv;
                gb_pow2_scale(m, as[i], inv);
                if (kq == 0) rinv[rq + Cfg::RSTEP * i] = inv;
            }
        }
        auto store_tile = [&](int stage, const float4 (&a)[Cfg::QA], const gb_u32x4 (&rb)[4]) {
            unsigned char *base = gb_smem + stage * Cfg::STAGE;
            const int ks = kq >> 2;
#pragma unroll
            for (int i = 0; i < Cfg::QA; i++) {
                if (ABL & 8) { if (a[i].x == 12345.678f) base[t] = 1; }      // (keeps the loads alive)
                else gb_split_store<F16>(a[i], as[i], base, base + Cfg::A_HALF, (ks * GB_BM + rq + Cfg::RSTEP * i) * 32 + (kq & 3) * 8);
            }
            if (!(ABL & 16) && b_on) gb_store_b<Cfg::BN>(base + 2 * Cfg::A_HALF, Cfg::B_HALF, t, rb);
        };
        gb_f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
        auto multiply = [&](int stage) {
            if ((ABL & 1) || !wave_on) return;
            const unsigned char *base = gb_smem + stage * Cfg::STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                T8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int offa = (ks * GB_BM + wm * 64 + i * 32 + r) * 32 + h * 16;
                    ah[i] = *reinterpret_cast<const T8 *>(base + offa);
                    al[i] = *reinterpret_cast<const T8 *>(base + Cfg::A_HALF + offa);
                    const int offb = (ks * Cfg::BN + wn * 64 + i * 32 + r) * 32 + h * 16;
                    bh[i] = *reinterpret_cast<const T8 *>(base + 2 * Cfg::A_HALF + offb);
                    bl[i] = *reinterpret_cast<const T8 *>(base + 2 * Cfg::A_HALF + Cfg::B_HALF + offb);
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (ABL & 64) {                                     // fragment reads without the matrix instructions
                            const gb_u32x4 x = __builtin_bit_cast(gb_u32x4, al[i]) ^ __builtin_bit_cast(gb_u32x4, bh[j]) ^ __builtin_bit_cast(gb_u32x4, ah[i]) ^ __builtin_bit_cast(gb_u32x4, bl[j]);
                            acc[i][j][0] += __uint_as_float(x[0] ^ x[1] ^ x[2] ^ x[3]);
                            continue;
                        }
                        acc[i][j] = GbT<F16>::mfma(al[i], bh[j], acc[i][j]);
                        acc[i][j] = GbT<F16>::mfma(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = GbT<F16>::mfma(ah[i], bh[j], acc[i][j]);
                    }
            }
        };
        NRF_GSTAMP(1);                  // (F16: the row maxima = the wait for the whole A block)
        load_b(0, rb0);
        if (TK > 1) load_b(1, rb1);
        store_tile(0, ra[0], rb0);
        NRF_GSTAMP(2);
        __syncthreads();
        NRF_GSTAMP(3);
#pragma unroll
        for (int tile = 0; tile < TK; tile++) {
            // B of tile + 2 into the set that tile's store has freed
            if (tile + 2 < TK) { if (tile & 1) load_b(tile + 2, rb1); else load_b(tile + 2, rb0); }
            multiply(tile & 1);
            NRF_GSTAMP(4);
            if (tile + 1 < TK) { if (tile & 1) store_tile(0, ra[tile + 1 < TK ? tile + 1 : 0], rb0); else store_tile(1, ra[tile + 1 < TK ? tile + 1 : 0], rb1); }
            // ra[tile] went to LDS one iteration ago (tile 0: before the loop): the next output tile's K tile `tile` into it
            if (has_next) issue_a(tile);
            NRF_GSTAMP(5);
            __syncthreads();
            NRF_GSTAMP(3);
        }
        if (ABL & 2) { if (acc[0][0][0] + acc[1][1][3] + acc[0][1][5] + acc[1][0][7] == 12345.678f) g.c[t] = 1.0f; }
        else {
            const uint64_t pre = gb_preload_bits<8>(g.bits_in, g.bits_in_ld, g.M, m0, t);
            if (wave_on) gb_stage_c<2>(acc, gb_smem, wm * 64, wn, r, h);
            NRF_GSTAMP(6);
            __syncthreads();
            NRF_GSTAMP(7);
            gb_readout_c<F16, 8>(gb_smem, rinv, binv, g.c, g.ldc, (ABL & 4) ? (int64_t)(g.N < 0) : g.M, g.N, g.bias, g.relu, g.mask, g.mask_ld, m0, n0, t, g.add, g.add_ld, g.r1s, g.r1s_ld, g.r1w, g.bits_out, g.bits_out_ld, g.bits_in, pre);
            NRF_GSTAMP(8);
        }
        __syncthreads();          // the C block staged in LDS is read out: the next tile's stages may overwrite it
        NRF_GSTAMP(9);
    }
#ifdef NRF_GB_TRACE
    tr[15] = __builtin_readcyclecounter() - tstart;
    if ((wave == 0 || wave == 7) && lane == 0) for (int i = 0; i < 16; i++) g_gb_trace[((blockIdx.x & 255) * 2 + (wave ? 1 : 0)) * 16 + i] += tr[i];
#endif
}


// NRF_POISON=1 (debugging): every stream-ordered scratch buffer of this file is filled with NaN patterns before use -- a kernel that reads what it did not write shows
static void gb_poison(void *p, size_t bytes, hipStream_t st)
{
    static const bool on = [] { const char *e = getenv("NRF_POISON"); return e && atoi(e) != 0; }();
    if (on && p) (void)hipMemsetAsync(p, 0xff, bytes, st);
}

static int vec_class(const float *p, int ld, int col0)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p + col0);
    if ((a & 15) == 0 && (ld & 3) == 0) return 4;
    if ((a & 7) == 0 && (ld & 1) == 0) return 2;
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
// The WEIGHT GRADIENT  dW [out x n] += G [P x out]^T . X [P x n]:  both operands are row-major over the POINTS, i.e. the contraction runs over their slow index -- a
// "TN" product whose K is 10^5..10^6 and whose result is one small matrix.  (rocBLAS needs 32 strided-batch slices + a sum to fill the chip with it: 6-9 ms of a
// training step.)  Here: 128 x 128 output tile per 256-thread workgroup (2 x 2 waves of 64 x 64) over one SLICE of the points; the slices' partial matrices are summed
// by k_sum_partials in a fixed order (deterministic, no atomics).  K tile = 32 points:
//   * a thread loads, for ONE operand (waves 0-1: G, waves 2-3: X), four consecutive columns of eight consecutive points (eight 16-byte loads; the 32 lanes of a
//     half-wave cover 512 contiguous bytes of a point's row) -- that is four columns x eight k values = four 16-byte MFMA operand pieces after the split: the transpose
//     happens in registers, four ds_write_b128 per half.
//   * LDS image per operand and half: [k-step][position][16 elements] with position = (column & 3) * 32 + (column >> 2): one write instruction's 64 lanes (32 column
//     quads x the two 8-point halves of a k-step) cover one contiguous KB, and a fragment read (32 positions) another: no bank conflicts either way.  MFMA tile t,
//     row r is therefore column 4 r + t of the workgroup's 128.
//   * arithmetic: bf16x3 (hi + lo bf16, fp32's exponent range: a column of G spans many orders of magnitude over the points, and a per-POINT scale cannot be undone in a
//     sum over points).  16 significant bits in a LEAF product -- nothing is computed from dW inside the step -- measured 5e-6 of its largest entry.
struct GemmTN {
    const float *g; int ldg, out;         // G [P][ldg], columns [0, out)
    const float *x; int ldx, n;           // X [P][ldx], columns [0, n)
    const float *x1; int ldx1, n1;        // an optional SECOND X segment (tiles bi >= nbi0): the same G tile is then read once from memory and once from the L2, not twice
    int nbi0;                             // output tiles along n that belong to segment 0
    int64_t P, slice_pts;                 // points (a multiple of 32), points per slice (a multiple of 32)
    float *part;                          // [slices][out][n + n1]
    float *bpart;                         // optional [slices][out]: the slice's column sums of G (the layer's bias gradient), by the workgroups of X tile 0
    int vg, vx;                           // 4: rows 16-byte aligned and the column count a multiple of 4 (vector loads); 1: scalar loads
    int nbo, nbi;                         // output tiles along out / n
    const float *xscale; int xscale_ld;   // optional: segment 0's row p is multiplied by xscale[p * xscale_ld] on its way to LDS (dW = G^T diag(s) X without a scaled copy of X)
};

constexpr int TN_T = 128;                                                 // tile edge
constexpr int TN_OP = 2 * 2 * TN_T * 32;                                  // bytes of one operand's [hi | lo][k-step][position][16] image
constexpr int TN_STAGE = 2 * TN_OP;

template <bool VEC>          // VEC: both operands take 16-byte loads; otherwise both go element by element with clamped columns (unaligned segments, ragged widths)
__global__ void __launch_bounds__(256, 2) k_gemm_tn(GemmTN a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    int64_t bid = blockIdx.x;
    {
        const int64_t nb_all = gridDim.x, per = nb_all / 8;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);           // XCD x works through a contiguous eighth: the tiles of one slice share an L2
    }
    const int tiles = a.nbo * a.nbi;
    const int64_t slice = bid / tiles;
    const int tl = (int)(bid - slice * tiles);
    const int bo = tl / a.nbi, bi = tl - bo * a.nbi;
    const int64_t p_begin = slice * a.slice_pts;
    int64_t p_end = p_begin + a.slice_pts; if (p_end > a.P) p_end = a.P;
    const int T = (int)((p_end - p_begin) / 32);
    // staging map
    const int opnd = t >> 7, unit = t & 127, cq = unit & 31, pg = unit >> 5;          // pg: which eight of the tile's 32 points; waves 0-1: G, 2-3: X
    const bool seg1 = bi >= a.nbi0;
    const int bil = seg1 ? bi - a.nbi0 : bi;
    const float *src = opnd ? (seg1 ? a.x1 : a.x) : a.g;
    const int ld = opnd ? (seg1 ? a.ldx1 : a.ldx) : a.ldg, ncols = opnd ? (seg1 ? a.n1 : a.n) : a.out;
    int col = (opnd ? bil : bo) * TN_T + 4 * cq;
    // columns past the operand's end are clamped to valid memory: what they produce lands in output rows / columns that are not stored
    int cj[4];
#pragma unroll
    for (int j = 0; j < 4; j++) cj[j] = col + j < ncols ? col + j : ncols - 1;
    if (VEC && col + 4 > ncols) col = ncols - 4;
    float4 r0[8];
    float sc0[8];
    const bool scaled = a.xscale && opnd == 1 && !seg1;            // (wave-uniform)
    const bool bsum_on = a.bpart && opnd == 0 && bi == 0;          // (wave-uniform: waves 0-1 of the workgroups of X tile 0)
    float bs0 = 0.0f, bs1 = 0.0f, bs2 = 0.0f, bs3 = 0.0f;          // this thread's four G columns, summed over its eight points of every K tile
    auto load_tile = [&](int tile, float4 (&rr)[8]) {
        const float *rp = src + (p_begin + (int64_t)tile * 32 + pg * 8) * ld;
        if (scaled) {
#pragma unroll
            for (int k = 0; k < 8; k++) sc0[k] = a.xscale[(p_begin + (int64_t)tile * 32 + pg * 8 + k) * a.xscale_ld];
        }
        if (VEC) {
#pragma unroll
            for (int k = 0; k < 8; k++) rr[k] = *reinterpret_cast<const float4 *>(rp + (int64_t)k * ld + col);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float *q = rp + (int64_t)k * ld;
                rr[k].x = q[cj[0]]; rr[k].y = q[cj[1]]; rr[k].z = q[cj[2]]; rr[k].w = q[cj[3]];
            }
        }
    };
    auto store_tile = [&](int stage, const float4 (&rr)[8]) {
        unsigned char *base = gb_smem + stage * TN_STAGE + opnd * TN_OP;
        const int ks = pg >> 1, hh = pg & 1;
        if (bsum_on) {
#pragma unroll
            for (int k = 0; k < 8; k++) { bs0 += rr[k].x; bs1 += rr[k].y; bs2 += rr[k].z; bs3 += rr[k].w; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            gb_bf16x8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float v = j == 0 ? rr[k].x : (j == 1 ? rr[k].y : (j == 2 ? rr[k].z : rr[k].w));
                if (scaled) v = sc0[k] * v;
                const __bf16 b = (__bf16)v;
                hi[k] = b; lo[k] = (__bf16)(v - (float)b);
            }
            const int off = (ks * TN_T + j * 32 + cq) * 32 + hh * 16;
            *reinterpret_cast<gb_bf16x8 *>(base + off) = hi;
            *reinterpret_cast<gb_bf16x8 *>(base + TN_OP / 2 + off) = lo;
        }
    };
    const int wm = wave >> 1, wn = wave & 1;
    gb_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    auto multiply = [&](int stage) {
        const unsigned char *base = gb_smem + stage * TN_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            gb_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int offa = (ks * TN_T + (2 * wm + i) * 32 + r) * 32 + h * 16;
                ah[i] = *reinterpret_cast<const gb_bf16x8 *>(base + offa);
                al[i] = *reinterpret_cast<const gb_bf16x8 *>(base + TN_OP / 2 + offa);
                const int offb = (ks * TN_T + (2 * wn + i) * 32 + r) * 32 + h * 16;
                bh[i] = *reinterpret_cast<const gb_bf16x8 *>(base + TN_OP + offb);
                bl[i] = *reinterpret_cast<const gb_bf16x8 *>(base + TN_OP + TN_OP / 2 + offb);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = GbT<false>::mfma(al[i], bh[j], acc[i][j]);
                    acc[i][j] = GbT<false>::mfma(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = GbT<false>::mfma(ah[i], bh[j], acc[i][j]);
                }
        }
    };
    // ONE register set: tile t + 1 is requested before tile t is multiplied, and split into the other LDS stage behind it; the CU's second workgroup covers what
    // latency is left.  (Two sets with tiles t + 1, t + 2 in flight need 256 registers and spill -- or, with the k-step loop kept rolled to make room, the compiler
    // ping-pongs the accumulators between two register sets: 575-630 us against 544 for the 786 432 x 256 x 256 product.)
    if (T > 0) { load_tile(0, r0); store_tile(0, r0); }
    __syncthreads();
    for (int tile = 0; tile < T; tile++) {
        if (tile + 1 < T) load_tile(tile + 1, r0);
        multiply(tile & 1);
        if (tile + 1 < T) store_tile((tile + 1) & 1, r0);
        __syncthreads();
    }
    if (a.bpart && bi == 0) {
        // the four point groups' sums of every column quad through LDS (the K loop's last barrier is behind us), added in group order
        float *bl = reinterpret_cast<float *>(gb_smem);
        if (opnd == 0) { bl[(pg * 32 + cq) * 4 + 0] = bs0; bl[(pg * 32 + cq) * 4 + 1] = bs1; bl[(pg * 32 + cq) * 4 + 2] = bs2; bl[(pg * 32 + cq) * 4 + 3] = bs3; }
        __syncthreads();
        if (t < TN_T) {
            const int o = bo * TN_T + t;
            if (o < a.out) a.bpart[(size_t)slice * a.out + o] = ((bl[t] + bl[128 + t]) + bl[256 + t]) + bl[384 + t];
        }
    }
    // register q of lane (r, h) of tile pair (i, j): G column o = 128 bo + 4 ((q & 3) + 8 (q >> 2) + 4 h) + (2 wm + i), X column c = 128 bi + 4 r + (2 wn + j)
    const int nn = a.n + a.n1;
    float *dst = a.part + (size_t)slice * a.out * nn;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int cl = bil * TN_T + 4 * r + 2 * wn + j;
            if (cl >= (seg1 ? a.n1 : a.n)) continue;
            const int c = seg1 ? a.n + cl : cl;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int o = bo * TN_T + 4 * ((q & 3) + 8 * (q >> 2) + 4 * h) + 2 * wm + i;
                if (o < a.out) dst[(size_t)o * nn + c] = acc[i][j][q];
            }
        }
}

// dw += the slices' partial matrices, in slice order.  Columns [0, n0) of a partial row go to dw columns col0 ..; columns n0 + skip1 .. n0 + skip1 + keep1 to col1 ..
// (the other columns of the second segment are computed but not wanted: an aligned read of [sigma | geo | padding] for the geo columns)
__global__ void k_tn_sum(int slices, int out, int n0, int n1, int in, int col0, int col1, int skip1, int keep1, const float *__restrict__ part, float *__restrict__ dw)
{
    const int nn = n0 + n1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= out * nn) return;
    const int o = e / nn, i = e - o * nn;
    if (i >= n0 && (i - n0 < skip1 || i - n0 >= skip1 + keep1)) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = 0;
    for (; b + 4 <= slices; b += 4) {
        s0 += part[(size_t)b * out * nn + e]; s1 += part[(size_t)(b + 1) * out * nn + e]; s2 += part[(size_t)(b + 2) * out * nn + e]; s3 += part[(size_t)(b + 3) * out * nn + e];
    }
    for (; b < slices; b++) s0 += part[(size_t)b * out * nn + e];
    const int c = i < n0 ? col0 + i : col1 + (i - n0 - skip1);
    dw[(size_t)o * in + c] += (s0 + s1) + (s2 + s3);
}

// db[o] += the slices' column sums of G: one wave per column, lane l adds slices l, l + 64, ... and the lanes' sums are combined in a fixed butterfly (deterministic)
__global__ void __launch_bounds__(64) k_tn_bias_sum(int slices, int out, const float *__restrict__ bpart, float *__restrict__ db)
{
    const int o = blockIdx.x, lane = threadIdx.x;
    float s = 0.0f;
    for (int b = lane; b < slices; b += 64) s += bpart[(size_t)b * out + o];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) db[o] += s;
}

// the last P % 32 points (fp32 FMAs, one thread per dw entry)
__global__ void k_tn_tail(int pts, int out, int n, const float *__restrict__ g, int ldg, const float *__restrict__ x, int ldx, int in, int col0, float *__restrict__ dw,
                          const float *__restrict__ xscale = nullptr, int xscale_ld = 0)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= out * n) return;
    const int o = e / n, i = e - o * n;
    float acc = 0.0f;
    for (int p = 0; p < pts; p++) acc = fmaf(g[(size_t)p * ldg + o], xscale ? xscale[(size_t)p * xscale_ld] * x[(size_t)p * ldx + i] : x[(size_t)p * ldx + i], acc);
    dw[(size_t)o * in + col0 + i] += acc;
}

// db[o] += sum over the last P % 32 points of g[p][o]
__global__ void k_tn_tail_bias(int pts, int out, const float *__restrict__ g, int ldg, float *__restrict__ db)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out) return;
    float acc = 0.0f;
    for (int p = 0; p < pts; p++) acc += g[(size_t)p * ldg + o];
    db[o] += acc;
}

// dw [out][in] += g^T [x | x1] over P points (and, with db, db[o] += the column sums of g: the layer's bias gradient out of the same pass): x's columns land at dw columns col0 .., x1's columns skip1 .. skip1 + keep1 at col1 .. (x1.n == 0: one segment; x1.n may be
// rounded up past skip1 + keep1 for aligned 16-byte reads: the row must hold that many floats, what they contain does not matter)
int gemm_tn_bf16x3_2(int64_t P, Seg g, Seg x, int col0, Seg x1, int col1, int skip1, int keep1, int out, int in, float *dw, hipStream_t st, float *db, const float *xscale,
                     int xscale_ld)
{
    if (P <= 0 || x.n <= 0 || out <= 0) return NRF_OK;
    const bool two = x1.p && keep1 > 0 && x1.n >= skip1 + keep1;
    const int64_t P32 = P & ~(int64_t)31;
    if (P32 < P) {
        const unsigned nb = (unsigned)ceil_div((int64_t)out * x.n, (int64_t)256);
        hipLaunchKernelGGL(k_tn_tail, dim3(nb), dim3(256), 0, st, (int)(P - P32), out, x.n, g.p + g.off + P32 * g.stride, g.stride, x.p + x.off + P32 * x.stride, x.stride, in, col0, dw,
                           xscale ? xscale + P32 * xscale_ld : nullptr, xscale_ld);
        if (two)
            hipLaunchKernelGGL(k_tn_tail, dim3((unsigned)ceil_div((int64_t)out * keep1, (int64_t)256)), dim3(256), 0, st, (int)(P - P32), out, keep1,
                               g.p + g.off + P32 * g.stride, g.stride, x1.p + x1.off + skip1 + P32 * x1.stride, x1.stride, in, col1, dw);
        if (db) hipLaunchKernelGGL(k_tn_tail_bias, dim3((unsigned)ceil_div((int64_t)out, (int64_t)256)), dim3(256), 0, st, (int)(P - P32), out, g.p + g.off + P32 * g.stride, g.stride, db);
        NRF_LAUNCH_CHECK();
    }
    if (P32 == 0) return NRF_OK;
    GemmTN a{};
    a.g = g.p + g.off; a.ldg = g.stride; a.out = out;
    a.x = x.p + x.off; a.ldx = x.stride; a.n = x.n;
    if (two) { a.x1 = x1.p + x1.off; a.ldx1 = x1.stride; a.n1 = x1.n; }
    a.P = P32;
    a.xscale = xscale; a.xscale_ld = xscale_ld;
    a.vg = (vec_class(a.g, a.ldg, 0) == 4 && (out & 3) == 0) ? 4 : 1;
    a.vx = (vec_class(a.x, a.ldx, 0) == 4 && (x.n & 3) == 0 && (!two || (vec_class(a.x1, a.ldx1, 0) == 4 && (x1.n & 3) == 0))) ? 4 : 1;
    a.nbo = (out + TN_T - 1) / TN_T; a.nbi0 = (x.n + TN_T - 1) / TN_T; a.nbi = a.nbi0 + (two ? (x1.n + TN_T - 1) / TN_T : 0);
    const int tiles = a.nbo * a.nbi, nn = a.n + a.n1;
    // slices: two rounds of the chip's 512 resident workgroups (each slice's partial matrix is written and read once more: 512 slices cost more in k_tn_sum than their
    // balance bought), at least 8 K tiles (256 points) each
    int64_t slices = (1024 + tiles - 1) / tiles;
    const int64_t max_slices = (P32 / 32 + 7) / 8;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    a.slice_pts = ((P32 / 32 + slices - 1) / slices) * 32;
    slices = (P32 + a.slice_pts - 1) / a.slice_pts;
    float *part = nullptr;
    if (scratch_take(reinterpret_cast<void **>(&part), ((size_t)slices * out * nn + (db ? (size_t)slices * out : 0)) * sizeof(float), st) != hipSuccess) {
        set_error("gemm_tn_bf16x3: scratch allocation failed");
        return NRF_ERR_HIP;
    }
    gb_poison(part, ((size_t)slices * out * nn + (db ? (size_t)slices * out : 0)) * sizeof(float), st);
    a.part = part;
    a.bpart = db ? part + (size_t)slices * out * nn : nullptr;
    static PerDeviceOnce attr;          // (the large dynamic LDS window is a per-device function attribute: common.h)
    if (attr.needed()) {
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_tn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN_STAGE));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_tn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN_STAGE));
        attr.done();
    }
    if (a.vg == 4 && a.vx == 4) hipLaunchKernelGGL(k_gemm_tn<true>, dim3((unsigned)(slices * tiles)), dim3(256), 2 * TN_STAGE, st, a);
    else hipLaunchKernelGGL(k_gemm_tn<false>, dim3((unsigned)(slices * tiles)), dim3(256), 2 * TN_STAGE, st, a);
    hipLaunchKernelGGL(k_tn_sum, dim3((unsigned)ceil_div((int64_t)out * nn, (int64_t)256)), dim3(256), 0, st, (int)slices, out, a.n, a.n1, in, col0, col1, two ? skip1 : 0, two ? keep1 : 0, (const float *)part, dw);
    if (db) hipLaunchKernelGGL(k_tn_bias_sum, dim3((unsigned)out), dim3(64), 0, st, (int)slices, out, (const float *)a.bpart, db);
    const hipError_t le = hipGetLastError();
    (void)scratch_give(part, st);
    if (le != hipSuccess) { set_error("gemm_tn_bf16x3: launch failed: %s", hipGetErrorString(le)); return NRF_ERR_HIP; }
    return NRF_OK;
}

// The weight gradient of a THIN layer (out <= 4 rows per pass: the alpha and rgb heads): dW[o][i] += sum_p g[p][o] x[p][i] is a pass over X with a handful of
// multipliers per row -- no matrix shape to speak of.  One thread per column of X, a slice of the points per workgroup (fp32 FMAs in point order), the slices' partial
// rows summed by k_tn_sum in slice order: deterministic.  part: [slices][OUT][n].
template <int OUT>
__global__ void __launch_bounds__(256) k_dw_thin(int64_t P, int64_t slice_pts, int n, const float *__restrict__ g, int ldg, const float *__restrict__ x, int ldx, float *__restrict__ part)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int64_t slice = blockIdx.y;
    const int64_t p0 = slice * slice_pts;
    int64_t p1 = p0 + slice_pts; if (p1 > P) p1 = P;
    float acc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; o++) acc[o] = 0.0f;
    if (i < n) {
#pragma unroll 4
        for (int64_t p = p0; p < p1; p++) {
            const float xv = x[p * ldx + i];
#pragma unroll
            for (int o = 0; o < OUT; o++) acc[o] = fmaf(g[p * ldg + o], xv, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < OUT; o++) part[((size_t)slice * OUT + o) * n + i] = acc[o];
    }
}

// dw [out][in] (columns col0 .. col0 + x.n) += g^T x for a layer of fewer than 32 rows, four rows per pass over X
int gemm_tn_thin(int64_t P, Seg g, Seg x, int out, int in, int col0, float *dw, hipStream_t st)
{
    if (P <= 0 || x.n <= 0 || out <= 0) return NRF_OK;
    int64_t slices = P / 1024; if (slices > 512) slices = 512; if (slices < 1) slices = 1;
    const int64_t slice_pts = (P + slices - 1) / slices;
    slices = (P + slice_pts - 1) / slice_pts;
    float *part = nullptr;
    if (scratch_take(reinterpret_cast<void **>(&part), (size_t)slices * 4 * x.n * sizeof(float), st) != hipSuccess) { set_error("gemm_tn_thin: scratch allocation failed"); return NRF_ERR_HIP; }
    gb_poison(part, (size_t)slices * 4 * x.n * sizeof(float), st);
    const dim3 grid((unsigned)ceil_div((int64_t)x.n, (int64_t)256), (unsigned)slices);
    for (int o0 = 0; o0 < out; o0 += 4) {
        const int oc = out - o0 < 4 ? out - o0 : 4;
        const float *gp = g.p + g.off + o0, *xp = x.p + x.off;
        if (oc == 1) hipLaunchKernelGGL(k_dw_thin<1>, grid, dim3(256), 0, st, P, slice_pts, x.n, gp, g.stride, xp, x.stride, part);
        else if (oc == 2) hipLaunchKernelGGL(k_dw_thin<2>, grid, dim3(256), 0, st, P, slice_pts, x.n, gp, g.stride, xp, x.stride, part);
        else if (oc == 3) hipLaunchKernelGGL(k_dw_thin<3>, grid, dim3(256), 0, st, P, slice_pts, x.n, gp, g.stride, xp, x.stride, part);
        else hipLaunchKernelGGL(k_dw_thin<4>, grid, dim3(256), 0, st, P, slice_pts, x.n, gp, g.stride, xp, x.stride, part);
        hipLaunchKernelGGL(k_tn_sum, dim3((unsigned)ceil_div((int64_t)oc * x.n, (int64_t)256)), dim3(256), 0, st, (int)slices, oc, x.n, 0, in, col0, 0, 0, 0, (const float *)part,
                           dw + (size_t)o0 * in);
    }
    const hipError_t le = hipGetLastError();
    (void)scratch_give(part, st);
    if (le != hipSuccess) { set_error("gemm_tn_thin: launch failed: %s", hipGetErrorString(le)); return NRF_ERR_HIP; }
    return NRF_OK;
}

// dw [out][in] (columns col0 .. col0 + x.n) += g^T x over P points
int gemm_tn_bf16x3(int64_t P, Seg g, Seg x, int out, int in, int col0, float *dw, hipStream_t st, const float *xscale, int xscale_ld)
{
    return gemm_tn_bf16x3_2(P, g, x, col0, Seg{nullptr, 0, 0, 0}, 0, 0, 0, out, in, dw, st, nullptr, xscale, xscale_ld);
}

// Arithmetic of the training paths' forward / back-propagation products.  -1 (the default, NRF_TRAIN_GEMM=auto): by network family -- f16x3 for the classic NeRF and the
// LeRF head (whose 256-wide layers ARE the training step; their reference-autograd goldens hold in it), fp32 products for NeRFSmall's fp32 backward (the hash path's
// parity chain: 4 096 points x 64-wide layers, where one ReLU decided the other way by a last-bit difference already shows against the golden; its fast chain is the fused
// fp16 backward, not these products).  0 / 1 / 2 (NRF_TRAIN_GEMM=f32 | bf16x3 | f16x3, nrf_set_train_gemm): every family as said.
//   0  fp32 products -- rocBLAS sgemm / mlp.hip's FMA kernels
//   1  bf16x3: hi + lo bf16, 16 significant bits; a classic step's weight gradients end 2.5e-4 (norm-wise) from the fp32 chain's
//   2  f16x3: hi + lo fp16 of power-of-two scaled operands, 22 significant bits: one product is closer to the float64 product than sgemm's (5e-7 vs 8e-7 of the largest
//      entry); a classic step's weight gradients end 6e-5 from the rocBLAS chain's, where rocBLAS and the FMA kernels are 2e-5 apart (profiles/round6/r6o_*)
#ifndef NRF_TRAIN_GEMM_DEFAULT
#define NRF_TRAIN_GEMM_DEFAULT -1
#endif
static std::atomic<int> g_train_gemm{-2};
int train_gemm_mode()
{
    int m = g_train_gemm.load(std::memory_order_relaxed);
    if (m >= -1) return m;
    m = NRF_TRAIN_GEMM_DEFAULT;
    if (const char *e = getenv("NRF_TRAIN_GEMM")) {
        if (!strcmp(e, "bf16x3") || !strcmp(e, "1")) m = 1;
        else if (!strcmp(e, "f16x3") || !strcmp(e, "2")) m = 2;
        else if (!strcmp(e, "f32") || !strcmp(e, "0")) m = 0;
        else if (!strcmp(e, "auto") || !strcmp(e, "-1")) m = -1;
    }
    g_train_gemm.store(m, std::memory_order_relaxed);
    return m;
}
void set_train_gemm_mode(int m) { g_train_gemm.store(m < -1 ? -1 : (m > 2 ? 2 : m), std::memory_order_relaxed); }
int train_gemm_for(const nrf_mlp *m)
{
    const int mode = train_gemm_mode();
    if (mode >= 0) return mode;
    return (m && m->family == MLP_SMALL) ? 0 : 2;
}


template <bool F16>
static int gemm_nt_launch(GemmNT &g, const float *B, int ldb, hipStream_t st)
{
    static const int force_wide = [] { const char *e = getenv("NRF_GEMM_WNW"); return e ? atoi(e) : 0; }();          // tuning: 2 / 4 forces the tile width
    // a narrow product (N <= 128: the sigma head, the gradient w.r.t. a 128-wide input) whose A operand fits the whole-row kernel takes that kernel's 256-wide tile with
    // the missing columns as zeros: its matrix work is wasted, its A stream is not (0.55 against 0.77 ms per 10^6 x 256 rows)
    static const bool narrow_rows = [] { const char *e = getenv("NRF_GEMM_NARROW_ROWS"); return !e || atoi(e) != 0; }();
    const int ktot0 = g.k0 + g.k1;
    const bool rows_shape = g.va0 == 4 && (g.k1 == 0 || g.va1 == 4) && (g.k0 % GB_BK) == 0 && (g.k1 % GB_BK) == 0 && (ktot0 == 128 || ktot0 == 160 || ktot0 == 256);
    const bool wide = force_wide == 4 || (force_wide != 2 && (g.N > 128 || (narrow_rows && rows_shape && g.M >= 65536)));
    const int bn = wide ? 256 : 128;
    if (!(wide && g.N <= 256)) {
        if (g.bits_out) { set_error("gemm_nt_split: mask bits asked of a product that cannot write them (gemm_nt_bits_ok)"); return NRF_ERR_INVALID_ARG; }
        g.bits_in = nullptr;                          // (the float mask given beside them applies)
    } else if (g.bits_in) g.mask = nullptr;
    const int64_t blocks = ceil_div(g.M, GB_BM) * ceil_div((int64_t)g.N, (int64_t)bn);
    if (blocks > 0x7fffffff) { set_error("gemm_nt_split: too many tiles"); return NRF_ERR_INVALID_ARG; }
    constexpr int LDS2 = 2 * GbCfg<2>::STAGE + GB_BM * 4, LDS4 = GB_ROWS_LDS;          // two stages + the rows' inverse scales; the 256-wide tile: its C block staged through LDS
    static PerDeviceOnce attr_set;
    if (attr_set.needed()) {
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<2, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<4, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_rows<4, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, GB_ROWS_LDS));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_rows<5, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, GB_ROWS_LDS));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_rows<8, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, GB_ROWS_LDS));
        attr_set.done();
    }
    // B: its largest entry (F16), then its split image -- two small launches over a cache-resident matrix, in stream order before the product
    const int T = (g.k0 + GB_BK - 1) / GB_BK + (g.k1 + GB_BK - 1) / GB_BK;
    g.npad = (int)ceil_div((int64_t)g.N, (int64_t)bn) * bn;
    g.bimg_half = (int64_t)T * 2 * g.npad * 32;
    unsigned char *ws = nullptr;
    if (scratch_take(reinterpret_cast<void **>(&ws), (size_t)(2 * g.bimg_half + 16), st) != hipSuccess) { set_error("gemm_nt_split: scratch allocation failed"); return NRF_ERR_HIP; }
    gb_poison(ws, (size_t)(2 * g.bimg_half + 16), st);
    uint32_t *bmax = reinterpret_cast<uint32_t *>(ws + 2 * g.bimg_half);
    if (F16) {
        (void)hipMemsetAsync(bmax, 0, sizeof(uint32_t), st);
        hipLaunchKernelGGL(k_gb_absmax, dim3((unsigned)(g.N < 256 ? (g.N + 3) / 4 : 64)), dim3(256), 0, st, g.N, g.k0 + g.k1, B, ldb, bmax);
    }
    hipLaunchKernelGGL((k_gb_split_b<F16>), dim3((unsigned)ceil_div((int64_t)T * 2 * g.npad, (int64_t)256)), dim3(256), 0, st, g.N, g.npad, g.k0, g.k1, B, ldb, (const uint32_t *)bmax, ws,
                       g.bimg_half);
    g.bimg = ws; g.bmax = bmax;
    static const bool no_rows = [] { const char *e = getenv("NRF_GEMM_ROWS"); return e && atoi(e) == 0; }();
    const int ktot = g.k0 + g.k1;
    const bool rows_ok = wide && !no_rows && g.va0 == 4 && (g.k1 == 0 || g.va1 == 4) && (g.k0 % GB_BK) == 0 && (g.k1 % GB_BK) == 0 && (ktot == 128 || ktot == 160 || ktot == 256);
    if (rows_ok) {
        int cus = 256;
        { int dev = 0, n = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 1) cus = n; }
        static const int per_cu = [] { const char *e = getenv("NRF_GEMM_ROWS_PERSIST"); return e ? atoi(e) : 1; }();          // 0: one workgroup per output tile (A/B)
        const unsigned pgrid = (unsigned)(per_cu > 0 && blocks > (int64_t)cus * per_cu ? (int64_t)cus * per_cu : blocks);          // persistent: one workgroup per CU (LDS: 133 KB)
        if (ktot == 128) hipLaunchKernelGGL((k_gemm_nt_rows<4, F16>), dim3(pgrid), dim3(512), GB_ROWS_LDS, st, g);
        else if (ktot == 160) hipLaunchKernelGGL((k_gemm_nt_rows<5, F16>), dim3(pgrid), dim3(512), GB_ROWS_LDS, st, g);
        else hipLaunchKernelGGL((k_gemm_nt_rows<8, F16>), dim3(pgrid), dim3(512), GB_ROWS_LDS, st, g);
    } else if (wide) hipLaunchKernelGGL((k_gemm_nt<4, F16>), dim3((unsigned)blocks), dim3(512), LDS4, st, g);
    else hipLaunchKernelGGL((k_gemm_nt<2, F16>), dim3((unsigned)blocks), dim3(256), LDS2, st, g);
    const hipError_t le = hipGetLastError();
    (void)scratch_give(ws, st);
    if (le != hipSuccess) { set_error("gemm_nt_split: launch failed: %s", hipGetErrorString(le)); return NRF_ERR_HIP; }
    return NRF_OK;
}

// whether a product of N output columns runs on the 256-wide tiles with ONE column block -- the launches that can write / read ReLU masks as bits (GemmNT::bits_out / bits_in)
bool gemm_nt_bits_ok(int64_t M, int N)
{
    static const int force_wide = [] { const char *e = getenv("NRF_GEMM_WNW"); return e ? atoi(e) : 0; }();
    static const bool on = [] { const char *e = getenv("NRF_GEMM_MASK_BITS"); return !e || atoi(e) != 0; }();          // 0: float masks everywhere (A/B)
    return on && force_wide != 2 && N > 128 && N <= 256 && (N & 3) == 0 && M > 0;
}

// C = cat[a, b] . B^T (+ bias)(ReLU)(mask): B [N][ldb] holds the columns of segment a first, then segment b's.  arithmetic: 1 = bf16x3, 2 = f16x3 (scaled)
int gemm_nt_split(int arithmetic, int64_t M, int N, Seg a, Seg b, const float *B, int ldb, float *c, int ldc, const float *bias, int relu, const float *mask, int mask_ld,
                  hipStream_t st, const float *add, int add_ld, const float *r1s, int r1s_ld, const float *r1w, uint64_t *bits_out, int bits_out_ld, const uint64_t *bits_in,
                  int bits_in_ld)
{
    if (M <= 0 || N <= 0) return NRF_OK;
    GemmNT g{};
    g.a0 = a.p ? a.p + a.off : nullptr; g.lda0 = a.stride; g.k0 = a.p ? a.n : 0;
    g.a1 = (b.p && b.n > 0) ? b.p + b.off : nullptr; g.lda1 = b.stride; g.k1 = g.a1 ? b.n : 0;
    if (g.k0 == 0 && g.k1 > 0) { g.a0 = g.a1; g.lda0 = g.lda1; g.k0 = g.k1; g.a1 = nullptr; g.k1 = 0; }
    g.c = c; g.ldc = ldc; g.M = M; g.N = N; g.bias = bias; g.relu = relu; g.mask = mask; g.mask_ld = mask_ld; g.add = add; g.add_ld = add_ld; g.r1s = r1w ? r1s : nullptr; g.r1s_ld = r1s_ld; g.r1w = r1w;
    g.bits_out = bits_out; g.bits_out_ld = bits_out_ld; g.bits_in = bits_in; g.bits_in_ld = bits_in_ld;
    g.va0 = g.a0 ? vec_class(g.a0, g.lda0, 0) : 1;
    g.va1 = g.a1 ? vec_class(g.a1, g.lda1, 0) : 1;
    return arithmetic == 2 ? gemm_nt_launch<true>(g, B, ldb, st) : gemm_nt_launch<false>(g, B, ldb, st);
}

}  // namespace nrf

#ifdef NRF_GB_TRACE
extern "C" NRF_API int nrf_dbg_gb_trace(unsigned long long *host_out, int reset)
{
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nrf::g_gb_trace), sizeof(unsigned long long) * 256 * 2 * 16) != hipSuccess) return NRF_ERR_HIP;
    if (reset) { static unsigned long long z[256 * 2 * 16]; if (hipMemcpyToSymbol(HIP_SYMBOL(nrf::g_gb_trace), z, sizeof(z)) != hipSuccess) return NRF_ERR_HIP; }
    return NRF_OK;
}
#endif

// -1: by network family (the default); 0: the training paths' forward / back-propagation products run as fp32 products (rocBLAS sgemm or mlp.hip's kernels); 1: as bf16x3,
// 2: as f16x3 (scaled) split-precision matrix-core GEMMs (this file)
extern "C" NRF_API int nrf_get_train_gemm(void) { return nrf::train_gemm_mode(); }
extern "C" NRF_API int nrf_set_train_gemm(int mode) { nrf::set_train_gemm_mode(mode); return NRF_OK; }

// C [M x N] (ldc) = A [M x K] (lda) . B [N x K]^T (ldb) (+ bias [N]) (ReLU): the split-precision products as stand-alone entries (tests, tools/scratch/gemm_probe.py)
extern "C" NRF_API int nrf_gemm_nt_bf16x3(const float *d_a, int lda, int64_t m, int k, const float *d_b, int ldb, int n, float *d_c, int ldc, const float *d_bias, int relu, void *stream)
{
    NRF_CHECK_ARG(d_a && d_b && d_c && m >= 0 && n >= 1 && k >= 1 && lda >= k && ldb >= k && ldc >= n, "nrf_gemm_nt_bf16x3: bad argument");
    return nrf::gemm_nt_split(1, m, n, nrf::Seg{d_a, lda, 0, k}, nrf::Seg{nullptr, 0, 0, 0}, d_b, ldb, d_c, ldc, d_bias, relu, nullptr, 0, nrf::as_stream(stream));
}
// dW [out x in] (ld in; columns col0 .. col0 + n) += G [p x out]^T (ldg) . X [p x n] (ldx): the weight-gradient product, bf16x3 arithmetic, deterministic
extern "C" NRF_API int nrf_gemm_tn_bf16x3(const float *d_g, int ldg, int out, const float *d_x, int ldx, int n, int64_t p, float *d_dw, int in, int col0, void *stream)
{
    NRF_CHECK_ARG(d_g && d_x && d_dw && p >= 0 && out >= 1 && n >= 1 && ldg >= out && ldx >= n && col0 >= 0 && in >= col0 + n, "nrf_gemm_tn_bf16x3: bad argument");
    return nrf::gemm_tn_bf16x3(p, nrf::Seg{d_g, ldg, 0, out}, nrf::Seg{d_x, ldx, 0, n}, out, in, col0, d_dw, nrf::as_stream(stream));
}
// The weight (and, with d_db, bias) gradient of one layer as the training paths' split-precision modes compute it: dW [out x in] (columns col0 .. col0 + n) += G^T X,
// db [out] += the column sums of G -- gemm_tn_bf16x3_2 for 32 rows and more (the bias sums out of the same pass over G), gemm_tn_thin for a head of fewer rows
extern "C" NRF_API int nrf_layer_grad_split(const float *d_g, int ldg, int out, const float *d_x, int ldx, int n, int64_t p, float *d_dw, int in, int col0, float *d_db, void *stream)
{
    NRF_CHECK_ARG(d_g && d_x && d_dw && p >= 0 && out >= 1 && n >= 1 && ldg >= out && ldx >= n && col0 >= 0 && in >= col0 + n, "nrf_layer_grad_split: bad argument");
    const nrf::Seg g{d_g, ldg, 0, out}, x{d_x, ldx, 0, n};
    hipStream_t st = nrf::as_stream(stream);
    if (out >= 32) return nrf::gemm_tn_bf16x3_2(p, g, x, col0, nrf::Seg{nullptr, 0, 0, 0}, 0, 0, 0, out, in, d_dw, st, d_db);
    NRF_TRY(nrf::gemm_tn_thin(p, g, x, out, in, col0, d_dw, st));
    if (d_db && p > 0) return nrf::run_grad_b(p, g, out, d_db, st);          // (a thin head's bias gradient: the pass the training paths run)
    return NRF_OK;
}
extern "C" NRF_API int nrf_gemm_nt_f16x3(const float *d_a, int lda, int64_t m, int k, const float *d_b, int ldb, int n, float *d_c, int ldc, const float *d_bias, int relu, void *stream)
{
    NRF_CHECK_ARG(d_a && d_b && d_c && m >= 0 && n >= 1 && k >= 1 && lda >= k && ldb >= k && ldc >= n, "nrf_gemm_nt_f16x3: bad argument");
    return nrf::gemm_nt_split(2, m, n, nrf::Seg{d_a, lda, 0, k}, nrf::Seg{nullptr, 0, 0, 0}, d_b, ldb, d_c, ldc, d_bias, relu, nullptr, 0, nrf::as_stream(stream));
}
