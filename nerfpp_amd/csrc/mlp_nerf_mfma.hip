// mlp_nerf_mfma.hip -- NeRFImpl::forward (NeRF.cpp:41-126), the classic 8 x 256 MLP with view directions, on the gfx950
// matrix cores (NRF_PREC_F16_MFMA: fp16 operands, fp32 accumulate).
//
// Same transposed formulation as mlp_small_mfma.hip: H_{l+1}^T [neurons x points] = W_{l+1} . H_l^T with
// v_mfma_f32_32x32x16_f16, A = weights (32 neurons x 16 k), B = activations (16 k x 32 points).  A layer's 32 x 32 fp32
// output tiles become, after bias + ReLU + fp16 conversion, the B fragments of the next layer IN PLACE (registers 8s..8s+7
// of a tile = k-step s; the k permutation inside a k-step is folded into the weight image), so activations never leave
// the register file: one wavefront carries 32 points through all 11 GEMMs.
//
// What does not fit on chip is the weights: 1.16 MB of fp16 against 160 KB of LDS per CU.  The weight image
// is therefore cut into 40 CHUNKS (two 32-neuron tiles x all k-steps of a layer, <= 40 KB) laid out in consumption
// order and streamed through THREE LDS buffers by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write
// pass): a 512-thread workgroup (8 waves x 32 points = 256 points, one workgroup per CU, two waves per SIMD, persistent)
// requests chunk i+2 at the top of chunk i, runs the MFMAs of chunk i out of LDS, and ends the chunk with a counted
// s_waitcnt vmcnt (chunk i+1 landed, i+2 may stay in flight) and a raw s_barrier.  Every A fragment is one conflict-free
// ds_read_b128.  L2 -> LDS weight traffic is 1.16 MB per 256 points, ~4.6 KB per point.
//
// Measured on MI355X (800x800, 64+128; same box for every row): register-staged double buffer whose buffer_loads sat in
// waterfall loops (the descriptor had ended up in VGPRs) 187 ms -> descriptor rebuilt from the pinned scalar pointer 179 ->
// LDS-DMA, two buffers 169 -> three buffers, two chunks ahead 166.  Ablations of the final kernel: no barriers 162; the DMA
// reading the same 1 KB every time (L2 out of the picture) 160; a quarter of the DMA instructions 133; no DMA at all 122.
// So what the stream costs is neither its L2 bandwidth (4.5 TB/s aggregate) nor its latency but the LDS-side write of
// 1.16 MB per 256 points, ~35 clocks per 1-KB instruction during which the workgroup's fragment reads make no progress;
// the lever that remains is more points per weight pass, which at 64 VGPRs of activations per 32 points and buffer is a
// register-file problem (two tiles per wave = 256 VGPRs for the two activation buffers alone).
//
// Layer plan (D = 8, W = 256, skip 4, PE(10) positions = 63 -> 4 k-steps, PE(4) directions = 27 -> 2 k-steps):
//   0      : pts_linears_0   [nat 4]              -> 8 tiles  ReLU
//   1-4,6,7: pts_linears_i   [chained 16]         -> 8 tiles  ReLU
//   5      : pts_linears_5   [nat 4 | chained 16]    cat[input_pts, h] after layer 4 (NeRF.cpp:103-104)
//   8      : views_linears_0 o feature_linear [chained 16 | nat 2] -> 4 tiles ReLU, + alpha_linear (tile 4, row 0)
//            feature_linear has no activation (NeRF.cpp:110-113), so views_linears_0(cat[feature_linear(h), views]) is ONE affine map of
//            cat[h, views]: W_v[:, :256] . F (128 x 256, formed in double at pack time) on h, W_v[:, 256:] on the views, bias W_v[:, :256] . b_f + b_v.
//            90 matrix instructions per 32 points instead of the 216 of the two layers run one after the other (and 126 KB less weight stream).
//   9      : rgb_linear      [chained 8]          -> 1 tile (rows 0..2)         out = cat[rgb, alpha] (NeRF.cpp:119)
#include "mlp_nerf_net.h"

#include <functional>
#include <thread>

#include <utility>

namespace nrf {

// Chunk CI of the weight image -> LDS buffer `dst` by LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves one 1-KB fragment
// (64 lanes x 16 B, lane-linear on both sides -- exactly the fragment layout), wave w takes fragments w, w + NW, ...  No staging registers
// (the register-staged version carried 20 VGPRs per thread in a kernel at the 256-VGPR cap) and no ds_write pass; the data is in flight
// while the chunk's MFMAs run and is retired by the vmcnt(0) that __syncthreads() emits at the end of the chunk.
// The fragment's address is a pinned SGPR base + lane * 16: left to itself the compiler hoists 40 chunks x 5 lane addresses out of the
// persistent loop as 64-bit VGPR pairs.
template <int CI>
__device__ __forceinline__ void stage_dma(half8 *__restrict__ dst, const half8 *__restrict__ packed, int wave, int lane)
{
    constexpr int ci = CI % NerfNet::total_chunks();
    constexpr int nf = NerfNet::chunk_frags(ci);
    constexpr int base = NerfNet::chunk_off(ci);
#pragma unroll
    for (int q = 0; q < (nf + NW - 1) / NW; q++) {
        // SGPR base with the fragment's constant offset added on the scalar side, then made opaque (not hoistable out of the persistent loop), + lane * 16: the
        // saddr form of the DMA; with the offset added behind the opaque point the compiler forms a 64-bit per-lane address (two v_lshl_add_u64 per DMA)
        const half8 *pk = packed + (size_t)wave * 64;
        asm volatile("" : "+s"(pk));                     // not hoistable out of the persistent loop ...
        pk += (size_t)(base + q * NW) * 64;
        asm volatile("" : "+s"(pk));                     // ... and the offset added here, on the scalar side
        if (q * NW + wave < nf)                          // wave-uniform
            __builtin_amdgcn_global_load_lds(pk + lane, (__attribute__((address_space(3))) void *)(dst + (q * NW + wave) * 64), 16, 0, 0);
    }
}

template <bool RELU>
__device__ __forceinline__ half8 nerf_tile_to_frag(const f32x16 &acc, int s)
{
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (_Float16)acc[8 * s + j];
    // ReLU after the (monotonic) rounding: max(round(x), 0) == round(max(x, 0)); packed, 4 v_pk_max_f16 instead of 8 v_max_f32
    if (RELU) r = __builtin_elementwise_max(r, half8{0, 0, 0, 0, 0, 0, 0, 0});
    return r;
}

struct Ctx {
    half8 *wbuf;            // [2][MAXF*64]
    const float *bias_s;    // LDS
    const half8 *packed;    // the weight image (wave-uniform)
    int tid, lane, h, wave;
    int *cur;               // LDS buffer (0..2) holding the chunk being consumed; wave-uniform, advanced by every chunk
};

// One chunk (<= 2 neuron tiles of layer L): start fetching the following chunk, run this chunk's MFMAs out of LDS, turn the
// finished tiles into next-layer fragments, then publish the fetched chunk.  Every index is a template constant (the
// constexpr table functions of NerfNet must be evaluated at compile time: called with a loop variable they become runtime
// loops).  `last` receives the layer's final tile (alpha row / rgb rows are read from it).
template <int L, int C, bool RELU, int NN, int NC, int NOUT>
__device__ __forceinline__ void nerf_chunk_body(const Ctx &cx, const half8 *__restrict__ w, half8 *__restrict__ dma_dst, const float *__restrict__ bias_s,
                                                const half8 (&bn)[NPT][NN], const half8 (&bc)[NPT][NC], half8 (&bout)[NPT][NOUT], f32x16 (&last)[NPT])
{
    // w / dma_dst / bias_s are __restrict__ PARAMETERS on purpose: inlining turns that into alias-scope metadata on the LDS reads and on the DMA's LDS
    // write, which is what lets the compiler see that this chunk's reads do not touch the look-ahead's destination.  Without it every LDS read issued
    // while an LDS-DMA is pending is preceded by s_waitcnt vmcnt(0) and the look-ahead is drained at the top of the chunk.
    constexpr int KSN = NerfNet::ks_nat(L), KSC = NerfNet::ks_ch(L), KS = KSN + KSC;
    constexpr int CI = NerfNet::first_chunk(L) + C;
    constexpr int NT = NerfNet::chunk_tiles(L, C);
    constexpr int BOFF = NerfNet::bias_off(L);
    constexpr int NTILES = NerfNet::tiles(L);
    constexpr bool NATF = NerfNet::nat_first(L);
    static_assert(KSN <= NN && KSC <= NC, "operand fragment arrays too small");
    stage_dma<CI + 2>(dma_dst, cx.packed, cx.wave, cx.lane);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int tile = 2 * C + t;
        f32x16 acc[NPT];
        const float *bp = bias_s + BOFF + tile * 32 + 4 * cx.h;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 bv = *reinterpret_cast<const float4 *>(bp + 8 * g);
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) { acc[pt][4 * g + 0] = bv.x; acc[pt][4 * g + 1] = bv.y; acc[pt][4 * g + 2] = bv.z; acc[pt][4 * g + 3] = bv.w; }
        }
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const half8 a = w[(t * KS + k) * 64 + cx.lane];
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                half8 b;
                if (NATF) b = (k < KSN) ? bn[pt][k < KSN ? k : 0] : bc[pt][k >= KSN ? k - KSN : 0];
                else b = (k < KSC) ? bc[pt][k < KSC ? k : 0] : bn[pt][k >= KSC ? k - KSC : 0];
                acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[pt], 0, 0, 0);
            }
            // keep the weight-fragment reads at most a group of 4 ahead of their MFMAs: without the fence the scheduler hoists all 16-20 ds_read_b128 of a
            // tile (64-80 VGPRs).  (Fetching a whole group ahead of the previous group's MFMAs was measured: no gain, two waves per SIMD already cover it.)
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) {
            if (2 * tile + 1 < NOUT) {
                bout[pt][2 * tile] = nerf_tile_to_frag<RELU>(acc[pt], 0);
                bout[pt][2 * tile + 1] = nerf_tile_to_frag<RELU>(acc[pt], 1);
            }
            if (tile == NTILES - 1) last[pt] = acc[pt];
        }
    }
    // End of the chunk: chunk CI + 1 (requested one whole chunk ago) must have landed, chunk CI + 2 (requested at the top of this one) may stay in
    // flight -- a counted vmcnt and a RAW barrier (__syncthreads() would drain the DMA with vmcnt(0)).  KEEP = the fewest DMA instructions any wave
    // issued for chunk CI + 2.  The "memory" clobber keeps the compiler from moving LDS reads across it.
    constexpr int KEEP = NerfNet::chunk_frags((CI + 2) % NerfNet::total_chunks()) / NW;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}

template <int L, int C, bool RELU, int NN, int NC, int NOUT>
__device__ __forceinline__ void nerf_chunk(const Ctx &cx, const half8 (&bn)[NPT][NN], const half8 (&bc)[NPT][NC], half8 (&bout)[NPT][NOUT], f32x16 (&last)[NPT])
{
    // chunk CI + 2 -> the buffer chunk CI - 1 was consumed from (every wave is past the barrier that ended it)
    const int cur = *cx.cur;
    nerf_chunk_body<L, C, RELU>(cx, cx.wbuf + cur * (MAXF * 64), cx.wbuf + (cur == 0 ? 2 : cur - 1) * (MAXF * 64), cx.bias_s, bn, bc, bout, last);
    *cx.cur = cur == 2 ? 0 : cur + 1;
}

template <int L, bool RELU, int NN, int NC, int NOUT, int... Cs>
__device__ __forceinline__ void nerf_layer_seq(const Ctx &cx, const half8 (&bn)[NPT][NN], const half8 (&bc)[NPT][NC], half8 (&bout)[NPT][NOUT], f32x16 (&last)[NPT],
                                               std::integer_sequence<int, Cs...>)
{
    (nerf_chunk<L, Cs, RELU>(cx, bn, bc, bout, last), ...);
}

template <int L, bool RELU, int NN, int NC, int NOUT>
__device__ __forceinline__ void nerf_layer(const Ctx &cx, const half8 (&bn)[NPT][NN], const half8 (&bc)[NPT][NC], half8 (&bout)[NPT][NOUT], f32x16 (&last)[NPT])
{
    nerf_layer_seq<L, RELU>(cx, bn, bc, bout, last, std::make_integer_sequence<int, NerfNet::chunks(L)>{});
}

template <bool FUSED>
__global__ void __launch_bounds__(64 * NW)
k_mlp_nerf_mfma(int64_t npts, NerfInput in, const half8 *__restrict__ packed, const float *__restrict__ biases,
                float *__restrict__ out, int out_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The biases live in their OWN LDS object: the compiler makes every LDS read that may alias a pending LDS-DMA wait for vmcnt(0), which would drain the
    // look-ahead at the first bias read of a chunk; reads of a distinct object are provably clear of the DMA destinations in `smem`.
    __shared__ __attribute__((aligned(16))) float bias_s[NBIAS];
    half8 *wbuf = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < NBIAS; i += 64 * NW) bias_s[i] = biases[i];
    // three LDS buffers, weights requested TWO chunks ahead of their use: a chunk's MFMAs take 1-2 us, an L2 round trip under this load about as long
    stage_dma<0>(wbuf, packed, wave, lane);
    stage_dma<1>(wbuf + MAXF * 64, packed, wave, lane);
    __syncthreads();                               // vmcnt(0): chunks 0 and 1 are in place
    int cur = 0;
    const int64_t nblocks = (npts + NBLK - 1) / NBLK;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        Ctx cx{wbuf, bias_s, packed, tid, lane, h, wave, &cur};
        // point index of this lane's column in tile pt (recomputed where needed: nothing per-point stays live across the network)
        auto point_of = [&](int pt) -> int64_t { return blk * NBLK + (wave * NPT + pt) * 32 + r; };
        auto clamped = [&](int pt) -> int64_t { const int64_t q = point_of(pt); return q < npts ? q : npts - 1; };
        // natural-order operand fragments: element j of k-step s is input 16s + 8h + j.  They are (re)loaded right where a
        // layer consumes them (L0, L5: positions; L9: directions) instead of being kept live across the whole network.
        auto load_pe = [&](half8 (&pe)[NPT][4]) {
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                const int64_t q = clamped(pt);
                float px[3] = {0.0f, 0.0f, 0.0f};
                const float *row = nullptr;
                if constexpr (FUSED) {
                    if (in.x) {        // explicit sample points [p,3] (stochastic branches: scattered / preconditioned points)
                        px[0] = in.x[q * 3]; px[1] = in.x[q * 3 + 1]; px[2] = in.x[q * 3 + 2];
                    } else {
                        const float *rp = in.rays + (int64_t)((uint32_t)q / (uint32_t)in.s) * in.ray_stride;
                        const float zz = in.z[q];
                        px[0] = rp[0] + rp[3] * zz; px[1] = rp[1] + rp[4] * zz; px[2] = rp[2] + rp[5] * zz;
                    }
                } else row = in.x + q * in.x_stride;
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if constexpr (FUSED) {
                            // feature index k = 16s + 8h + j: [x(3) | per frequency f: sin(x 2^f)(3), cos(x 2^f)(3)].  Both lane halves'
                            // indices are COMPILE-TIME constants and h only selects between them: index arithmetic on h would be
                            // hoisted out of the persistent loop as ~90 loop-invariant VGPRs and spilled.
                            constexpr auto arg_axis = [](int k) { return k < 3 ? k : ((k - 3) % 6) % 3; };
                            constexpr auto arg_freq = [](int k) { return k < 3 ? 0 : (k - 3) / 6; };
                            constexpr auto kind = [](int k) { return k < 3 ? 0 : k >= 63 ? 3 : (((k - 3) % 6) < 3 ? 1 : 2); };   // 0 raw, 1 sin, 2 cos, 3 pad
                            const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
                            const float a0 = px[arg_axis(k0)] * __builtin_ldexpf(1.0f, arg_freq(k0));
                            const float a1 = px[arg_axis(k1 < 63 ? k1 : 0)] * __builtin_ldexpf(1.0f, arg_freq(k1 < 63 ? k1 : 0));
                            float sn, cs;
                            nrf_sincosf(h ? a1 : a0, &sn, &cs);
                            const int kd0 = kind(k0), kd1 = kind(k1);
                            const float v0 = kd0 == 0 ? px[arg_axis(k0)] : kd0 == 1 ? sn : kd0 == 2 ? cs : 0.0f;
                            const float v1 = kd1 == 0 ? px[arg_axis(k1 < 63 ? k1 : 0)] : kd1 == 1 ? sn : kd1 == 2 ? cs : 0.0f;
                            pe[pt][s][j] = (_Float16)(h ? v1 : v0);
                        } else pe[pt][s][j] = (_Float16)row[16 * s + 8 * h + j];  // index 63 is the first view feature: its weight column is zero
                    }
            }
        };
        half8 ba[NPT][16], bb[NPT][16], none[NPT][1];
        f32x16 last[NPT];
        half8 pe[NPT][4];                                      // needed twice (layer 0 and the skip layer 5): 16 registers, freed by the DMA staging
        load_pe(pe);
        nerf_layer<0, true>(cx, pe, none, ba, last);
        nerf_layer<1, true>(cx, none, ba, bb, last);
        nerf_layer<2, true>(cx, none, bb, ba, last);
        nerf_layer<3, true>(cx, none, ba, bb, last);
        nerf_layer<4, true>(cx, none, bb, ba, last);
        nerf_layer<5, true>(cx, pe, ba, bb, last);
        nerf_layer<6, true>(cx, none, bb, ba, last);
        nerf_layer<7, true>(cx, none, ba, bb, last);
        float alpha[NPT];
        {
            half8 vw[NPT][2];
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                const int64_t q = clamped(pt);
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    if constexpr (FUSED) vw[pt][s] = *reinterpret_cast<const half8 *>(in.dirs + (int64_t)((uint32_t)q / (uint32_t)in.s) * 32 + 16 * s + 8 * h);
                    else {
                        const float *row = in.x + q * in.x_stride;
#pragma unroll
                        for (int j = 0; j < 8; j++) { const int k = 16 * s + 8 * h + j; vw[pt][s][j] = (k < 27) ? (_Float16)row[63 + k] : (_Float16)0.0f; }
                    }
                }
            }
            nerf_layer<8, true>(cx, vw, bb, ba, last);          // views_linears_0 o feature_linear -> ba[..][0..7] (ReLU); tile 4 row 0 = alpha (from the accumulator)
        }
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) alpha[pt] = last[pt][0];
        nerf_layer<9, false>(cx, none, ba, bb, last);
        if (h == 0) {
#pragma unroll
            for (int pt = 0; pt < NPT; pt++) {
                const int64_t q = point_of(pt);
                if (q < npts) {
                    float *o = out + q * out_stride;
                    o[0] = last[pt][0]; o[1] = last[pt][1]; o[2] = last[pt][2]; o[3] = alpha[pt];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last two chunks' look-ahead requests are still in flight
}

// ---------------------------------------------------------------------------------------------------
// weight image
// ---------------------------------------------------------------------------------------------------
static bool nerf_mfma_supported(const nrf_mlp_nerf_desc &d)
{
    return d.depth == 8 && d.width == 256 && d.input_ch == 63 && d.input_ch_views == 27 && d.skip == 4 && d.use_viewdirs;
}

// blob offset (floats) of views_linears_0 in the built family (8 x 256, 63 inputs, skip 4): behind the eight pts_linears (NeRF.cpp:75-89 order)
size_t nerf_blob_offset_views() { return (size_t)63 * 256 + 256 + (size_t)6 * (256 * 256 + 256) + (size_t)(256 + 63) * 256 + 256; }

void host_parallel_for(int n, const std::function<void(int, int)> &range_fn)
{
    const int nt = n < 2 ? 1 : host_pack_threads();
    if (nt == 1) { range_fn(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(range_fn, (int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt));
    for (auto &x : th) x.join();
}

void nerf_merged_views_host(const float *wv, int wv_stride, const float *wf, const float *bf, const float *bv, int rows, int w, std::vector<float> &merged, std::vector<float> &merged_b)
{
    merged.assign((size_t)rows * w, 0.0f); merged_b.assign((size_t)rows, 0.0f);
    host_parallel_for(rows, [&](int r0, int r1) {
        std::vector<double> acc((size_t)w);
        for (int r = r0; r < r1; r++) {
            std::fill(acc.begin(), acc.end(), 0.0);
            double b = (double)bv[r];
            for (int f = 0; f < w; f++) {
                const double c = (double)wv[(size_t)r * wv_stride + f];
                b += c * (double)bf[f];
                const float *frow = wf + (size_t)f * w;
                for (int k = 0; k < w; k++) acc[k] += c * (double)frow[k];
            }
            for (int k = 0; k < w; k++) merged[(size_t)r * w + k] = (float)acc[k];
            merged_b[r] = (float)b;
        }
    });
}

// The fp16 image, the split image and the bias array of the classic network as host vectors, from the parameter blob `hp` and the merged views layer
// (views_linears_0 o feature_linear: `merged` [128][256], `merged_b` [128], nerf_merged_views_host) -- every entry a copy (hi / lo half) of one entry of those three
// arrays or zero, which is what lets mlp.hip decode the layout from probe blobs and rebuild the images on the device.  false: outside the built family.
bool nerf_f16_images_host(const nrf_mlp_nerf_desc &d, const float *hp, const float *merged, const float *merged_b, std::vector<_Float16> &img, std::vector<_Float16> &img2,
                          std::vector<float> &bias)
{
    if (!nerf_mfma_supported(d)) return false;
    const int W = 256, IN = 63, V = 27;
    // blob offsets (NeRF.cpp:75-89 order)
    std::vector<size_t> w_off(12), b_off(12);
    std::vector<int> in_dim(12), out_dim(12);
    size_t off = 0;
    for (int l = 0; l < 8; l++) {
        in_dim[l] = l == 0 ? IN : (l == 5 ? W + IN : W); out_dim[l] = W;
        w_off[l] = off; off += (size_t)in_dim[l] * W; b_off[l] = off; off += W;
    }
    const int VIEWS = 8, FEAT = 9, ALPHA = 10, RGB = 11;
    in_dim[VIEWS] = V + W; out_dim[VIEWS] = W / 2; w_off[VIEWS] = off; off += (size_t)(V + W) * (W / 2); b_off[VIEWS] = off; off += W / 2;
    in_dim[FEAT] = W; out_dim[FEAT] = W; w_off[FEAT] = off; off += (size_t)W * W; b_off[FEAT] = off; off += W;
    in_dim[ALPHA] = W; out_dim[ALPHA] = 1; w_off[ALPHA] = off; off += W; b_off[ALPHA] = off; off += 1;
    in_dim[RGB] = W / 2; out_dim[RGB] = 3; w_off[RGB] = off; off += (size_t)(W / 2) * 3; b_off[RGB] = off; off += 3;

    img.clear(); img2.clear();
    bias.assign(NBIAS, 0.0f);
    auto chained = [](int k, int h, int j) { return 32 * (k >> 1) + nerf_perm_row(k & 1, h, j); };
    auto natural = [](int k, int h, int j) { return 16 * k + 8 * h + j; };
    // value of the weight that multiplies operand element (kstep, h, j) for output row `row` of kernel-layer L (-> 0 if padding)
    auto wval = [&](int L, int row, int kstep, int h, int j) -> float {
        const int ksn = NerfNet::ks_nat(L), ksc = NerfNet::ks_ch(L);
        const bool nat = NerfNet::nat_first(L) ? (kstep < ksn) : (kstep >= ksc);
        const int kk = NerfNet::nat_first(L) ? (nat ? kstep : kstep - ksn) : (nat ? kstep - ksc : kstep);
        const int idx = nat ? natural(kk, h, j) : chained(kk, h, j);
        if (L < 8) {
            const float *w = hp + w_off[L];
            if (L == 0) return (idx < IN) ? w[(size_t)row * IN + idx] : 0.0f;
            if (L == 5) return nat ? ((idx < IN) ? w[(size_t)row * (W + IN) + idx] : 0.0f) : w[(size_t)row * (W + IN) + IN + idx];
            return w[(size_t)row * W + idx];
        }
        if (L == 8) {
            if (row < W / 2) return nat ? ((idx < V) ? hp[w_off[VIEWS] + (size_t)row * (V + W) + W + idx] : 0.0f) : merged[(size_t)row * W + idx];
            return (row == W / 2 && !nat) ? hp[w_off[ALPHA] + idx] : 0.0f;
        }
        return row < 3 ? hp[w_off[RGB] + (size_t)row * (W / 2) + idx] : 0.0f;
    };
    for (int L = 0; L < NerfNet::NLAYER; L++) {
        float *bp = bias.data() + NerfNet::bias_off(L);
        if (L < 8) for (int i = 0; i < W; i++) bp[i] = hp[b_off[L] + i];
        else if (L == 8) { for (int i = 0; i < W / 2; i++) bp[i] = merged_b[i]; bp[W / 2] = hp[b_off[ALPHA]]; }
        else for (int i = 0; i < 3; i++) bp[i] = hp[b_off[RGB] + i];
    }
    // every fragment's (layer, tile, k-step); the image values ONCE -- the fp16 image and the split image's (hi, lo) fragments both come from them -- on up to 8 threads
    // (a training loop re-packs every step: this function and mlp_nerf_pack_sigma_f32 were 25 ms of an 85 ms classic step on one thread)
    struct FragId { int L, tile, k; };
    std::vector<FragId> frags;
    for (int L = 0; L < NerfNet::NLAYER; L++)
        for (int tile = 0; tile < NerfNet::tiles(L); tile++)
            for (int k = 0; k < NerfNet::ks(L); k++) frags.push_back(FragId{L, tile, k});
    const int NF = (int)frags.size();
    if (NF != NerfNet::total_frags()) { set_error("internal: classic NeRF weight image has %d fragments, expected %d", NF, NerfNet::total_frags()); return false; }
    img.resize((size_t)NF * 512);
    img2.resize((size_t)NF * 1024);
    host_parallel_for(NF, [&](int f0, int f1) {
        for (int f = f0; f < f1; f++) {
            const FragId id = frags[(size_t)f];
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const float v = wval(id.L, id.tile * 32 + (lane & 31), id.k, lane >> 5, j);
                    const _Float16 hv = (_Float16)v;
                    const size_t e = (size_t)lane * 8 + j;
                    img[(size_t)f * 512 + e] = hv;
                    img2[(size_t)(2 * f) * 512 + e] = hv;                                  // NRF_PREC_F16_SPLIT image: every fragment followed by the fragment of the residuals w - f16(w)
                    img2[(size_t)(2 * f + 1) * 512 + e] = (_Float16)(v - (float)hv);
                }
        }
    });
    return true;
}

int mlp_nerf_pack_f16(nrf_mlp *m, const std::vector<float> &hp)
{
    const auto &d = m->nerf;
    if (!nerf_mfma_supported(d)) return NRF_OK;
    const int W = 256, V = 27;
    // views_linears_0 o feature_linear: merged[r][k] = sum_f W_v[r][f] F[f][k], merged_b[r] = sum_f W_v[r][f] b_f[f] + b_v[r]   (double accumulation)
    const size_t o_views = nerf_blob_offset_views(), o_feat = o_views + (size_t)(V + W) * (W / 2) + W / 2;
    std::vector<float> merged, merged_b;
    nerf_merged_views_host(hp.data() + o_views, V + W, hp.data() + o_feat, hp.data() + o_feat + (size_t)W * W, hp.data() + o_views + (size_t)(V + W) * (W / 2), W / 2, W, merged, merged_b);
    std::vector<_Float16> img, img2;
    std::vector<float> bias;
    if (!nerf_f16_images_host(d, hp.data(), merged.data(), merged_b.data(), img, img2, bias)) return NRF_ERR_INVALID_ARG;
    m->host_merged = merged; m->host_merged_b = merged_b;          // for mlp_nerf_pack_sigma_f32 of the same upload
    const size_t b1 = img.size() * sizeof(_Float16) + bias.size() * sizeof(float), b2 = img2.size() * sizeof(_Float16) + bias.size() * sizeof(float);
    if (m->d_packed_f16 && m->packed_f16_bytes != b1) { (void)hipFree(m->d_packed_f16); m->d_packed_f16 = nullptr; }        // a re-pack of the same shape writes in place (the caller has synchronised)
    m->packed_f16_bytes = b1;
    if (!m->d_packed_f16) NRF_HIP(hipMalloc(&m->d_packed_f16, b1));
    NRF_HIP(hipMemcpy(m->d_packed_f16, img.data(), img.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    NRF_HIP(hipMemcpy(static_cast<char *>(m->d_packed_f16) + img.size() * sizeof(_Float16), bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    if (m->d_packed_split && m->packed_split_bytes != b2) { (void)hipFree(m->d_packed_split); m->d_packed_split = nullptr; }
    m->packed_split_bytes = b2;
    if (!m->d_packed_split) NRF_HIP(hipMalloc(&m->d_packed_split, b2));
    NRF_HIP(hipMemcpy(m->d_packed_split, img2.data(), img2.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    NRF_HIP(hipMemcpy(static_cast<char *>(m->d_packed_split) + img2.size() * sizeof(_Float16), bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    return NRF_OK;
}

static int launch_nerf(const nrf_mlp *m, const NerfInput &in, bool fused, int64_t p, float *out, int os, hipStream_t st)
{
    if (!m->d_packed_f16) {
        set_error("NRF_PREC_F16_MFMA: this NeRF shape is outside the built matrix-core family (8 x 256, skip 4, PE(10)/PE(4), view directions); use NRF_PREC_F32");
        return NRF_ERR_UNSUPPORTED;
    }
    const size_t lds = (size_t)3 * MAXF * 1024;          // + the static bias array
    const int64_t nblocks = ceil_div(p, NBLK);
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);       // one persistent workgroup per CU
    static PerDeviceOnce attr_set;          // idempotent one-time setup per device (common.h)
    if (attr_set.needed()) {
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_nerf_mfma<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_nerf_mfma<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    const half8 *packed = reinterpret_cast<const half8 *>(m->d_packed_f16);
    const float *biases = reinterpret_cast<const float *>(static_cast<const char *>(m->d_packed_f16) + (size_t)NerfNet::total_frags() * 1024);
    if (fused) hipLaunchKernelGGL(k_mlp_nerf_mfma<true>, dim3(grid), dim3(64 * NW), lds, st, p, in, packed, biases, out, os);
    else hipLaunchKernelGGL(k_mlp_nerf_mfma<false>, dim3(grid), dim3(64 * NW), lds, st, p, in, packed, biases, out, os);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int mlp_nerf_forward_mfma(const nrf_mlp *m, const float *x, int xs, int64_t p, float *out, int os, hipStream_t st)
{
    NerfInput in{x, xs, nullptr, 0, nullptr, 1, nullptr};
    return launch_nerf(m, in, false, p, out, os, st);
}

int mlp_nerf_mfma_available(const nrf_mlp *m) { return m && m->family == MLP_NERF && m->d_packed_f16 != nullptr; }

// renderer fast path: points from (rays, z) -- or explicit `pts` [p,3] when not NULL --, PE in registers, per-ray fp16 direction encodings -> raw [p,4]
int mlp_nerf_forward_mfma_fused(const nrf_mlp *m, const float *pts, const float *rays, int ray_stride, const float *z, int s, const __half *dirs, int64_t p, float *out, hipStream_t st)
{
    ProfScope prof(NRF_PROF_MLP, st);
    NerfInput in{pts, 3, rays, ray_stride, z, s, dirs};
    return launch_nerf(m, in, true, p, out, 4, st);
}

// per-ray PE(4) of the view direction as fp16 rows [n, 32] (27 features, zero padded): the L9 operand of the fused path
__global__ void k_dirs_pe_f16(int64_t n, const float *__restrict__ rays, int stride, __half *__restrict__ out)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * 32) return;
    const int64_t i = gid >> 5;
    const int k = (int)(gid & 31);
    const float *dp = rays + i * stride + 8;
    float v = 0.0f;
    if (k < 3) v = dp[k];
    else if (k < 27) {
        const int f = (k - 3) / 6, q = (k - 3) - 6 * f;
        float sn, cs;
        nrf_sincosf(dp[q < 3 ? q : q - 3] * __builtin_ldexpf(1.0f, f), &sn, &cs);
        v = q < 3 ? sn : cs;
    }
    out[gid] = __float2half_rn(v);
}

int launch_dirs_pe_f16(const float *rays, int stride, int64_t n, __half *out, hipStream_t st)
{
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_dirs_pe_f16, dim3((unsigned)ceil_div(n * 32, 256)), dim3(256), 0, st, n, rays, stride, out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf
