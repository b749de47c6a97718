// mlp_lerf_mfma.hip -- the LeRF language head (LeRFImpl::forward, LeRF.cpp:28-111) FUSED with its render pass
// (LeRFRenderer::RawToLEOutputs / RenderCLIPEmbedding, LeRFRenderer.cpp:27-76, LeRFRenderer.h:45-54) on the gfx950 matrix cores.
//
// The reference materialises raw_le [N, S, 769] (3 KB per sample point: 500 GB per 800x800 frame at 64+128 samples) only to
// reduce it to one 768-vector per ray.  Here the 768-wide output never leaves the register file:
//
//   kernel A  nrf_lerf_sigma             sigma net only  (in 128 -> 256 -> 33):  sigma_le per point, for the compositing weights
//   kernel B  nrf_lerf_render_embedding  sigma net (for the 32 geo features) -> LE0 (cat[geo, in] 160 -> 256, ReLU) = a_s; the embedding layer LE1
//               (256 -> 768) is bias-free and LINEAR, and so is the ray's weighted sum, hence
//                   sum_s w_s normalize(W a_s) = W . sum_s (w_s / max(||W a_s||, 1e-8)) a_s        with ||W a||^2 = a^T (W^T W) a:
//               per SAMPLE only the 256 x 256 Gram product (8 tiles) is evaluated, the scaled a_s of the 32 points of a tile (= 32 consecutive
//               samples of ONE ray) are summed across lanes and added to asum[ray][256] with one 128-byte float atomic per 32 neurons;
//   kernel C  W is applied ONCE PER RAY to asum (384 matrix instructions per 32 rays instead of per 32 samples): out[ray][768].
//             out[ray] = normalize(...) is finished by the caller's L2-normalise.  320 matrix instructions per 32 sample points instead of 960 (704 with
//             the Gram norm but the layer still run per sample).
//
// Same transposed MFMA formulation and the same L2 -> LDS weight streaming as mlp_nerf_mfma.hip (8 waves x 32 points per workgroup,
// chunks of <= 2 neuron tiles x all k-steps, LDS-DMA through three buffers two chunks ahead); fp16 operands, fp32 accumulate.
// Built for the reference's LeRF shape (main.cpp:203-213): in 16 x 8 = 128, hidden 256, 2 + 2 layers, geo 32, embedding 768.
#include "mlp_lerf_net.h"

#include <cmath>
#include <thread>
#include <utility>
#include <vector>

namespace nrf {
namespace lerf {

// Chunk CI of the weight image -> LDS buffer `dst` by LDS-DMA (global_load_lds_dwordx4, one 1-KB fragment per wave-instruction, wave w takes fragments
// w, w + NW, ...): same scheme as mlp_nerf_mfma.hip -- three LDS buffers, requested two chunks ahead, no staging registers.
template <class N, int CI>
__device__ __forceinline__ void stage_dma(half8 *__restrict__ dst, const half8 *__restrict__ packed, int wave, int lane)
{
    constexpr int ci = CI % N::total_chunks();
    constexpr int nf = N::chunk_frags(ci);
    constexpr int base = N::chunk_off(ci);
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
__device__ __forceinline__ half8 tile_to_frag(const f32x16 &acc, int s)
{
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (_Float16)acc[8 * s + j];
    if (RELU) r = __builtin_elementwise_max(r, half8{0, 0, 0, 0, 0, 0, 0, 0});
    return r;
}

struct Ctx {
    half8 *wbuf;                       // [3][MAXF*64]
    const half8 *packed;
    int tid, lane, h, wave;
    int *cur;                          // LDS buffer (0..2) of the chunk being consumed; wave-uniform
};

// One chunk: fetch the following chunk, run this chunk's MFMAs out of LDS, hand each finished tile to `hook(tile, acc)`, publish the
// fetched chunk.  All indices are template constants.
// w / dma_dst are __restrict__ parameters so that, inlined, the LDS reads and the DMA's LDS write carry alias scopes: otherwise every LDS read issued
// while an LDS-DMA is pending waits for vmcnt(0) (see mlp_nerf_mfma.hip).
template <class N, int L, int C, int NN, int NC, class Hook>
__device__ __forceinline__ void chunk_body(const Ctx &cx, const half8 *__restrict__ w, half8 *__restrict__ dma_dst, const half8 (&bn)[NN], const half8 (&bc)[NC], Hook &hook)
{
    constexpr int KSN = N::ks_nat(L), KSC = N::ks_ch(L), KS = KSN + KSC;
    constexpr int CI = N::first_chunk(L) + C;
    constexpr int NT = N::chunk_tiles(L, C);
    constexpr int TILE0 = (L == 1) ? C : 2 * C;
    constexpr bool NATF = N::nat_first(L);
    static_assert(KSN <= NN && KSC <= NC, "operand fragment arrays too small");
    stage_dma<N, CI + 2>(dma_dst, cx.packed, cx.wave, cx.lane);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; t++) {
        f32x16 acc = zero;
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const half8 a = w[(t * KS + k) * 64 + cx.lane];
            half8 b;
            if (NATF) b = (k < KSN) ? bn[k < KSN ? k : 0] : bc[k >= KSN ? k - KSN : 0];
            else b = (k < KSC) ? bc[k < KSC ? k : 0] : bn[k >= KSC ? k - KSC : 0];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        hook(TILE0 + t, acc);
        // the tile is consumed HERE: without the fence the scheduler runs ahead with the next tiles' MFMAs and parks finished accumulators
        // (16 VGPRs each) until it gets round to their epilogues -- hundreds of spills on the 24-tile layers
        __builtin_amdgcn_sched_barrier(0);
    }
    // chunk CI + 1 (requested a whole chunk ago) must have landed; chunk CI + 2 -- and any float atomics the hook issued after it -- may stay in flight:
    // vector-memory operations retire in issue order, so "at most KEEP outstanding" leaves only operations younger than chunk CI + 1's loads
    constexpr int KEEP = N::chunk_frags((CI + 2) % N::total_chunks()) / NW;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}

template <class N, int L, int C, int NN, int NC, class Hook>
__device__ __forceinline__ void chunk(const Ctx &cx, const half8 (&bn)[NN], const half8 (&bc)[NC], Hook &hook)
{
    const int cur = *cx.cur;
    chunk_body<N, L, C>(cx, cx.wbuf + cur * (MAXF * 64), cx.wbuf + (cur == 0 ? 2 : cur - 1) * (MAXF * 64), bn, bc, hook);
    *cx.cur = cur == 2 ? 0 : cur + 1;
}

template <class N, int L, int NN, int NC, class Hook, int... Cs>
__device__ __forceinline__ void layer_seq(const Ctx &cx, const half8 (&bn)[NN], const half8 (&bc)[NC], Hook &hook, std::integer_sequence<int, Cs...>)
{
    (chunk<N, L, Cs>(cx, bn, bc, hook), ...);
}

template <class N, int L, int NN, int NC, class Hook>
__device__ __forceinline__ void layer(const Ctx &cx, const half8 (&bn)[NN], const half8 (&bc)[NC], Hook &hook)
{
    layer_seq<N, L>(cx, bn, bc, hook, std::make_integer_sequence<int, N::chunks(L)>{});
}

// tile -> the two operand fragments it contributes to the next layer
template <bool RELU, int NOUT, bool KEEP0 = false>
struct ConvHook {
    half8 (&bout)[NOUT];
    float row0;                                // KEEP0: row 0 of tile 0 as it came out of the accumulator (sigma_le of the sigma net's last layer)
    __device__ __forceinline__ void operator()(int tile, const f32x16 &acc)
    {
        if (2 * tile + 1 < NOUT) { bout[2 * tile] = tile_to_frag<RELU>(acc, 0); bout[2 * tile + 1] = tile_to_frag<RELU>(acc, 1); }
        if (KEEP0 && tile == 0) row0 = acc[0];
    }
};

// a . (G a): tile t of G a against the operand fragments 2t, 2t+1 of a -- registers 8s..8s+7 of a D tile and the elements of fragment 2t+s are the
// same neurons on the same lane (that identity is what the whole transposed formulation rests on)
struct DotHook {
    const half8 (&a)[16];
    float ss = 0.0f;
    __device__ __forceinline__ void operator()(int tile, const f32x16 &acc)
    {
#pragma unroll
        for (int i = 0; i < 16; i++) ss = __builtin_fmaf(acc[i], (float)a[2 * tile + (i >> 3)][i & 7], ss);
        // pin the partial sum here: its only real use is after the 24th tile, and the IR-level code sinking would otherwise move all 384
        // FMAs down there -- keeping every finished accumulator alive (12 tiles in registers, 12 spilled)
        asm volatile("" : "+v"(ss));
    }
};

// Scale the point's column by f and sum the 32 points of the tile (the lanes of one half) with a reduce-scatter: at the step with
// partner lane ^ m a lane keeps the half of its registers selected by its own bit m and adds the partner's copy of that half, so after
// m = 16, 8, 4, 2 a lane holds ONE register, index r >> 1, summed over 16 lanes; the last exchange (m = 1) completes it.  62 VALU per
// tile instead of 160 for sixteen full butterflies, and the sums land one per lane -- the shape the 128-byte atomic wants.
struct ReduceHook {
    float f;
    float *out_row;        // nullptr: this wave's points lie beyond the batch
    int r, h;
    template <int NKEEP>      // partner = lane ^ (2 * NKEEP): keep NKEEP of 2 * NKEEP registers
    __device__ __forceinline__ void step(float (&v)[16]) const
    {
        const bool up = (r & (2 * NKEEP)) != 0;
#pragma unroll
        for (int i = 0; i < NKEEP; i++) {
            const float keep = up ? v[i + NKEEP] : v[i], send = up ? v[i] : v[i + NKEEP];
            v[i] = keep + __shfl_xor(send, 2 * NKEEP);
        }
    }
    __device__ __forceinline__ void operator()(int tile, const f32x16 &acc)
    {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = acc[i] * f;
        step<8>(v); step<4>(v); step<2>(v); step<1>(v);
        v[0] += __shfl_xor(v[0], 1);
        // lane r (even) of half h owns accumulator register i = r >> 1 = row 16(i>>3) + 8((i&7)>>2) + 4h + (i&3) of the tile
        const int i = r >> 1;
        if (out_row && (r & 1) == 0) unsafeAtomicAdd(out_row + tile * 32 + 16 * (i >> 3) + 8 * ((i & 7) >> 2) + 4 * h + (i & 3), v[0]);
    }
};

template <int NL>
__global__ void __launch_bounds__(64 * NW)
k_lerf_mfma(int64_t npts, Args in, const half8 *__restrict__ packed)
{
    using N = Net<NL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *wbuf = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    stage_dma<N, 0>(wbuf, packed, wave, lane);
    stage_dma<N, 1>(wbuf + MAXF * 64, packed, wave, lane);
    __syncthreads();                               // vmcnt(0): chunks 0 and 1 are in place
    int cur = 0;
    const int64_t nblocks = (npts + NBLK - 1) / NBLK;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        Ctx cx{wbuf, packed, tid, lane, h, wave, &cur};
        const int64_t p0 = blk * NBLK + wave * 32;
        const int64_t q = p0 + r;
        const bool live = q < npts;
        const int64_t qc = live ? q : npts - 1;
        // input operand: element j of k-step s is x[q][16 s + 8 h + j].  Needed by layer 0 and again by layer 2 (cat[geo, in]): read twice
        // (512 B per point from L2) rather than kept in 32 VGPRs across the sigma net
        half8 none[1];
        auto load_x = [&](half8 (&xin)[8]) {
            if (in.x_lm) {
                const int64_t col = in.src ? (int64_t)in.src[qc] : qc;
#pragma unroll
                for (int s = 0; s < 8; s++) xin[s] = *reinterpret_cast<const half8 *>(in.x_lm + ((int64_t)(2 * s + h) * in.pstride + col) * 8);
                return;
            }
            const float *row = in.x + qc * in.x_stride;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const float4 lo = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h);
                const float4 hi = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h + 4);
                xin[s] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
            }
        };
        half8 ba[16], bb[4];
        ConvHook<true, 16> c0{ba, 0.0f};
        {
            half8 xin[8];
            load_x(xin);
            layer<N, 0>(cx, xin, none, c0);                   // sigma0: 128 -> 256, ReLU
        }
        ConvHook<false, 4, NL == 2> c1{bb, 0.0f};
        layer<N, 1>(cx, none, ba, c1);                        // sigma1: 256 -> (sigma, geo32), tiles -> bb[0..3]
        if constexpr (NL == 2) {
            if (h == 0 && live) {
                float sg = c1.row0;
                if (in.keep && !in.keep[q]) sg = 0.0f;        // raw_le[~keep, -1] = 0 (LeRFRenderer.cpp:22-23)
                in.sigma[q] = sg;
            }
        } else {
            ConvHook<true, 16> c2{ba, 0.0f};
            {
                half8 xin[8];
                load_x(xin);
                layer<N, 2>(cx, xin, bb, c2);                 // LE0: cat[geo, in] -> 256, ReLU
            }
            DotHook ssq{ba};
            layer<N, 3>(cx, none, ba, ssq);                   // ||LE1(a)||^2 = a . (W^T W) a
            // G rounded to fp16 is not guaranteed positive semi-definite: a tiny negative a^T G a is clamped to 0 (-> the 1e-8 floor of normalize, LeRF.cpp:106-107)
            const float tot = fmaxf(ssq.ss + __shfl_xor(ssq.ss, 32), 0.0f) * in.gram_scale;
            const float wgt = live ? in.weights[q] : 0.0f;
            // sum over the tile's 32 samples of (w_s / ||h_s||) a_s: fragments 2t, 2t+1 of a hold, on each lane, the neurons of D-tile t's 16 registers
            ReduceHook red{wgt / fmaxf(sqrtf(tot), 1e-8f), (p0 < npts) ? in.out + (p0 / in.s) * (int64_t)HID : nullptr, r, h};
#pragma unroll
            for (int t = 0; t < 8; t++) {
                f32x16 v;
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = (float)ba[2 * t + (i >> 3)][i & 7];
                red(t, v);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last two chunks' look-ahead requests are still in flight
}

// kernel C: out[ray][768] = W . asum[ray][256] -- the embedding layer applied once per ray.  One wave per 32 rays; the operand (16 k-steps, natural order) is read
// from asum in fp32 and rounded to fp16 once; the 384 weight fragments (384 KB, L2-resident) are read straight from global memory, coalesced 1 KB per wave
// instruction -- at 1/192 of the per-sample work this kernel is 2 % of the pass and needs no staging.
__global__ void __launch_bounds__(256)
k_lerf_embed(int64_t nrays, const float *__restrict__ asum, const half8 *__restrict__ packed, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t ray = tile * 32 + r;
    if (tile * 32 >= nrays) return;
    const int64_t rc = ray < nrays ? ray : nrays - 1;
    half8 b[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const float4 lo = *reinterpret_cast<const float4 *>(asum + rc * HID + 16 * s + 8 * h), hi = *reinterpret_cast<const float4 *>(asum + rc * HID + 16 * s + 8 * h + 4);
        b[s] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
    }
    const half8 *w = packed + (size_t)LE1_FRAG0 * 64 + lane;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 24; t++) {
        f32x16 acc = zero;
#pragma unroll
        for (int k = 0; k < 16; k++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(size_t)(t * 16 + k) * 64], b[k], acc, 0, 0, 0);
        if (ray < nrays) {
            float *o = out + ray * EMB + 32 * t + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; g++) *reinterpret_cast<float4 *>(o + 8 * g) = float4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};      // rows 8g + 4h + 0..3
        }
    }
}

// value of the weight that multiplies operand element (kstep, h, j) for output row `row` of kernel layer L (host packer and device packer: one statement of the layout)
__host__ __device__ inline float lerf_wval(const float *hp, const float *gram, int L, int row, int kstep, int h, int j)
{
    const size_t off0 = 0, off1 = off0 + (size_t)HID * IN, off2 = off1 + (size_t)(1 + GEO) * HID, off3 = off2 + (size_t)HID * (GEO + IN);
    const int chained = 32 * (kstep >> 1) + perm_row(kstep & 1, h, j);
    const int natural = 16 * kstep + 8 * h + j;
    if (L == 0) return hp[off0 + (size_t)row * IN + natural];
    if (L == 1) return row < 1 + GEO ? hp[off1 + (size_t)row * HID + chained] : 0.0f;
    if (L == 2) {
        if (kstep < 4) {                                   // the two sigma1 tiles: row 0 = sigma (no weight), rows 1..32 = geo
            return (chained >= 1 && chained <= GEO) ? hp[off2 + (size_t)row * (GEO + IN) + (chained - 1)] : 0.0f;
        }
        return hp[off2 + (size_t)row * (GEO + IN) + GEO + (16 * (kstep - 4) + 8 * h + j)];
    }
    if (L == 3) return gram[(size_t)row * HID + chained];
    return hp[off3 + (size_t)row * HID + natural];          // LE1 for kernel C: its operand comes from memory, natural order
}

// fragment f of the image -> (layer, tile, k-step): layers in order, tiles of a layer in order, k-steps of a tile in order
__host__ __device__ inline void lerf_frag_id(int f, int &L, int &tile, int &k)
{
    const int cnt[5] = {8 * 8, 2 * 16, 8 * 12, 8 * 16, 24 * 16}, ksl[5] = {8, 16, 12, 16, 16};
    L = 0;
    while (L < 4 && f >= cnt[L]) { f -= cnt[L]; L++; }
    tile = f / ksl[L]; k = f - tile * ksl[L];
}

// ---- the same image built ON THE DEVICE from the parameter blob (a training loop uploads parameters every step: the host packer's 3 ms were GPU idle time) ----
// G = W^T W of the embedding layer W [768][256], every entry with the host packer's four partial double sums in its order: the same bits
__global__ void __launch_bounds__(256) k_lerf_gram_f64(const float *__restrict__ w3, float *__restrict__ gram, uint32_t *__restrict__ gmax_bits)
{
    const int a = blockIdx.x, b = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int o = 0; o + 4 <= EMB; o += 4) {
        s0 += (double)w3[(size_t)o * HID + a] * (double)w3[(size_t)o * HID + b];
        s1 += (double)w3[(size_t)(o + 1) * HID + a] * (double)w3[(size_t)(o + 1) * HID + b];
        s2 += (double)w3[(size_t)(o + 2) * HID + a] * (double)w3[(size_t)(o + 2) * HID + b];
        s3 += (double)w3[(size_t)(o + 3) * HID + a] * (double)w3[(size_t)(o + 3) * HID + b];
    }
    const float g = (float)((s0 + s1) + (s2 + s3));
    gram[(size_t)a * HID + b] = g;
    gram[(size_t)HID * HID + (size_t)a * HID + b] = g;          // the unscaled matrix, kept for the training backward's Gram form (lerf_train.hip)
    float mx = fabsf(g);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(gmax_bits, __float_as_uint(mx));
}
static_assert(EMB % 4 == 0, "the Gram kernel's four partial sums cover EMB exactly, as the host loop's main part does");

// the fp16-range scale (an exact power of two) and the upper block triangle (G_TT, 2 G_TU, zeros below), in place
__global__ void k_lerf_gram_finish(float *__restrict__ gram, int e)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= HID * HID) return;
    const int a = idx / HID, b = idx - a * HID;
    float g = gram[idx];
    if (e != 0) g = ldexpf(g, -e);
    const int ta = a / 32, ub = b / 32;
    if (ub < ta) g = 0.0f;
    else if (ub > ta) g *= 2.0f;
    gram[idx] = g;
}

// one workgroup per fragment, one thread per element: the fp16 image and the split image's (hi, lo) fragment pair
__global__ void __launch_bounds__(512) k_lerf_fill(const float *__restrict__ hp, const float *__restrict__ gram, _Float16 *__restrict__ img, _Float16 *__restrict__ img2)
{
    const int f = blockIdx.x, e = threadIdx.x, lane = e >> 3, j = e & 7;
    int L, tile, k;
    lerf_frag_id(f, L, tile, k);
    const float v = lerf_wval(hp, gram, L, tile * 32 + (lane & 31), k, lane >> 5, j);
    const _Float16 hv = (_Float16)v;
    img[(size_t)f * 512 + e] = hv;
    img2[(size_t)(2 * f) * 512 + e] = hv;
    img2[(size_t)(2 * f + 1) * 512 + e] = (_Float16)(v - (float)hv);
}

}  // namespace lerf

using namespace lerf;

static bool lerf_mfma_supported(const nrf_mlp_small_desc &d)
{
    return d.input_ch == IN && d.num_layers == 2 && d.hidden_dim == HID && d.geo_feat_dim == GEO && d.hidden_dim_color == EMB;
}

int mlp_lerf_pack_f16(nrf_mlp *m, const std::vector<float> &hp)
{
    m->lerf_gram_current = false;          // (a host re-pack: the device-side W^T W, if any, belongs to older parameters)
    if (!lerf_mfma_supported(m->small)) return NRF_OK;
    std::vector<_Float16> img;
    img.reserve((size_t)IMAGE_FRAGS * 512);
    using N = Net<5>;
    // Gram matrix of the embedding layer W [768][256] (bias-free): G = W^T W, accumulated in double
    std::vector<float> gram((size_t)HID * HID);
    {
        const float *w3 = hp.data() + (size_t)HID * IN + (size_t)(1 + GEO) * HID + (size_t)HID * (GEO + IN);
        std::vector<double> wt((size_t)HID * EMB);                    // W^T [256][768]: contiguous dot products
        for (int o = 0; o < EMB; o++)
            for (int k = 0; k < HID; k++) wt[(size_t)k * EMB + o] = (double)w3[(size_t)o * HID + k];
        // 256 x 257 / 2 dot products of 768 terms in double: the whole of a per-step parameter upload's host time while one thread did it (a LeRF training step re-packs
        // every step: 21 ms of a 76 ms step, tools/scratch/lerf_train_phases.py).  Rows dealt out to up to 8 threads; four partial sums per dot product (same double
        // precision, another association: the entries are rounded to fp32 and then to fp16 anyway)
        auto rows = [&](int a0, int a1) {
            for (int a = a0; a < a1; a++)
                for (int b = a; b < HID; b++) {
                    const double *pa = wt.data() + (size_t)a * EMB, *pb = wt.data() + (size_t)b * EMB;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int o = 0;
                    for (; o + 4 <= EMB; o += 4) { s0 += pa[o] * pb[o]; s1 += pa[o + 1] * pb[o + 1]; s2 += pa[o + 2] * pb[o + 2]; s3 += pa[o + 3] * pb[o + 3]; }
                    for (; o < EMB; o++) s0 += pa[o] * pb[o];
                    gram[(size_t)a * HID + b] = gram[(size_t)b * HID + a] = (float)((s0 + s1) + (s2 + s3));
                }
        };
        const int nt = host_pack_threads();
        if (nt == 1) rows(0, HID);
        else {
            // row a costs HID - a dot products: cut at equal areas of the triangle
            std::vector<std::thread> th;
            int a0 = 0;
            for (int t = 0; t < nt; t++) {
                const double frac = (double)(t + 1) / nt;
                int a1 = t == nt - 1 ? HID : (int)(HID * (1.0 - sqrt(1.0 - frac)));
                if (a1 < a0) a1 = a0;
                th.emplace_back(rows, a0, a1);
                a0 = a1;
            }
            for (auto &x : th) x.join();
        }
    }
    // fp16 range: entries of G are sums of 768 products and pass 65504 for weights of moderate size; G is stored divided by a power of two that brings
    // max|G| to <= 1024 (exact scaling; the kernels multiply a^T G a back), so neither the entries nor the fp16 rounding of large ones can overflow
    {
        float gmax = 0.0f;
        for (float g : gram) gmax = fmaxf(gmax, fabsf(g));
        int e = 0;
        if (gmax > 1024.0f) (void)frexpf(gmax / 1024.0f, &e);
        m->lerf_gram_scale = ldexpf(1.0f, e);
        if (e != 0) for (float &g : gram) g = ldexpf(g, -e);
    }
    // ... and as its upper BLOCK triangle: a^T G a = sum_T a_T^T G_TT a_T + 2 sum_{T < U} a_T^T G_TU a_U over 32-neuron blocks (G is symmetric), so the image
    // holds G_TT as is, 2 G_TU (exact: a power of two, |2 G| <= 2048) for U > T and zeros below the diagonal.  A kernel that runs all sixteen k-steps of a neuron
    // tile gets the same sum (the zero blocks contribute nothing); the split-precision kernels start tile T at k-step 2 T (k-steps 2 U, 2 U + 1 are neuron block U
    // in the chained operand order): 72 instead of 128 k-steps per point tile, 80 instead of 128 streamed fragments.
    for (int a = 0; a < HID; a++)
        for (int b = 0; b < HID; b++) {
            const int ta = a / 32, ub = b / 32;
            if (ub < ta) gram[(size_t)a * HID + b] = 0.0f;
            else if (ub > ta) gram[(size_t)a * HID + b] *= 2.0f;
        }
    // every fragment's (layer, tile, k-step), then the image values ONCE (fp16 image and the split image's (hi, lo) fragments both come from them), dealt out to threads:
    // a per-step re-pack evaluated wval 1.1 M times on one thread
    struct FragId { int L, tile, k; };
    std::vector<FragId> frags;
    frags.reserve(IMAGE_FRAGS);
    for (int L = 0; L < 5; L++)
        for (int tile = 0; tile < N::tiles(L); tile++)
            for (int k = 0; k < N::ks(L); k++) frags.push_back(FragId{L, tile, k});
    if ((int)frags.size() != IMAGE_FRAGS) { set_error("internal: LeRF weight image has %zu fragments, expected %d", frags.size(), IMAGE_FRAGS); return NRF_ERR_INVALID_ARG; }
    img.resize((size_t)IMAGE_FRAGS * 512);
    std::vector<_Float16> img2((size_t)IMAGE_FRAGS * 1024);
    auto fill = [&](int f0, int f1) {
        for (int f = f0; f < f1; f++) {
            const FragId id = frags[(size_t)f];
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const float v = lerf_wval(hp.data(), gram.data(), id.L, id.tile * 32 + (lane & 31), id.k, lane >> 5, j);
                    const _Float16 hv = (_Float16)v;
                    const size_t e = (size_t)lane * 8 + j;
                    img[(size_t)f * 512 + e] = hv;
                    img2[(size_t)(2 * f) * 512 + e] = hv;                                  // split image: every fragment followed by the fragment of the residuals w - f16(w)
                    img2[(size_t)(2 * f + 1) * 512 + e] = (_Float16)(v - (float)hv);
                }
        }
    };
    {
        const int nt = host_pack_threads();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back(fill, IMAGE_FRAGS * t / nt, IMAGE_FRAGS * (t + 1) / nt);
        for (auto &x : th) x.join();
    }
    if (m->d_packed_f16 && m->packed_f16_bytes != img.size() * sizeof(_Float16)) { (void)hipFree(m->d_packed_f16); m->d_packed_f16 = nullptr; }
    m->packed_f16_bytes = img.size() * sizeof(_Float16);
    if (!m->d_packed_f16) NRF_HIP(hipMalloc(&m->d_packed_f16, m->packed_f16_bytes));          // a re-pack of the same shape writes in place (the caller has synchronised: mlp.hip)
    NRF_HIP(hipMemcpy(m->d_packed_f16, img.data(), m->packed_f16_bytes, hipMemcpyHostToDevice));
    if (m->d_packed_split && m->packed_split_bytes != img2.size() * sizeof(_Float16)) { (void)hipFree(m->d_packed_split); m->d_packed_split = nullptr; }
    m->packed_split_bytes = img2.size() * sizeof(_Float16);
    if (!m->d_packed_split) NRF_HIP(hipMalloc(&m->d_packed_split, m->packed_split_bytes));
    NRF_HIP(hipMemcpy(m->d_packed_split, img2.data(), m->packed_split_bytes, hipMemcpyHostToDevice));
    return NRF_OK;
}

// The images from m->d_params on the device, in `st`'s order; one 4-byte read-back (the Gram matrix's largest entry decides its power-of-two scale, which the
// launchers pass by value).  Needs the image buffers of a first host pack.
int mlp_lerf_pack_f16_device(nrf_mlp *m, hipStream_t st)
{
    if (!lerf_mfma_supported(m->small) || !m->d_packed_f16 || !m->d_packed_split || m->packed_f16_bytes != (size_t)IMAGE_FRAGS * 512 * sizeof(_Float16)) return NRF_ERR_UNSUPPORTED;
    if (!m->d_lerf_gram) NRF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_lerf_gram), (size_t)2 * HID * HID * sizeof(float) + 16));          // [packed form | W^T W itself | max bits]
    uint32_t *gmax = reinterpret_cast<uint32_t *>(m->d_lerf_gram + (size_t)2 * HID * HID);
    const float *w3 = m->d_params + (size_t)HID * IN + (size_t)(1 + GEO) * HID + (size_t)HID * (GEO + IN);
    NRF_HIP(hipMemsetAsync(gmax, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_lerf_gram_f64, dim3(HID), dim3(HID), 0, st, w3, m->d_lerf_gram, gmax);
    NRF_LAUNCH_CHECK();
    uint32_t bits = 0;
    NRF_HIP(hipMemcpyAsync(&bits, gmax, sizeof(bits), hipMemcpyDeviceToHost, st));
    NRF_HIP(hipStreamSynchronize(st));
    float gm; memcpy(&gm, &bits, 4);
    int e = 0;
    if (gm > 1024.0f) (void)frexpf(gm / 1024.0f, &e);
    m->lerf_gram_scale = ldexpf(1.0f, e);
    hipLaunchKernelGGL(k_lerf_gram_finish, dim3(HID * HID / 256), dim3(256), 0, st, m->d_lerf_gram, e);
    NRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lerf_fill, dim3(IMAGE_FRAGS), dim3(512), 0, st, (const float *)m->d_params, (const float *)m->d_lerf_gram, reinterpret_cast<_Float16 *>(m->d_packed_f16),
                       reinterpret_cast<_Float16 *>(m->d_packed_split));
    NRF_LAUNCH_CHECK();
    m->lerf_gram_current = true;
    return NRF_OK;
}

template <int NL>
static int launch_lerf(const nrf_mlp *m, const Args &a_in, int64_t p, hipStream_t st)
{
    Args a = a_in;
    a.gram_scale = m->lerf_gram_scale;
    const size_t lds = (size_t)3 * MAXF * 1024;
    const int64_t nblocks = ceil_div(p, NBLK);
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);       // persistent: one 8-wave workgroup per CU (216 VGPRs: two waves per SIMD)
    hipLaunchKernelGGL((k_lerf_mfma<NL>), dim3(grid), dim3(64 * NW), lds, st, p, a, reinterpret_cast<const lerf::half8 *>(m->d_packed_f16));
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// kernel B into a stream-ordered scratch asum [n][256] (zeroed), then kernel C
static int lerf_embedding_passes(const nrf_mlp *m, lerf::Args a, int64_t n, int s, float *d_out, hipStream_t st)
{
    float *asum = nullptr;
    const size_t bytes = (size_t)n * lerf::HID * sizeof(float);
    NRF_HIP(scratch_take(reinterpret_cast<void **>(&asum), bytes, st));
    int rc = hipMemsetAsync(asum, 0, bytes, st) == hipSuccess ? NRF_OK : NRF_ERR_HIP;
    if (rc != NRF_OK) set_error("hipMemsetAsync failed");
    else {
        ProfScope prof(NRF_PROF_MLP, st);
        a.out = asum;
        rc = launch_lerf<4>(m, a, n * (int64_t)s, st);
        if (rc == NRF_OK) {
            hipLaunchKernelGGL(lerf::k_lerf_embed, dim3((unsigned)ceil_div(ceil_div(n, (int64_t)32), (int64_t)4)), dim3(256), 0, st, n, (const float *)asum,
                               reinterpret_cast<const lerf::half8 *>(m->d_packed_f16), d_out);
            if (hipGetLastError() != hipSuccess) { set_error("k_lerf_embed launch failed"); rc = NRF_ERR_HIP; }
        }
    }
    (void)scratch_give(asum, st);
    return rc;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_lerf_mfma_available(const nrf_mlp *m) { return m && m->family == MLP_LERF && m->d_packed_f16 != nullptr; }

int nrf_lerf_set_precision(nrf_mlp *m, int precision)
{
    NRF_CHECK_ARG(m && m->family == MLP_LERF, "nrf_lerf_set_precision: not a LeRF handle");
    NRF_CHECK_ARG(precision == NRF_PREC_F16_MFMA || precision == NRF_PREC_F16_SPLIT, "nrf_lerf_set_precision: the fused passes run in NRF_PREC_F16_MFMA or NRF_PREC_F16_SPLIT (NRF_PREC_F32: nrf_mlp_forward + the stage functions)");
    if (precision == NRF_PREC_F16_SPLIT && !lerf_split_available(m)) { set_error("nrf_lerf_set_precision: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    m->lerf_precision = precision;
    return NRF_OK;
}

int nrf_lerf_sigma(const nrf_mlp *m, const float *d_x, const uint8_t *d_keep, int64_t p, float *d_sigma, void *stream)
{
    NRF_CHECK_ARG(m && d_x && d_sigma && p >= 0, "nrf_lerf_sigma: bad argument");
    if (!nrf_lerf_mfma_available(m)) { set_error("nrf_lerf_sigma: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG((reinterpret_cast<uintptr_t>(d_x) & 15) == 0, "nrf_lerf_sigma: feature rows must be 16-byte aligned");
    if (p == 0) return NRF_OK;
    lerf::Args a{d_x, lerf::IN, nullptr, 0, nullptr, d_keep, d_sigma, nullptr, 32};
    if (m->lerf_precision == NRF_PREC_F16_SPLIT) return lerf_split_sigma(m, a, p, as_stream(stream));
    ProfScope prof(NRF_PROF_MLP, as_stream(stream));
    return launch_lerf<2>(m, a, p, as_stream(stream));
}

int nrf_lerf_render_embedding(const nrf_mlp *m, const float *d_x, const float *d_weights, int64_t n, int s, float *d_out, void *stream)
{
    NRF_CHECK_ARG(m && d_x && d_weights && d_out && n >= 0 && s >= 1, "nrf_lerf_render_embedding: bad argument");
    if (!nrf_lerf_mfma_available(m)) { set_error("nrf_lerf_render_embedding: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG(s % 32 == 0, "nrf_lerf_render_embedding: samples per ray (%d) must be a multiple of 32 (a wave's 32-point tile lies inside one ray)", s);
    NRF_CHECK_ARG((reinterpret_cast<uintptr_t>(d_x) & 15) == 0, "nrf_lerf_render_embedding: feature rows must be 16-byte aligned");
    if (n == 0) return NRF_OK;
    lerf::Args a{d_x, lerf::IN, nullptr, 0, d_weights, nullptr, nullptr, nullptr, s};
    if (m->lerf_precision == NRF_PREC_F16_SPLIT) return lerf_split_embedding_passes(m, a, n, s, d_out, as_stream(stream));
    return lerf_embedding_passes(m, a, n, s, d_out, as_stream(stream));
}

// the same two passes reading level-major fp16 features (nrf_hash_encode_lm_f16 of a 16-level, 8-feature CuHashEmbedder): [16][p][8] halfs
int nrf_lerf_sigma_lm(const nrf_mlp *m, const void *d_feats_lm, const uint8_t *d_keep, int64_t p, float *d_sigma, void *stream)
{
    return nrf_lerf_sigma_lm_strided(m, d_feats_lm, p, d_keep, p, d_sigma, stream);
}

int nrf_lerf_sigma_lm_strided(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const uint8_t *d_keep, int64_t p, float *d_sigma, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_sigma && p >= 0 && pstride >= p, "nrf_lerf_sigma_lm: bad argument");
    if (!nrf_lerf_mfma_available(m)) { set_error("nrf_lerf_sigma_lm: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG((reinterpret_cast<uintptr_t>(d_feats_lm) & 15) == 0, "nrf_lerf_sigma_lm: features must be 16-byte aligned");
    if (p == 0) return NRF_OK;
    lerf::Args a{nullptr, 0, reinterpret_cast<const __half *>(d_feats_lm), pstride, nullptr, d_keep, d_sigma, nullptr, 32};
    if (m->lerf_precision == NRF_PREC_F16_SPLIT) return lerf_split_sigma(m, a, p, as_stream(stream));
    ProfScope prof(NRF_PROF_MLP, as_stream(stream));
    return launch_lerf<2>(m, a, p, as_stream(stream));
}

size_t nrf_lerf_geo_bytes(int64_t columns) { return columns > 0 ? (size_t)columns * (size_t)lerf::GEO_BYTES_PER_COLUMN : 0; }

int nrf_lerf_sigma_geo_lm_strided(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const uint8_t *d_keep, int64_t p, float *d_sigma, void *d_geo,
                                  int64_t geo_stride, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_sigma && d_geo && p >= 0 && pstride >= p && geo_stride >= p, "nrf_lerf_sigma_geo_lm_strided: bad argument");
    if (!nrf_lerf_mfma_available(m)) { set_error("nrf_lerf_sigma_geo_lm_strided: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    if (m->lerf_precision != NRF_PREC_F16_SPLIT) { set_error("nrf_lerf_sigma_geo_lm_strided: built for NRF_PREC_F16_SPLIT (nrf_lerf_set_precision); the fp16 passes re-evaluate the sigma net"); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG(((reinterpret_cast<uintptr_t>(d_feats_lm) | reinterpret_cast<uintptr_t>(d_geo)) & 15) == 0, "nrf_lerf_sigma_geo_lm_strided: features / geo must be 16-byte aligned");
    if (p == 0) return NRF_OK;
    lerf::Args a{nullptr, 0, reinterpret_cast<const __half *>(d_feats_lm), pstride, nullptr, d_keep, d_sigma, nullptr, 32};
    a.geo = d_geo; a.geo_stride = geo_stride;
    return lerf_split_sigma(m, a, p, as_stream(stream));
}

int nrf_lerf_render_embedding_lm_geo(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const int32_t *d_src, const void *d_geo, int64_t geo_stride,
                                     const float *d_weights, int64_t n, int s, float *d_out, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_geo && d_weights && d_out && n >= 0 && s >= 1 && pstride >= 1 && geo_stride >= 1, "nrf_lerf_render_embedding_lm_geo: bad argument");
    if (!nrf_lerf_mfma_available(m)) { set_error("nrf_lerf_render_embedding_lm_geo: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    if (m->lerf_precision != NRF_PREC_F16_SPLIT) { set_error("nrf_lerf_render_embedding_lm_geo: built for NRF_PREC_F16_SPLIT (nrf_lerf_set_precision)"); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG(s % 32 == 0, "nrf_lerf_render_embedding_lm_geo: samples per ray (%d) must be a multiple of 32 (a wave's 32-point tile lies inside one ray)", s);
    NRF_CHECK_ARG(((reinterpret_cast<uintptr_t>(d_feats_lm) | reinterpret_cast<uintptr_t>(d_geo)) & 15) == 0, "nrf_lerf_render_embedding_lm_geo: features / geo must be 16-byte aligned");
    if (n == 0) return NRF_OK;
    lerf::Args a{nullptr, 0, reinterpret_cast<const __half *>(d_feats_lm), pstride, d_weights, nullptr, nullptr, nullptr, s};
    a.src = d_src; a.geo = const_cast<void *>(d_geo); a.geo_stride = geo_stride;
    return lerf_split_embedding_passes(m, a, n, s, d_out, as_stream(stream));
}

int nrf_lerf_render_embedding_lm(const nrf_mlp *m, const void *d_feats_lm, const float *d_weights, int64_t n, int s, float *d_out, void *stream)
{
    return nrf_lerf_render_embedding_lm_gather(m, d_feats_lm, n * (int64_t)s, nullptr, d_weights, n, s, d_out, stream);
}

int nrf_lerf_render_embedding_lm_gather(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const int32_t *d_src, const float *d_weights, int64_t n, int s,
                                        float *d_out, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_weights && d_out && n >= 0 && s >= 1, "nrf_lerf_render_embedding_lm: bad argument");
    NRF_CHECK_ARG(d_src ? pstride >= 1 : pstride >= n * (int64_t)s, "nrf_lerf_render_embedding_lm: column stride %lld too small", (long long)pstride);
    if (!nrf_lerf_mfma_available(m)) { set_error("nrf_lerf_render_embedding_lm: the matrix-core LeRF path is built for in 128 / hidden 256 / 2+2 layers / geo 32 / embedding 768"); return NRF_ERR_UNSUPPORTED; }
    NRF_CHECK_ARG(s % 32 == 0, "nrf_lerf_render_embedding_lm: samples per ray (%d) must be a multiple of 32 (a wave's 32-point tile lies inside one ray)", s);
    NRF_CHECK_ARG((reinterpret_cast<uintptr_t>(d_feats_lm) & 15) == 0, "nrf_lerf_render_embedding_lm: features must be 16-byte aligned");
    if (n == 0) return NRF_OK;
    lerf::Args a{nullptr, 0, reinterpret_cast<const __half *>(d_feats_lm), pstride, d_weights, nullptr, nullptr, nullptr, s};
    a.src = d_src;
    if (m->lerf_precision == NRF_PREC_F16_SPLIT) return lerf_split_embedding_passes(m, a, n, s, d_out, as_stream(stream));
    return lerf_embedding_passes(m, a, n, s, d_out, as_stream(stream));
}

}  // extern "C"
