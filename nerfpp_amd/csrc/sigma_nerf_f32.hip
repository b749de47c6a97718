// sigma_nerf_f32.hip -- the density branch of NeRFImpl::forward (NeRF.cpp:92-108: eight Linear(+bias)+ReLU layers of width 256 with the skip-concat
// h = cat[input_pts, h] after layer 4, then alpha_linear) in EXACT fp32 on the matrix cores, for the coarse pass of the classic renderer.
//
// Why.  With N_importance > 0 the coarse pass contributes only its compositing weights (NeRFRenderer.h:422-428), i.e. only sigma, and those weights choose the
// fine samples through searchsorted (Sampler.h:6-43) -- a discontinuous function.  In the split (hi + lo fp16) arithmetic of the timed mode 99.93 % of the pixel
// values came out within 1e-4 of the fp32 render and the rest did not: a moved sample.  "Certified sampling" (re-evaluating only the rays whose u values come
// within an error bound of a CDF edge) cannot close that gap: u_127 = 1.0 is compared with the CDF's top plateau, whose entries sit within 1e-7 of 1.0 on every
// ray that saturates, so whether cdf_i <= 1.0 holds depends on the last bit of the weight sum -- nearly every ray would be flagged
// (docs/history/profiles/round3/r3c_classic_cdf_stats.log).  So the coarse sigma is evaluated in the parity arithmetic itself: v_mfma_f32_32x32x2_f32 is, bit for bit, the
// ascending-k fmaf chain of NRF_PREC_F32 / the oracle (sigma_small_f32.hip).  The colour branch (feature_linear, views_linears, rgb_linear: 17 % of the MACs) is
// dead work for the coarse pass's own result -- but the fine pass re-evaluates the network at the S coarse depths unless the coarse pass leaves whole (rgb, sigma)
// rows to reuse.  So the kernel's tail runs it on the exact h8 in SPLIT precision on the fp16 matrix instructions (240 of them per 32 points next to 7 680 fp32
// ones): the renderer's raw-reuse path then evaluates the split-precision network on the N_importance new samples only (frame 645-690 -> 587 ms).
//
// Formulation (as sigma_small_f32.hip / sigma_lerf_f32.hip): layers transposed, H^T [256 x points] = W [256 x K] . X^T, A = 32 neurons x 2 k, B = 2 k x 32 points,
// K / 2 ascending k-steps through ONE accumulator per neuron tile (a dependent chain of this instruction issues back to back, MI355X_MICROARCH.md).  Row i of a
// neuron tile carries neuron 2(4(i/8) + i%4) + (i/4)%2, so register q of lane half hh of a finished tile is neuron 2q + hh: after bias and ReLU it IS the B operand
// of k-step 16 tile + q of the next layer.  Padding: K = 63 runs as 64 with a zero product at the end, and the skip layer's cat[input(63), h(256)] runs as
// [input(63), 0, h(256)] -- fmaf(0, 0, acc) returns acc exactly (an accumulator that started at +0 is never -0), so both chains equal the oracle's.
// alpha_linear (256 -> 1) is a 256-term chain per point on the vector ALUs.
//
// Resources: one wave per SIMD (4 waves, 128 points per workgroup pass).  A wave holds a layer's input and output for its 32 points in registers (2 x 128, the
// unified VGPR + AGPR file); the 1.9 MB of fp32 weight fragments stream L2 -> LDS by LDS-DMA in chunks of one neuron tile (8 / 32 / 40 KB), one chunk ahead,
// shared by the four waves.  Layers 1-4 and 6-7 have the same shape and run the same unrolled code (1 024 matrix instructions) in a loop.
#include "mlp.h"

namespace nrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace nsig {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NW = 4;
constexpr int BLK = 32 * NW;
constexpr int NL = 8;
__host__ __device__ constexpr int kdim(int l) { return l == 0 ? 64 : l == 5 ? 320 : 256; }
__host__ __device__ constexpr int groups(int l) { return kdim(l) / 8; }                    // float4 fragment groups (4 k-steps each) per neuron tile
__host__ __device__ constexpr int chunk_first_group(int l, int mt)                          // offset of chunk (l, mt) in the image, in groups
{
    int n = 0;
    for (int i = 0; i < l; i++) n += 8 * groups(i);
    return n + mt * groups(l);
}
constexpr int SIGMA_GROUPS = chunk_first_group(NL, 0);
// the colour branch behind it, as fp16 (hi, lo) fragment pairs (1 KB each, same DMA units): four tiles of views_linears_0 o feature_linear (16 chained k-steps of
// h8 + 2 natural ones of the direction encoding) and the rgb tile (8 chained k-steps)
// The views layer runs K-OUTER: a chunk holds four k-steps of ALL four neuron tiles ([tile][k][hi | lo]), so that each operand fragment of h8 is converted once, on the
// fly from its fp32 tile, and used by the four tiles' matrix instructions at once -- h8 is never held as 128 registers of fragments next to its 128 of fp32.
constexpr int RGB_KS = 8;                             // (the views layer has 16 chained + 2 natural k-steps)
constexpr int VIEW_CHUNK_KS = 4;
constexpr int VIEW_GROUPS = 4 * VIEW_CHUNK_KS * 2, VIEW_LAST_GROUPS = 4 * 2 * 2, RGB_GROUPS = 2 * RGB_KS;      // 32, 16 (the two direction k-steps), 16
constexpr int TOTAL_GROUPS = SIGMA_GROUPS + 4 * VIEW_GROUPS + VIEW_LAST_GROUPS + RGB_GROUPS;
constexpr int MAXG = 40;
constexpr int BIAS_FLOATS = NL * 256;                 // [layer][tile][lane half][16]
constexpr int ALPHA_FLOATS = 256 + 4;                 // alpha_linear.weight [256] | bias | pad
constexpr int VIEW_BIAS_FLOATS = 128 + 4;             // merged bias [tile 4][lane half 2][16] | rgb bias [3] | pad
constexpr size_t LDS_BYTES = (size_t)2 * MAXG * 1024 + (BIAS_FLOATS + ALPHA_FLOATS + VIEW_BIAS_FLOATS) * 4;
static_assert(SIGMA_GROUPS == 8 * (8 + 6 * 32 + 40) && VIEW_GROUPS <= MAXG, "image size");
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__host__ __device__ inline int perm_row(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }      // row of a 32x32 D tile in register 8s + j of lane half h

__host__ __device__ inline int row_neuron(int i) { return 2 * (4 * (i >> 3) + (i & 3)) + ((i >> 2) & 1); }

struct Ctx {
    f32x4 *wbuf;                 // [2][MAXG * 64]
    const float *bias_s;         // LDS
    const f32x4 *image;          // global
    int lane, hh, wave;
    int cur;                     // LDS buffer holding the chunk being consumed
    int next_group;              // image offset (groups) of the chunk after the one being consumed
};

// the DMA of one chunk (ng groups starting at image group g0) into LDS buffer `dst`: wave w moves groups w, w + 4, ...; 1 KB per instruction
__device__ __forceinline__ void stage_chunk(const Ctx &cx, f32x4 *dst, int g0, int ng)
{
    // address = SGPR base (the group's offset added on the scalar side, then opaque) + lane * 16: the saddr form of the DMA, no per-lane 64-bit add
    for (int q = cx.wave; q < ng; q += NW) {
        const f32x4 *sb = cx.image + (size_t)(g0 + q) * 64;
        asm volatile("" : "+s"(sb));
        __builtin_amdgcn_global_load_lds(sb + cx.lane, (__attribute__((address_space(3))) void *)(dst + q * 64), 16, 0, 0);
    }
}

// One neuron tile: G groups of four k-steps, A fragments from LDS (read through asm so that the compiler does not order them against the look-ahead DMA into the
// OTHER buffer), B operand bfn(ks); returns the finished tile with bias and ReLU applied.  NEXT_G: groups of the chunk to prefetch.
// Tried: accumulating tile mt straight into the caller's output tile while the PREVIOUS tile's bias / ReLU epilogue runs behind this tile's first matrix
// instructions (one wave per SIMD: nothing else overlaps those ~40 vector instructions).  The extra live ranges cost 141 spilled registers on top of the
// 499 in use, the build was no faster (342 ms per frame's coarse pass, inside this kernel's 328-357 ms run-to-run spread) and it failed the bit-exactness test
// (not chased further) -- not kept.
template <int G, class BFn>
__device__ __forceinline__ f32x16 tile(Ctx &cx, int layer, int mt, int next_ng, BFn bfn)
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 *w = cx.wbuf + cx.cur * (MAXG * 64);
    f32x4 *nxt = cx.wbuf + (cx.cur ^ 1) * (MAXG * 64);
    // The look-ahead DMA of the next chunk: this wave's pieces (groups wave, wave + 4, ...: 2-10 of them) go out one per group behind the tile's FIRST matrix
    // instructions -- a piece costs its issuing wave ~60 cycles (MI355X_MICROARCH.md), which overlap the 256 pipe cycles of the group in flight instead of standing in
    // front of the tile's first matrix instruction -- and have the remaining >= 20 groups (~5 000 cycles) to land before the end-of-tile wait.  Worth ~1 %: same
    // box, alternating runs, this / one burst per tile: 355.6 / 357.0 and 328.3 / 332.1 ms per frame's coarse pass (the kernel's time moves 8 % from run to run with
    // the clock; a first comparison across two gpurun calls had suggested 9 %).
#ifndef NRF_NSIG_DMA_SPREAD
#define NRF_NSIG_DMA_SPREAD 1
#endif
    const f32x4 *dsrc_s = cx.image + (size_t)cx.next_group * 64;             // uniform: the per-lane part (lane * 16) is the DMA's vector offset
    const int npieces = (next_ng - cx.wave + NW - 1) / NW;
    if (!NRF_NSIG_DMA_SPREAD) stage_chunk(cx, nxt, cx.next_group, next_ng);
    cx.next_group += next_ng;
    if (cx.next_group >= TOTAL_GROUPS) cx.next_group = 0;
    const uint32_t waddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(w + cx.lane);
    f32x4 fa[3];
    // fragments two groups ahead through three register slots, counted waits
#define NRF_RD(g_, slot_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fa[slot_]) : "v"(waddr), "i"((g_) * 1024))          /* offset in the instruction, not a v_add_u32 per read */
    NRF_RD(0, 0);
    if (G > 1) NRF_RD(1, 1);
    f32x16 acc = zero;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g + 2 < G) {
            NRF_RD(g + 2, (g + 2) % 3);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[g % 3]));
        } else if (g + 1 < G) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[g % 3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[g % 3]));
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 a4 = fa[g % 3];
#pragma unroll
        for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], bfn(4 * g + j), acc, 0, 0, 0);
        if (NRF_NSIG_DMA_SPREAD && g < 10 && g < npieces) {
            const int q = cx.wave + NW * g;
            const f32x4 *sb = dsrc_s + (size_t)q * 64;
            asm volatile("" : "+s"(sb));
            __builtin_amdgcn_global_load_lds(sb + cx.lane, (__attribute__((address_space(3))) void *)(nxt + q * 64), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NRF_NSIG_DMA_SPREAD) {
        for (int g = G < 10 ? G : 10; g < npieces; g++) {          // layer 0's 8-group tiles ahead of a 10-piece chunk
            const int q = cx.wave + NW * g;
            const f32x4 *sb = dsrc_s + (size_t)q * 64;
            asm volatile("" : "+s"(sb));
            __builtin_amdgcn_global_load_lds(sb + cx.lane, (__attribute__((address_space(3))) void *)(nxt + q * 64), 16, 0, 0);
        }
    }
#undef NRF_RD
    // the next chunk has landed (this wave's pieces) and every wave is done with this buffer
    // bias (one rounding, after the chain: the oracle's `acc += b`) and ReLU; [layer][tile][lane half][16] so that a lane reads 16 consecutive floats
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(cx.bias_s + ((layer * 8 + mt) * 2 + cx.hh) * 16);
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
        const f32x4 bv = b4[q4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float v = acc[4 * q4 + e] + bv[e];
            acc[4 * q4 + e] = fmaxf(v, 0.0f);
        }
    }
    // the next chunk has landed (this wave's pieces) and every wave is done with this buffer
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    cx.cur ^= 1;
    return acc;
}

__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}

// registers 8s..8s+7 of an fp32 tile -> the (hi, lo) fp16 operand fragments of one k-step (v = hi + lo to 22 bits).  The asm reads VALU results only (the max).
__device__ __forceinline__ void tile_to_frag2(const f32x16 &t, int s, float lim, half8 &hi, half8 &lo)
{
    union { half8 v; uint32_t u[4]; } h, l;
#pragma unroll
    for (int j = 0; j < 4; j++) split_pair(fmaxf(t[8 * s + 2 * j], lim), fmaxf(t[8 * s + 2 * j + 1], lim), h.u[j], l.u[j]);
    hi = h.v; lo = l.v;
}

// One tile of the colour branch in split precision (mlp_nerf_split_mfma.hip's arithmetic: Wl.xh + Wh.xl + Wh.xh per k-step into one fp32 accumulator): KS k-steps,
// fragments (hi, lo) adjacent in the LDS chunk, B operand pairs from bfn(ks, part).  Same chunk / look-ahead protocol as tile().
template <int KS, class BFn>
__device__ __forceinline__ f32x16 tile16(Ctx &cx, int next_ng, BFn bfn)
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 *w = cx.wbuf + cx.cur * (MAXG * 64);
    f32x4 *nxt = cx.wbuf + (cx.cur ^ 1) * (MAXG * 64);
    stage_chunk(cx, nxt, cx.next_group, next_ng);
    cx.next_group += next_ng;
    if (cx.next_group >= TOTAL_GROUPS) cx.next_group = 0;
    const uint32_t waddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(w + cx.lane);
    half8 fa[2][2];
#define NRF_RD2(k_, slot_) asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(fa[slot_][0]), "=&v"(fa[slot_][1]) : "v"(waddr), "i"((k_) * 2048), "i"((k_) * 2048 + 1024))
    NRF_RD2(0, 0);
    f32x16 acc = zero;
#pragma unroll
    for (int k = 0; k < KS; k++) {
        if (k + 1 < KS) {
            NRF_RD2(k + 1, (k + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[k & 1][0]), "+v"(fa[k & 1][1]));
        } else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k & 1][0]), "+v"(fa[k & 1][1]));
        __builtin_amdgcn_sched_barrier(0);
        const half8 ah = fa[k & 1][0], al = fa[k & 1][1];
        const half8 bh = bfn(k, 0), bl = bfn(k, 1);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef NRF_RD2
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    cx.cur ^= 1;
    return acc;
}

// K (<= 4) k-steps of the views layer for all four neuron tiles: acc[t] += (Wl.xh + Wh.xl + Wh.xh)(k) with the chunk laid out [tile][k][hi | lo].  The eight
// fragments of a k-step are read together; bfn(k, hi, lo) yields the operand pair.
template <int K, class BFn>
__device__ __forceinline__ void views_chunk(Ctx &cx, int next_ng, bool first, BFn bfn, f32x16 (&acc)[4])
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 *w = cx.wbuf + cx.cur * (MAXG * 64);
    f32x4 *nxt = cx.wbuf + (cx.cur ^ 1) * (MAXG * 64);
    stage_chunk(cx, nxt, cx.next_group, next_ng);
    cx.next_group += next_ng;
    if (cx.next_group >= TOTAL_GROUPS) cx.next_group = 0;
    const uint32_t waddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(w + cx.lane);
#pragma unroll
    for (int k = 0; k < K; k++) {
        half8 fa[4][2];
#pragma unroll
        for (int t = 0; t < 4; t++)
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(fa[t][0]), "=&v"(fa[t][1]) : "v"(waddr), "i"((t * K + k) * 2048), "i"((t * K + k) * 2048 + 1024));
        half8 bh, bl;
        bfn(k, bh, bl);                                     // the conversion runs while the reads are in flight
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t][1], bh, (first && k == 0) ? zero : acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t][0], bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t][0], bh, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    cx.cur ^= 1;
}

// image: fragments of the 64 + 6 chunks | bias [8][8][2][16] | alpha weight [256], alpha bias, pad | merged bias [4][2][16], rgb bias [3], pad
__global__ void __launch_bounds__(64 * NW, 1)
k_sigma_nerf_f32(int64_t npts, const float *__restrict__ pts, const float *__restrict__ rays, int ray_stride, const float *__restrict__ z, int s,
                 const _Float16 *__restrict__ dirs, const _Float16 *__restrict__ dirs_lo, const float *__restrict__ image, float *__restrict__ raw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *wbuf = reinterpret_cast<f32x4 *>(smem);
    float *bias_s = reinterpret_cast<float *>(smem + (size_t)2 * MAXG * 1024);
    float *alpha_s = bias_s + BIAS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const float *tail = image + (size_t)TOTAL_GROUPS * 256;
    for (int i = tid; i < BIAS_FLOATS + ALPHA_FLOATS + VIEW_BIAS_FLOATS; i += blockDim.x) bias_s[i] = tail[i];
    const float *vbias_s = alpha_s + ALPHA_FLOATS;
    Ctx cx{wbuf, bias_s, reinterpret_cast<const f32x4 *>(image), lane, hh, wave, 0, 0};
    stage_chunk(cx, wbuf, 0, groups(0));                           // chunk (0, 0) into buffer 0
    cx.next_group = groups(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int64_t nblocks = (npts + BLK - 1) / BLK;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t p0 = blk * BLK + wave * 32;
        int64_t q = p0 + r;
        if (q >= npts) q = npts - 1;                               // clamp loads; the store is guarded
        // ---- the 63 sinusoidal features of this lane's point (NeRF.cpp:33-37; the stand-alone encoder's arithmetic): k-step ks takes entries 2 ks + hh ----
        float px[3];
        if (pts) { px[0] = pts[q * 3]; px[1] = pts[q * 3 + 1]; px[2] = pts[q * 3 + 2]; }
        else {
            const float *rp = rays + (int64_t)((uint32_t)q / (uint32_t)s) * ray_stride;
            const float zz = z[q];
            px[0] = rp[0] + rp[3] * zz; px[1] = rp[1] + rp[4] * zz; px[2] = rp[2] + rp[5] * zz;
        }
        float xin[32];
#pragma unroll
        for (int ks = 0; ks < 32; ks++) {
            constexpr auto arg_axis = [](int k) { return k < 3 ? k : ((k - 3) % 6) % 3; };
            constexpr auto arg_freq = [](int k) { return k < 3 ? 0 : (k - 3) / 6; };
            constexpr auto kind = [](int k) { return k < 3 ? 0 : k >= 63 ? 3 : (((k - 3) % 6) < 3 ? 1 : 2); };   // 0 raw, 1 sin, 2 cos, 3 pad
            const int k0 = 2 * ks, k1 = 2 * ks + 1;
            const float a0 = px[arg_axis(k0)] * __builtin_ldexpf(1.0f, arg_freq(k0));
            const float a1 = px[arg_axis(k1 < 63 ? k1 : 0)] * __builtin_ldexpf(1.0f, arg_freq(k1 < 63 ? k1 : 0));
            float sn, cs;
            nrf_sincosf(hh ? a1 : a0, &sn, &cs);
            const int kd0 = kind(k0), kd1 = kind(k1);
            const float v0 = kd0 == 0 ? px[arg_axis(k0)] : kd0 == 1 ? sn : kd0 == 2 ? cs : 0.0f;
            const float v1 = kd1 == 0 ? px[arg_axis(k1 < 63 ? k1 : 0)] : kd1 == 1 ? sn : kd1 == 2 ? cs : 0.0f;
            xin[ks] = hh ? v1 : v0;
        }
        f32x16 hin[8], hout[8];
        // ---- layer 0: 63 (+ 1) -> 256 ----
#pragma unroll
        for (int mt = 0; mt < 8; mt++) hout[mt] = tile<groups(0)>(cx, 0, mt, mt < 7 ? groups(0) : groups(1), [&](int ks) { return xin[ks]; });
#pragma unroll
        for (int t = 0; t < 8; t++) hin[t] = hout[t];
        // ---- layers 1..7: 256 -> 256, layer 5 with the skip-concat in front (k-steps 0..31: the input features, then the 128 of h) ----
        for (int l = 1; l < NL; l++) {
            if (l == 5) {
#pragma unroll
                for (int mt = 0; mt < 8; mt++)
                    hout[mt] = tile<groups(5)>(cx, 5, mt, mt < 7 ? groups(5) : groups(6), [&](int ks) { return ks < 32 ? xin[ks < 32 ? ks : 0] : hin[ks >= 32 ? (ks - 32) >> 4 : 0][(ks - 32) & 15]; });
            } else {
                const int ng_after = l == 4 ? groups(5) : l == 7 ? VIEW_GROUPS : groups(1);
#pragma unroll
                for (int mt = 0; mt < 8; mt++) hout[mt] = tile<groups(1)>(cx, l, mt, mt < 7 ? groups(1) : ng_after, [&](int ks) { return hin[ks >> 4][ks & 15]; });
            }
#pragma unroll
            for (int t = 0; t < 8; t++) hin[t] = hout[t];
        }
        // ---- alpha_linear: a 256-term ascending chain per point; lanes 0-31 run it (even k is their own register, odd k comes from lane + 32) ----
        float a = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const f32x4 *wa = reinterpret_cast<const f32x4 *>(alpha_s + 32 * t);
#pragma unroll
            for (int q4 = 0; q4 < 8; q4++) {
                const f32x4 w4 = wa[q4];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int qq = 2 * q4 + e;
                    const uint32_t v = __float_as_uint(hin[t][qq]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);       // [1]: lanes 0-31 receive lanes 32-63 (the odd neuron of the same point)
                    a = __builtin_fmaf(w4[2 * e], hin[t][qq], a);
                    a = __builtin_fmaf(w4[2 * e + 1], __uint_as_float(sw[1]), a);
                }
            }
        }
        a = a + alpha_s[256];
        // ---- the colour branch on the exact h8, in split precision on the fp16 matrix instructions (2 % of this kernel's matrix instructions' time): the coarse pass
        // then leaves whole (rgb, sigma) rows and the fine pass need not evaluate the network at its 64 coarse depths again.
        //   rgb = rgb_linear(relu(views_linears_0(cat[feature_linear(h8), views])))        (NeRF.cpp:108-120), feature_linear folded into views_linears_0 at pack time
        // h8 is the chained operand: k-step 2 t + s' = registers 8 s'..8 s'+7 of tile t (neuron 32 t + 2 (8 s' + j) + h), split into (hi, lo) when its k-step comes up
        // (h8 >= 0 already: the max inside the split is the identity -- and a VALU result for the asm to read).
        f32x16 vacc[4];
#pragma unroll
        for (int ck = 0; ck < 4; ck++)
            views_chunk<VIEW_CHUNK_KS>(cx, ck < 3 ? VIEW_GROUPS : VIEW_LAST_GROUPS, ck == 0, [&](int k, half8 &bh, half8 &bl) {
                const int c = 4 * ck + k;                        // chained k-step 0..15
                tile_to_frag2(hin[c >> 1], c & 1, 0.0f, bh, bl);
            }, vacc);
        {
            // PE(4) of the view direction, one row per ray: natural order, element j of k-step s = dirs[16 s + 8 h + j]
            const int64_t doff = (int64_t)((uint32_t)q / (uint32_t)s) * 32;              // point q belongs to ray q / s, explicit points included ([n, s, 3])
            views_chunk<2>(cx, RGB_GROUPS, false, [&](int k, half8 &bh, half8 &bl) {
                bh = *reinterpret_cast<const half8 *>(dirs + doff + 16 * k + 8 * hh);
                bl = dirs_lo ? *reinterpret_cast<const half8 *>(dirs_lo + doff + 16 * k + 8 * hh) : half8{0, 0, 0, 0, 0, 0, 0, 0};
            }, vacc);
        }
        half8 bb[8][2];                                         // relu(views layer), 128 wide: k-step 2 t + s' of the rgb tile
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x16 v = vacc[t];
            const f32x4 *b4 = reinterpret_cast<const f32x4 *>(vbias_s + (t * 2 + hh) * 16);
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const f32x4 bv = b4[q4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[4 * q4 + e] = v[4 * q4 + e] + bv[e];
            }
#pragma unroll
            for (int sp = 0; sp < 2; sp++) tile_to_frag2(v, sp, 0.0f, bb[2 * t + sp][0], bb[2 * t + sp][1]);           // ReLU inside the split
        }
        const f32x16 c = tile16<RGB_KS>(cx, groups(0), [&](int k, int part) { return bb[k][part]; });
        const int64_t p = p0 + r;
        if (hh == 0 && p < npts)                                // rows 0..2 of the rgb tile sit in registers 0..2 of lane half 0
            *reinterpret_cast<float4 *>(raw + p * 4) = float4{c[0] + vbias_s[128], c[1] + vbias_s[129], c[2] + vbias_s[130], a};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead DMA of the last tile
}

}  // namespace nsig

static bool sigma_nerf_f32_supported(const nrf_mlp_nerf_desc &d)
{
    return d.depth == 8 && d.width == 256 && d.input_ch == 63 && d.skip == 4 && d.use_viewdirs;
}

// The exact-fp32 density image of the classic network as a host vector: fp32 fragments and biases with a block of fp16 (hi, lo) fragments in between (floats
// [f16_at, f16_at + f16_floats) hold two fp16 values each).  As nerf_f16_images_host: every entry a copy (or fp16 hi / lo half) of one entry of hp / merged / merged_b.
bool nerf_sigma_image_host(const nrf_mlp_nerf_desc &d, const float *hp, const float *merged, const float *merged_b, std::vector<float> &img, size_t &f16_at, size_t &f16_floats)
{
    if (!sigma_nerf_f32_supported(d)) return false;
    using namespace nsig;
    img.clear();
    img.reserve((size_t)TOTAL_GROUPS * 256 + BIAS_FLOATS + ALPHA_FLOATS + VIEW_BIAS_FLOATS);
    std::vector<float> bias((size_t)BIAS_FLOATS, 0.0f);
    size_t off = 0;
    // the fp32 fragments of the eight pts_linears, written by index on up to 8 host threads (a training loop re-packs every step): unit = (layer, row tile)
    size_t layer_at[NL + 1], layer_off[NL];
    layer_at[0] = 0;
    for (int l = 0; l < NL; l++) {
        const int in = l == 0 ? 63 : l == 5 ? 63 + 256 : 256;
        layer_off[l] = off; off += (size_t)in * 256 + 256;
        layer_at[l + 1] = layer_at[l] + (size_t)8 * groups(l) * 256;
    }
    img.resize(layer_at[NL]);
    host_parallel_for(NL * 8, [&](int u0, int u1) {
        for (int u = u0; u < u1; u++) {
            const int l = u >> 3, mt = u & 7;
            const int in = l == 0 ? 63 : l == 5 ? 63 + 256 : 256;
            const float *w = hp + layer_off[l];
            // padded k -> column of W: layer 0: k < 63; layer 5: [input 0..62 | pad | h 0..255] -> columns [0..62 | - | 63..318]
            auto col = [&](int k) { return l == 0 ? (k < 63 ? k : -1) : l == 5 ? (k < 63 ? k : k == 63 ? -1 : k - 1) : k; };
            float *dst = img.data() + layer_at[l] + (size_t)mt * groups(l) * 256;
            for (int g = 0; g < groups(l); g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 4; j++) {
                        const int row = 32 * mt + row_neuron(lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
                        const int c = col(k);
                        *dst++ = c >= 0 ? w[(size_t)row * in + c] : 0.0f;
                    }
        }
    });
    for (int l = 0; l < NL; l++) {
        const int in = l == 0 ? 63 : l == 5 ? 63 + 256 : 256;
        const float *b = hp + layer_off[l] + (size_t)in * 256;
        for (int mt = 0; mt < 8; mt++)
            for (int h = 0; h < 2; h++)
                for (int q = 0; q < 16; q++) bias[((size_t)(l * 8 + mt) * 2 + h) * 16 + q] = b[32 * mt + 2 * q + h];
    }
    // blob order after pts_linears: views_linears_0 (w, b), feature_linear (w, b), alpha_linear (w, b), rgb_linear (w, b)
    const int V = d.input_ch_views;                                  // 27
    const float *wv = hp + off;
    off += (size_t)(V + 256) * 128 + 128;
        off += (size_t)256 * 256 + 256;
    const float *aw = hp + off;
    off += 256 + 1;
    const float *wr = hp + off, *br = wr + (size_t)128 * 3;
    // (hi, lo) fp16 fragments of the colour branch: per k-step the hi fragment then the fragment of the rounding residuals
    std::vector<_Float16> himg;
    himg.reserve((size_t)(4 * VIEW_GROUPS + VIEW_LAST_GROUPS + RGB_GROUPS) * 512);
    auto push_frag = [&](auto value_of /* (lane, j) -> float */) {
        for (int part = 0; part < 2; part++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const float v = value_of(lane, j);
                    const _Float16 hv = (_Float16)v;
                    himg.push_back(part == 0 ? hv : (_Float16)(v - (float)hv));
                }
    };
    auto view_value = [&](int t, int k, int lane, int j) -> float {
        const int row = 32 * t + (lane & 31), h = lane >> 5;
        if (k < 16) return merged[(size_t)row * 256 + 32 * (k >> 1) + 2 * (8 * (k & 1) + j) + h];       // h8 as this kernel's fp32 tiles hold it: neuron 2 q + h in register q
        const int idx = 16 * (k - 16) + 8 * h + j;                                                       // direction encoding, natural order
        return idx < V ? wv[(size_t)row * (V + 256) + 256 + idx] : 0.0f;
    };
    for (int ck = 0; ck < 5; ck++) {                             // chunks of four k-steps (the last: the two direction k-steps), [tile][k][hi | lo] inside
        const int k0 = 4 * ck, kn = ck < 4 ? VIEW_CHUNK_KS : 2;
        for (int t = 0; t < 4; t++)
            for (int k = 0; k < kn; k++)
                push_frag([&](int lane, int j) -> float { return view_value(t, k0 + k, lane, j); });
    }
    for (int k = 0; k < RGB_KS; k++)
        push_frag([&](int lane, int j) -> float {
            const int row = lane & 31, h = lane >> 5;
            return row < 3 ? wr[(size_t)row * 128 + 32 * (k >> 1) + perm_row(k & 1, h, j)] : 0.0f;                // the views tiles are plain 32x32x16 D tiles: row 16 s + 8 (j >> 2) + 4 h + (j & 3)
        });
    {
        const size_t nf = himg.size() / 2, at = img.size();
        f16_at = at; f16_floats = nf;
        img.resize(at + nf);
        memcpy(img.data() + at, himg.data(), nf * sizeof(float));
    }
    img.insert(img.end(), bias.begin(), bias.end());
    for (int k = 0; k < 256; k++) img.push_back(aw[k]);
    img.push_back(aw[256]);
    for (int k = 0; k < 3; k++) img.push_back(0.0f);
    for (int t = 0; t < 4; t++)
        for (int h = 0; h < 2; h++)
            for (int q = 0; q < 16; q++) img.push_back(merged_b[32 * t + 8 * (q >> 2) + 4 * h + (q & 3)]);
    for (int k = 0; k < 3; k++) img.push_back(br[k]);
    img.push_back(0.0f);
    return true;
}

int mlp_nerf_pack_sigma_f32(nrf_mlp *m, const std::vector<float> &hp)
{
    const auto &d = m->nerf;
    if (!sigma_nerf_f32_supported(d)) return NRF_OK;
    // views_linears_0 o feature_linear (no activation between them, NeRF.cpp:112-115): merged[r][k] = sum_f Wv[r][f] Wf[f][k], merged_b[r] = sum_f Wv[r][f] bf[f] + bv[r], in double
    std::vector<float> merged, merged_b;
    if (m->host_merged.size() == (size_t)128 * 256 && m->host_merged_b.size() == 128) { merged.swap(m->host_merged); merged_b.swap(m->host_merged_b); }          // the same upload's product (mlp_nerf_pack_f16)
    else {
        const int V = d.input_ch_views;
        const size_t o_views = nerf_blob_offset_views(), o_feat = o_views + (size_t)(V + 256) * 128 + 128;
        nerf_merged_views_host(hp.data() + o_views, V + 256, hp.data() + o_feat, hp.data() + o_feat + (size_t)256 * 256, hp.data() + o_views + (size_t)(V + 256) * 128, 128, 256, merged, merged_b);
    }
    m->host_merged.clear(); m->host_merged_b.clear();
    std::vector<float> img;
    size_t f16_at = 0, f16_floats = 0;
    if (!nerf_sigma_image_host(d, hp.data(), merged.data(), merged_b.data(), img, f16_at, f16_floats)) return NRF_ERR_INVALID_ARG;
    const size_t bytes = img.size() * sizeof(float);
    if (m->d_packed_sigma_f32 && m->packed_sigma_f32_bytes != bytes) { (void)hipFree(m->d_packed_sigma_f32); m->d_packed_sigma_f32 = nullptr; }
    if (!m->d_packed_sigma_f32) NRF_HIP(hipMalloc(&m->d_packed_sigma_f32, bytes));
    m->packed_sigma_f32_bytes = bytes;
    NRF_HIP(hipMemcpy(m->d_packed_sigma_f32, img.data(), bytes, hipMemcpyHostToDevice));
    return NRF_OK;
}

int mlp_nerf_sigma_f32_available(const nrf_mlp *m) { return m && m->family == MLP_NERF && m->d_packed_sigma_f32 != nullptr; }

// The classic network's coarse pass: raw [p, 4] = (rgb, sigma) with sigma bit-identical to NRF_PREC_F32 (exact fp32 on the matrix cores) and rgb from the exact h8 in
// split precision.  Points explicit (pts [p, 3]) or formed from (rays, z) as o + d z; dirs / dirs_lo: PE(4) of the view direction as fp16 (hi, lo) rows [n, 32]
// (launch_dirs_pe_split; dirs_lo may be null), one row per ray (point i belongs to ray i / s).
int mlp_nerf_exact_coarse(const nrf_mlp *m, const float *pts, const float *rays, int ray_stride, const float *z, int s, const __half *dirs, const __half *dirs_lo,
                          int64_t p, float *raw, hipStream_t st)
{
    if (!mlp_nerf_sigma_f32_available(m)) { set_error("internal: fp32 matrix-core sigma image of the classic network missing"); return NRF_ERR_UNSUPPORTED; }
    if (p == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_SIGMA, st);
    NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nsig::k_sigma_nerf_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsig::LDS_BYTES));
    const int64_t nblocks = ceil_div(p, nsig::BLK);
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);          // persistent: one 4-wave workgroup per CU
    hipLaunchKernelGGL(nsig::k_sigma_nerf_f32, dim3(grid), dim3(64 * nsig::NW), nsig::LDS_BYTES, st, p, pts, rays, ray_stride, z, s,
                       reinterpret_cast<const _Float16 *>(dirs), reinterpret_cast<const _Float16 *>(dirs_lo), reinterpret_cast<const float *>(m->d_packed_sigma_f32), raw);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf
