// mlp_small_mfma.hip -- NeRFSmallImpl::forward (NeRF.cpp:322-412) on the gfx950 matrix cores, NRF_PREC_F16_MFMA.
//
// Formulation: every layer is computed TRANSPOSED, H_{l+1}^T [neurons x points] = W_{l+1} [neurons x k] . H_l^T [k x points],
// with v_mfma_f32_32x32x16_f16: A = a 32-neuron x 16-k weight fragment, B = a 16-k x 32-point activation fragment,
// D = 32 neurons x 32 points in fp32.  The D layout (column = point on the lane, rows = neurons in the 16 registers) is
// exactly what the NEXT layer's B operand wants when it sums over the neuron index: registers 8s..8s+7 of a D tile,
// converted to fp16 after the ReLU, ARE the B fragment of k-step s -- no LDS round trip, no lane movement.  The only
// price is a fixed permutation of k inside a k-step (element j of lane-half h is neuron 16s + 8(j>>2) + 4h + (j&3)),
// which is folded into the weight image at pack time.  So one wavefront carries 64 points (two 32-point tiles) through
// the whole network in registers; the ~40 KB fp16 weight image sits in LDS for the life of the (persistent) workgroup
// and every A fragment is one conflict-free ds_read_b128 shared by both point tiles.
//
// Layer plan (hidden = hidden_color = 64, 1+geo <= 32):
//   sigma net : x[in] -> 64 -> ... -> (1+geo)        ReLU between, none at the end          (NeRF.cpp:372-381)
//   colour net: cat[views, geo] -> 64 -> ... -> 3     the geo rows come straight from the sigma net's last D tile
//   out = (rgb, sigma)                                                                       (NeRF.cpp:408)
//
// NRF_PREC_F16_SPLIT (template SPLIT): every fp32 quantity v is carried as an UNEVALUATED SUM of two fp16 numbers,
// v = hi + lo with hi = f16(v), lo = f16(v - hi) (22 significant bits), weights split once at pack time, activations split
// when a D tile becomes the next B fragment.  A product W.x is then three matrix-core instructions accumulating into the same
// fp32 tile:  Wh.xh + Wl.xh + Wh.xl  (the dropped Wl.xl term is 2^-22 relative).  The CuHashEmbedder features are exactly
// fp16 in the reference itself (CuHashEmbedder.cu:95), so the first layer needs only two.  Result: fp32-grade pixels
// (render-vs-oracle PSNR > 90 dB where the plain fp16 mode gives ~45 dB on the adversarial synthetic scene) at 3x the MFMA
// work of a kernel that was not MFMA-bound to begin with.  The doubled weight image (80 KB) is shared by 8 waves per workgroup.
#include "mlp.h"

#include <type_traits>

namespace nrf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef NRF_SMALL_TRACE
// diagnostic build only (tools/scratch/small_trace.py): cycle stamps of wave 0 of every workgroup of the split kernel, summed per section:
// [0..2] sigma layers, [3] colour-net operand (geo split), [4..7] colour layers, [8] epilogue + store + operand hand-over, [9] whole iterations, [10] iterations
__device__ unsigned long long g_small_trace[256 * 12];
#define NRF_TSTAMP(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tr[i] += t__ - tprev; tprev = t__; } while (0)
#else
#define NRF_TSTAMP(i) do { } while (0)
#endif
#ifndef NRF_SMALL_PT
#define NRF_SMALL_PT 2
#endif
constexpr int PT = NRF_SMALL_PT; // 32-point tiles per wave
#ifndef NRF_SPLIT_WAVES
#define NRF_SPLIT_WAVES 8
#endif
#ifndef NRF_SPLIT_MINWAVES
#define NRF_SPLIT_MINWAVES 1           // waves per SIMD the register budget is sized for (2 with NRF_SPLIT_WAVES = 4: half the register file is left to other kernels)
#endif
constexpr int waves_of(bool split) { return split ? NRF_SPLIT_WAVES : 4; }
constexpr int block_pts_of(bool split) { return 32 * PT * waves_of(split); }

// neuron (row of a D tile / k of the next layer) held by element j of lane-half h in k-step s of a 32-row tile
__host__ __device__ inline int perm_row(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

#ifndef NRF_SPLIT_RTZ
#define NRF_SPLIT_RTZ 0             // split mode, hidden layers: 1 = ReLU folded into a truncating (hi, lo) split (2 instructions per value instead of 2.5);
                                    // measured 11.65 vs 11.79 ms (same box, warm) with the frame's max error vs fp32 5.2e-6 instead of 3.5e-6: off
#endif

// D tile registers 8s..8s+7 -> fp16 B fragment (round to nearest even), optional ReLU
template <bool RELU>
__device__ __forceinline__ half8 tile_to_frag(const f32x16 &acc, int s)
{
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (_Float16)acc[8 * s + j];
    // ReLU after the (monotonic) rounding: max(round(x), 0) == round(max(x, 0)); packed, 4 v_pk_max_f16 instead of 8 v_max_f32
    if (RELU) r = __builtin_elementwise_max(r, half8{0, 0, 0, 0, 0, 0, 0, 0});
    return r;
}

// D tile registers 8s..8s+7 -> (hi, lo) fp16 pair of B fragments: v = hi + lo to 22 bits.
// VALU cost matters here (the split kernel converts as many values as it multiplies tiles), so the hidden-layer form is 2.5
// instructions per value: one v_max_f32 (this file is built with -fno-honor-nans, otherwise fmaxf first canonicalises the MFMA
// result), half a v_cvt_pk_f16_f32 (RNE), and lo = f16(v - hi) as ONE mixed-precision FMA that reads hi as a half and writes a
// half (v_fma_mixlo/mixhi_f16: fma(f32(hi), -1, v), exact difference, rounded once) -- which the compiler does not select by
// itself (it emits cvt + sub + cvt).  The asm only ever reads compiler-produced VALU results, never an MFMA result directly,
// so the MFMA -> VALU hazard handling stays with the compiler.
__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}

// ReLU folded into the split, 2 instructions per value: hi' = f16(v) rounded TOWARD ZERO (v_cvt_pkrtz_f16_f32, two values per instruction), so that the
// residual v - hi' is zero or has the sign of v; then relu(v) = max(hi', 0) + max(v - hi', 0) exactly: the first max is one packed v_pk_max_f16 per pair,
// the second is the clamp bit of the mixed-precision FMA that forms the residual (clamp = [0, 1]; the residual of a finite half is below 32).  hi keeps
// 11 bits, lo 11 more starting at most one ulp(hi) down: v = hi + lo to 21 bits (22 with the round-to-nearest split).
// The cvt is the compiler's (builtin), reads the MFMA result first and so carries the MFMA -> VALU hazard handling; the asm FMAs depend on its result.
__device__ __forceinline__ void split_pair_relu(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 pre2 = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    const uint32_t pre = __builtin_bit_cast(uint32_t, pre2);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(lo) : "v"(pre), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(lo) : "v"(pre), "v"(v1));
    hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2, pre), f16x2{0, 0}));
}

template <bool RELU>
__device__ __forceinline__ void tile_to_frag2(const f32x16 &acc, int s, half8 &hi, half8 &lo)
{
    if constexpr (RELU) {
        union { half8 v; uint32_t u[4]; } h, l;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#if NRF_SPLIT_RTZ
            split_pair_relu(acc[8 * s + 2 * j], acc[8 * s + 2 * j + 1], h.u[j], l.u[j]);
#else
            split_pair(fmaxf(acc[8 * s + 2 * j], 0.0f), fmaxf(acc[8 * s + 2 * j + 1], 0.0f), h.u[j], l.u[j]);
#endif
        }
        hi = h.v; lo = l.v;
    } else {
        // same two roundings (hi = RNE(v), lo = RNE(v - hi)) through the 1.5-instruction-per-value path
        union { half8 v; uint32_t u[4]; } h, l;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // split_pair's asm must read VALU results, never the matrix instruction's destination directly (the MFMA -> VALU hazard handling is the compiler's and
            // does not look into asm): a max with -FLT_MAX is the cheapest instruction that is the identity on every finite value
            const float v0 = fmaxf(acc[8 * s + 2 * j], -3.402823466e38f), v1 = fmaxf(acc[8 * s + 2 * j + 1], -3.402823466e38f);
            split_pair(v0, v1, h.u[j], l.u[j]);
        }
        hi = h.v; lo = l.v;
    }
}

// acc[pt][mt] += A[mt][ks] . B[pt][ks] over all k-steps; A fragments stream from LDS in consumption order.
// NP = 1: plain fp16 operands.  NP = 2: split operands, fragments stored (hi, lo) adjacent; BLO = the B operand has a non-zero lo part.
struct NoJob { __device__ __forceinline__ void operator()(int, int, int, int) const {} };

// job(mt, ks, g, ng): extra work issued right after matrix instruction g of the ng of step (mt, ks) -- the software pipeline of the kernel puts the
// conversion of the PREVIOUS tile there, so that its VALU instructions execute while the MFMAs occupy the matrix pipe.  A 32x32x16 MFMA holds the SIMD's
// vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md): ~6 four-cycle vector instructions per gap are free, so the conversion work is dealt out in
// half-quad units (4-6 instructions) gap by gap instead of in one block per step (FINE: a scheduling fence after every matrix instruction keeps it there).
// The A fragments of the whole network lie in LDS in consumption order, so the fragment(s) of step i + 1 -- the next k-step, m-tile or
// LAYER -- are simply the next 1 (2) KB: they are fetched at the top of step i and arrive while its products run.  `pre` carries them
// from step to step and from layer to layer (fetched = false only for the very last step of the network).
template <int MT, int KS, int NP, bool BLO, bool LAST = false, bool FINE = false, int DROP = 0, class Job = NoJob>
__device__ __forceinline__ void gemm_layer(const half8 *__restrict__ frags, int lane, const half8 (&b)[PT][KS][NP], f32x16 (&acc)[PT][MT], half8 (&pre)[NP],
                                           Job job = Job())
{
    // the first product of every accumulator takes the literal zero as its C operand (an inline constant of the MFMA encoding)
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr bool P_WL = NP == 2 && !(DROP & 1);                   // Wl . xh
    constexpr bool P_XL = NP == 2 && BLO && !(DROP & 2);            // Wh . xl
    constexpr int NG = (NP == 2 ? ((P_WL ? 1 : 0) + (P_XL ? 1 : 0) + 1) : 1) * PT;          // matrix instructions per step
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const half8 a = pre[0];
            half8 al = a;
            if constexpr (NP == 2) al = pre[NP - 1];
            if (!(LAST && mt == MT - 1 && ks == KS - 1)) {
#pragma unroll
                for (int q = 0; q < NP; q++) pre[q] = frags[((mt * KS + ks + 1) * NP + q) * 64 + lane];
            }
            int g = 0;
            auto after = [&]() {
                job(mt, ks, g, NG);
                g++;
                if constexpr (FINE) __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (NP == 2) {
                // small terms first, the leading product last
                bool first = ks == 0;
                if constexpr (P_WL) {
#pragma unroll
                    for (int pt = 0; pt < PT; pt++) { acc[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[pt][ks][0], first ? zero : acc[pt][mt], 0, 0, 0); after(); }
                    first = false;
                }
                if constexpr (P_XL) {
#pragma unroll
                    for (int pt = 0; pt < PT; pt++) { acc[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[pt][ks][1], first ? zero : acc[pt][mt], 0, 0, 0); after(); }
                    first = false;
                }
#pragma unroll
                for (int pt = 0; pt < PT; pt++) { acc[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[pt][ks][0], first ? zero : acc[pt][mt], 0, 0, 0); after(); }
            } else {
#pragma unroll
                for (int pt = 0; pt < PT; pt++) { acc[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[pt][ks][0], ks == 0 ? zero : acc[pt][mt], 0, 0, 0); after(); }
            }
            // fence the scheduler: unfenced it hoists every ds_read_b128 of the network to the top (40 fragments = 160 VGPRs), which costs
            // the occupancy that hides the feature-load latency
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// number of 1-KB fragments of each layer, in consumption order
template <int IN_KS, int V_KS, int NL, int NLC>
struct SmallPlan {
    static constexpr int GEO_KS = 1;
    static constexpr int sigma_frags(int l) { return (l == NL - 1 ? 1 : 2) * (l == 0 ? IN_KS : 4); }
    static constexpr int color_frags(int l) { return (l == NLC - 1 ? 1 : 2) * (l == 0 ? (V_KS + GEO_KS) : 4); }
    static constexpr int sigma_total() { int t = 0; for (int l = 0; l < NL; l++) t += sigma_frags(l); return t; }
    static constexpr int total()
    {
        int t = 0;
        for (int l = 0; l < NL; l++) t += sigma_frags(l);
        for (int l = 0; l < NLC; l++) t += color_frags(l);
        return t;
    }
};

// Input of the fused kernel: either fp32 rows [p, in_ch + in_views] (the generic BaseNeRF::forward boundary) or the
// renderer's fast-path layout: level-major fp16 hash features feats[level][p] (half2), per-RAY direction features
// dirs[ray][V] (fp16) shared by the `s` samples of a ray (NeRFRenderer.h:179), and the embedder's keep mask, which is
// applied to sigma in the epilogue (NeRFRenderer.h:187-188).
struct SmallInput {
    const float *x; int x_stride; int in_ch;          // row-major fp32 input
    const __half2 *feats; int64_t pstride;            // level-major fp16 input
    const __half *dirs; int s;
    const uint8_t *keep;
    const __half *dirs_lo;                            // split mode: lo parts of the direction features, same layout
    const __half2 *feats_lo;                          // split mode, fp32-valued features (HashEmbedder): lo plane, same layout as feats
    const int32_t *src;                               // optional: point i reads column src[i] of feats / keep (the renderer's feature reuse: the fine pass's coarse depths
                                                      // point at the coarse pass's columns); NULL: column i
    // GEOIN (colour net only): the sigma net's output comes from the coarse pass's exact kernel (sigma_small_f32.hip, GEO) -- the (sigma, geo_feat) operand fragment of
    // point i as planes [hi | lo][geo_stride][2 lane halves] of 16 bytes, and sigma itself (keep mask applied) as [p] floats.  feats is not read.
    const half8 *geo; int64_t geo_stride;
    const float *sigma;
    // point -> ray without a division: ray = (p * ray_mul) >> (31 + ray_shift) for p < 2^31 (ray_mul = ceil(2^(31 + L) / s), L = ceil(log2 s); set by the launchers)
    uint32_t ray_mul; int ray_shift;
    // split mode: the powers of two that take the range scaling of the operand image out of the outputs again (mlp.h, SMALL_SCALE_*; all 1 while no scaling is in force)
    const float *scales;
};

#ifndef NRF_SMALL_PIPE_SPLIT
#define NRF_SMALL_PIPE_SPLIT 1
#endif
#ifndef NRF_SMALL_PIPE_F16
#define NRF_SMALL_PIPE_F16 0
#endif
#ifndef NRF_SMALL_PREFETCH_SPLIT
#define NRF_SMALL_PREFETCH_SPLIT 1
#endif
#ifndef NRF_SMALL_PREFETCH_F16
#define NRF_SMALL_PREFETCH_F16 0
#endif
#ifndef NRF_SMALL_STAGGER
#define NRF_SMALL_STAGGER 0         // split mode: waves 4-7 start this many x 8 128 cycles late
#endif
#ifndef NRF_SMALL_FINE
#define NRF_SMALL_FINE 1            // split mode: conversion work dealt out per matrix instruction (0: one block per step)
#endif
// Per-layer product budget of the split mode (tools/product_budget.py): layer ids 0..NL-1 = sigma net, NL.. = colour net.  Bit 2 id drops the
// Wl.xh product of layer id (the weights' rounding residuals), bit 2 id + 1 drops Wh.xl (the activations' residuals -- their (lo) halves are then not formed either).
// 0 (the shipped build): all three products everywhere.
#ifndef NRF_SMALL_DROP_MASK
#define NRF_SMALL_DROP_MASK 0
#endif
constexpr int small_drop_of(int id) { return (int)((((unsigned long long)NRF_SMALL_DROP_MASK) >> (2 * id)) & 3ull); }

// LMLO: the level-major features come as (hi, lo) planes (fp32-valued features of the LibTorch HashEmbedder); without it they are exact
// fp16 numbers (CuHashEmbedder rounds its output to fp16 itself, CuHashEmbedder.cu:95) and the layer-0 operand has no lo part.
// A32 (level-major input without a merge map, < 2^26 points, planes < 2^27 columns: every launch of the renderer's default mode): 32-bit addressing, see load_inputs
template <int IN_KS, int V_KS, int NL, int NLC, bool LM, bool SPLIT, bool LMLO = false, bool GEOIN = false, bool A32 = false>
__global__ void __launch_bounds__(64 * waves_of(SPLIT), SPLIT ? NRF_SPLIT_MINWAVES : 2)
k_mlp_small_mfma(int64_t npts, SmallInput in, const half8 *__restrict__ packed, float *__restrict__ out, int out_stride)
{
    using Plan = SmallPlan<IN_KS, V_KS, NL, NLC>;
    constexpr int NP = SPLIT ? 2 : 1;
    constexpr int BLOCK_PTS = block_pts_of(SPLIT);
    constexpr bool IN_LO = SPLIT && (!LM || LMLO);
    static_assert(!GEOIN || (LM && SPLIT && !LMLO), "the colour-only kernel exists in split precision on the level-major path");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *wl = reinterpret_cast<half8 *>(smem);
    constexpr int NFRAG = Plan::total() * NP;
    for (int i = threadIdx.x; i < NFRAG * 64; i += blockDim.x) wl[i] = packed[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t nblocks = (npts + BLOCK_PTS - 1) / BLOCK_PTS;
    // split mode: the image holds 2^e_l W_l per layer (both fp16 halves of every weight normal, activations near RMS 8 whatever the checkpoint's magnitudes: mlp.h); the
    // outputs carry the product of the scales, taken out in the epilogue -- powers of two, exact
    float inv_sigma = 1.0f, inv_rgb = 1.0f;
    if constexpr (SPLIT) { inv_sigma = in.scales[SMALL_SCALE_INV_SIGMA]; inv_rgb = in.scales[SMALL_SCALE_INV_RGB]; }
    (void)inv_sigma; (void)inv_rgb;
#if NRF_SMALL_STAGGER
    // SIMD partners (waves w and w + 4) run the same program and fall into lockstep -- both in their matrix phases, then both in their vector phases.
    // Waves 4-7 start a fraction of an iteration late (MI355X_MICROARCH.md, "two waves that run the SAME program: try a stagger").
    if (SPLIT && wave >= 4) {
#pragma unroll
        for (int i = 0; i < NRF_SMALL_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    auto split8 = [](const float4 &lo4, const float4 &hi4, half8 &hv, half8 &lv) {
        const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int j = 0; j < 8; j++) { const _Float16 t = (_Float16)v[j]; hv[j] = t; lv[j] = (_Float16)(v[j] - (float)t); }
    };
    // ---- layer-0 / colour-layer-0 B fragments of one block iteration: element j of k-step s is x[pt][16s + 8h + j] ----
    // The feature column of a point is src[p] when a caller passes a merge map (the renderer's feature-reusing fine pass outside the default mode -- plain fp16 with an
    // exact coarse pass; the default mode walks columns in order, A32 below): a load whose result is the ADDRESS of the feature loads, and of the keep-mask load in the epilogue.  Cycle stamps (tools/scratch/small_trace.py, one wave per SIMD) showed the wave waiting ~3 300 cycles for
    // src[p] before it could even issue its operand prefetch, and ~2 000-3 500 more at the end of the iteration for src[p] -> keep[src[p]]: a third of the iteration.
    // So the columns travel one iteration AHEAD of the operands (load_cols for block i + 2 while block i computes), and the keep byte is fetched with the operands.
    static_assert(!A32 || (LM && SPLIT), "32-bit addressing is instantiated for the split-precision level-major kernels");
    auto load_cols = [&](int64_t blk_, int32_t (&cols)[PT]) {
        if (A32 || !(LM && in.src)) return;
        const int64_t p0_ = blk_ * BLOCK_PTS + wave * (32 * PT);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            int64_t p = p0_ + pt * 32 + r;
            if (p >= npts) p = npts - 1;
            cols[pt] = (LM && in.src) ? in.src[p] : (int32_t)0;       // without a merge map the column is the point itself (formed where it is used)
        }
    };
    auto load_inputs = [&](int64_t blk_, const int32_t (&cols)[PT], half8 (&bx)[PT][IN_KS][NP], half8 (&bv)[PT][V_KS][NP], uint8_t (&kpv)[PT], half8 (&bg)[PT][NP], float (&sgv)[PT]) {
        const int64_t p0_ = blk_ * BLOCK_PTS + wave * (32 * PT);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            int64_t p = p0_ + pt * 32 + r;
            if (p >= npts) p = npts - 1;                 // clamp loads; stores are guarded
            kpv[pt] = 1;
            if constexpr (LM) {
                if constexpr (A32) {
                    // 32-bit addressing (the renderer's launches: no merge map, < 2^26 points, planes < 2^27 columns): per-lane byte offsets next to uniform plane bases,
                    // the ray index by a multiply -- the generic path below spends ~150 vector instructions per iteration on 64-bit indices and two divisions
                    uint32_t pu = (uint32_t)p0_ + (uint32_t)(pt * 32 + r);
                    pu = pu < (uint32_t)npts ? pu : (uint32_t)npts - 1u;
                    if (in.keep) kpv[pt] = in.keep[pu];
                    if constexpr (GEOIN) {
                        const uint32_t goff = (pu * 2u + (uint32_t)h) * 16u;
                        bg[pt][0] = *reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(in.geo) + goff);
                        bg[pt][NP - 1] = *reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(in.geo + in.geo_stride * 2) + goff);
                        sgv[pt] = in.sigma[pu];
                    } else {
                        const uint32_t voff = ((uint32_t)(4 * h) * (uint32_t)in.pstride + pu) * 4u;
#pragma unroll
                        for (int s = 0; s < IN_KS; s++) {
                            union { half8 v; __half2 q[4]; } u;
#pragma unroll
                            for (int q = 0; q < 4; q++) u.q[q] = *reinterpret_cast<const __half2 *>(reinterpret_cast<const char *>(in.feats + (int64_t)(8 * s + q) * in.pstride) + voff);
                            bx[pt][s][0] = u.v;
                            if constexpr (SPLIT && LMLO) {
#pragma unroll
                                for (int q = 0; q < 4; q++) u.q[q] = *reinterpret_cast<const __half2 *>(reinterpret_cast<const char *>(in.feats_lo + (int64_t)(8 * s + q) * in.pstride) + voff);
                                bx[pt][s][NP - 1] = u.v;
                            } else if constexpr (SPLIT) bx[pt][s][NP - 1] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                        }
                    }
                    const uint32_t ray = in.ray_shift < 0 ? pu : (__umulhi(pu, in.ray_mul) >> in.ray_shift);
                    const uint32_t doff = ray * (uint32_t)(32 * V_KS) + (uint32_t)(16 * h);          // bytes; rays < 2^26
#pragma unroll
                    for (int s = 0; s < V_KS; s++) {
                        bv[pt][s][0] = *reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(in.dirs) + doff + 32 * s);
                        if constexpr (SPLIT) bv[pt][s][NP - 1] = *reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(in.dirs_lo) + doff + 32 * s);
                    }
                    continue;
                }
                const int64_t col = in.src ? (int64_t)cols[pt] : p;
                if (in.keep) kpv[pt] = in.keep[col];
                if constexpr (GEOIN) {
                    bg[pt][0] = in.geo[(col << 1) + h];
                    bg[pt][NP - 1] = in.geo[((in.geo_stride + col) << 1) + h];
                    sgv[pt] = in.sigma[col];
                }
#pragma unroll
                for (int s = 0; s < (GEOIN ? 0 : IN_KS); s++) {
                    union { half8 v; __half2 q[4]; } u;
#pragma unroll
                    for (int q = 0; q < 4; q++) u.q[q] = in.feats[(int64_t)(8 * s + 4 * h + q) * in.pstride + col];   // features 16s+8h+2q, +1
                    bx[pt][s][0] = u.v;
                    if constexpr (SPLIT && LMLO) {
#pragma unroll
                        for (int q = 0; q < 4; q++) u.q[q] = in.feats_lo[(int64_t)(8 * s + 4 * h + q) * in.pstride + col];
                        bx[pt][s][NP - 1] = u.v;
                    } else if constexpr (SPLIT) bx[pt][s][NP - 1] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                // 32-bit division when the index fits (always, for a chunk): the 64-bit one is a ~60-instruction sequence
                const int64_t ray = (p >> 31) == 0 ? (int64_t)((uint32_t)p / (uint32_t)in.s) : p / in.s;
                const int64_t doff = ray * (int64_t)(16 * V_KS);
#pragma unroll
                for (int s = 0; s < V_KS; s++) {
                    bv[pt][s][0] = *reinterpret_cast<const half8 *>(in.dirs + doff + 16 * s + 8 * h);
                    if constexpr (SPLIT) bv[pt][s][NP - 1] = *reinterpret_cast<const half8 *>(in.dirs_lo + doff + 16 * s + 8 * h);
                }
            } else {
                const float *row = in.x + p * in.x_stride;
#pragma unroll
                for (int s = 0; s < IN_KS; s++) {
                    const float4 lo = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h);
                    const float4 hi = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h + 4);
                    if constexpr (SPLIT) split8(lo, hi, bx[pt][s][0], bx[pt][s][NP - 1]);
                    else bx[pt][s][0] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
                }
#pragma unroll
                for (int s = 0; s < V_KS; s++) {
                    const float4 lo = *reinterpret_cast<const float4 *>(row + in.in_ch + 16 * s + 8 * h);
                    const float4 hi = *reinterpret_cast<const float4 *>(row + in.in_ch + 16 * s + 8 * h + 4);
                    if constexpr (SPLIT) split8(lo, hi, bv[pt][s][0], bv[pt][s][NP - 1]);
                    else bv[pt][s][0] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
                }
            }
        }
    };
    // PREFETCH: the operands of the NEXT block iteration are requested right after this iteration's first layer has consumed its own, and
    // arrive while the rest of the network runs (the loads' latency, ~2 us under load, is otherwise exposed once per ~10 us iteration)
    constexpr bool PREFETCH = LM && (SPLIT ? (NRF_SMALL_PREFETCH_SPLIT != 0) : (NRF_SMALL_PREFETCH_F16 != 0));
    half8 bx[PT][IN_KS][NP];
    half8 bv[PT][V_KS][NP];
    half8 bxn[PREFETCH ? PT : 1][IN_KS][NP];
    half8 bvn[PREFETCH ? PT : 1][V_KS][NP];
    half8 bg[PT][NP], bgn[PT][NP];                           // GEOIN: the geo operand fragment, and the sigma that goes with it
    float sgv[PT], sgn[PT];
    (void)bg; (void)bgn; (void)sgv; (void)sgn;
    uint8_t kp[PT], kpn[PT];
    int32_t cols_next[PT];                                   // columns of the block after the one whose operands are being prefetched
#pragma unroll
    for (int pt = 0; pt < PT; pt++) { kp[pt] = 1; kpn[pt] = 1; cols_next[pt] = 0; }
    if constexpr (PREFETCH) {
        if ((int64_t)blockIdx.x < nblocks) {
            int32_t c0[PT];
            load_cols(blockIdx.x, c0);
            if ((int64_t)blockIdx.x + gridDim.x < nblocks) load_cols((int64_t)blockIdx.x + gridDim.x, cols_next);
            load_inputs(blockIdx.x, c0, bx, bv, kp, bg, sgv);
        }
    }
#ifdef NRF_SMALL_TRACE
    unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
#ifdef NRF_SMALL_TRACE
        const unsigned long long titer = __builtin_readcyclecounter();
        unsigned long long tprev = titer;
#endif
        const int64_t p0 = blk * BLOCK_PTS + wave * (32 * PT);
        if constexpr (!PREFETCH) { int32_t c0[PT]; load_cols(blk, c0); load_inputs(blk, c0, bx, bv, kp, bg, sgv); }
        const bool more = blk + gridDim.x < nblocks;
        const half8 *fr = wl;
        // D tiles of a 64-wide hidden layer -> the four k-step operands of the next layer (two buffers: the software pipeline writes the
        // next layer's operands while the current layer still reads its own)
        constexpr bool PIPE = SPLIT ? (NRF_SMALL_PIPE_SPLIT != 0) : (NRF_SMALL_PIPE_F16 != 0);
        constexpr int NBUF = PIPE ? 2 : 1;
        half8 bh[NBUF][PT][4][NP];
        f32x16 acc2[PT][2];
        // item i in 0..3 of tile t: point tile i >> 1, register half i & 1 -> operand fragment 2t + (i & 1) of buffer `buf`
        auto conv_item = [&](int buf, int t, int i) {
            const int pt = i >> 1, sh = i & 1;
            if constexpr (SPLIT) tile_to_frag2<true>(acc2[pt][t], sh, bh[buf][pt][2 * t + sh][0], bh[buf][pt][2 * t + sh][NP - 1]);
            else bh[buf][pt][2 * t + sh][0] = tile_to_frag<true>(acc2[pt][t], sh);
        };
        // The same conversion in HALF-QUAD units for the fine-grained schedule (split mode): quad q of item i = values 4q..4q+3 of the 8-register half tile.
        // Part 0: four ReLUs and the two packed roundings to fp16 (hi); part 1: the four residuals lo = f16(v - hi).  Two value pairs advance together so that
        // no instruction reads the result of the one right before it (a v_cvt_pk followed by the v_fma_mix that reads it costs an s_nop).
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        float qm[4];
        uint32_t qc[2];
        auto conv_half = [&](int buf, int t, int i, int q, int part, bool need_lo) __attribute__((always_inline)) {
            (void)qm; (void)qc;
            if constexpr (SPLIT) {
                const int pt = i >> 1, sh = i & 1;
#if NRF_SPLIT_RTZ
                // (see split_pair_relu) part 0: the two truncating packed conversions; part 1: four clamped residuals and the two packed ReLUs of hi
                typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                const int b0 = 8 * sh + 4 * q;
                if (part == 0) {
                    qc[0] = __builtin_bit_cast(uint32_t, (fp16x2)__builtin_amdgcn_cvt_pkrtz(acc2[pt][t][b0], acc2[pt][t][b0 + 1]));
                    qc[1] = __builtin_bit_cast(uint32_t, (fp16x2)__builtin_amdgcn_cvt_pkrtz(acc2[pt][t][b0 + 2], acc2[pt][t][b0 + 3]));
                } else {
                    uint32_t l0, l1;
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(l0) : "v"(qc[0]), "v"(acc2[pt][t][b0]));
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(l1) : "v"(qc[1]), "v"(acc2[pt][t][b0 + 2]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(l0) : "v"(qc[0]), "v"(acc2[pt][t][b0 + 1]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(l1) : "v"(qc[1]), "v"(acc2[pt][t][b0 + 3]));
                    const uint32_t h0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2, qc[0]), f16x2{0, 0}));
                    const uint32_t h1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2, qc[1]), f16x2{0, 0}));
                    u32x4 hv = __builtin_bit_cast(u32x4, bh[buf][pt][2 * t + sh][0]), lv = __builtin_bit_cast(u32x4, bh[buf][pt][2 * t + sh][NP - 1]);
                    hv[2 * q] = h0; hv[2 * q + 1] = h1; lv[2 * q] = l0; lv[2 * q + 1] = l1;
                    bh[buf][pt][2 * t + sh][0] = __builtin_bit_cast(half8, hv); bh[buf][pt][2 * t + sh][NP - 1] = __builtin_bit_cast(half8, lv);
                }
                return;
#endif
                if (part == 0) {
#pragma unroll
                    for (int e = 0; e < 4; e++) qm[e] = fmaxf(acc2[pt][t][8 * sh + 4 * q + e], 0.0f);
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(qc[0]) : "v"(qm[0]), "v"(qm[1]));
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(qc[1]) : "v"(qm[2]), "v"(qm[3]));
                } else if (!need_lo) {
                    // (product budget builds) the consuming layer drops its Wh.xl product: the residuals are not formed
                    u32x4 hv = __builtin_bit_cast(u32x4, bh[buf][pt][2 * t + sh][0]);
                    hv[2 * q] = qc[0]; hv[2 * q + 1] = qc[1];
                    bh[buf][pt][2 * t + sh][0] = __builtin_bit_cast(half8, hv);
                } else {
                    uint32_t l0, l1;
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(qc[0]), "v"(qm[0]));
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(qc[1]), "v"(qm[2]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(qc[0]), "v"(qm[1]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(qc[1]), "v"(qm[3]));
                    u32x4 hv = __builtin_bit_cast(u32x4, bh[buf][pt][2 * t + sh][0]), lv = __builtin_bit_cast(u32x4, bh[buf][pt][2 * t + sh][NP - 1]);
                    hv[2 * q] = qc[0]; hv[2 * q + 1] = qc[1]; lv[2 * q] = l0; lv[2 * q + 1] = l1;
                    bh[buf][pt][2 * t + sh][0] = __builtin_bit_cast(half8, hv); bh[buf][pt][2 * t + sh][NP - 1] = __builtin_bit_cast(half8, lv);
                }
            }
        };
        static_assert(PT == 2, "the conversion schedule below is written for two point tiles per wave");
        auto hidden_to_b = [&](int buf) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) conv_item(buf, t, i);
        };
        // Software pipeline (PIPE): while the matrix pipe works on tile 1 of a layer, tile 0 -- already complete -- is converted into the next
        // layer's first two operand fragments; tile 1 is converted during the first two k-steps of the next layer, which only need those.
        //   job of a layer with input buffer `bi` and KS k-steps:  mt == 0, ks < 2 : previous layer's tile 1 -> bh[bi][.][2..3]
        //                                                           mt == 1         : this layer's tile 0      -> bh[bi ^ 1][.][0..1]
        // FINE (split mode): the items of a step are cut into half-quad units (4 per item) and dealt out over the step's ng matrix instructions, unit u after
        // instruction floor(u * ng / units); otherwise the step's items follow its last matrix instruction as one block.
        constexpr bool FINE = SPLIT && PIPE && (NRF_SMALL_FINE != 0) && V_KS == 1;      // the 64-wide direction encodings (SH degree 8) are at the register limit already
        // LoPrev / LoNext (std::bool_constant): does the layer that consumes the previous layer's tile 1 (= this layer) / this layer's tile 0 (= the next layer) use the
        // (lo) halves?  Types, not values: every conv_half call site then carries a literal, and the register arrays stay registers (a run-time flag here sent them to scratch)
        auto make_job = [&](int bi, bool conv_prev_tile1, bool conv_this_tile0, int ks_count, auto lo_prev_t, auto lo_next_t) __attribute__((always_inline)) {
            return [=, &conv_item, &conv_half](int mt, int ks, int g, int ng) __attribute__((always_inline)) {
                (void)conv_item; (void)conv_half;           // not referenced when PIPE is off for this instantiation
                constexpr bool LoPrev = decltype(lo_prev_t)::value, LoNext = decltype(lo_next_t)::value;
                if constexpr (PIPE) {
                    // items of this step: [i0, i1) of tile tt into buffer bb
                    int i0 = 0, i1 = 0, tt = 0, bb = bi;
                    bool is_prev = false;
                    if (conv_prev_tile1 && mt == 0 && ks < 2) { i0 = 2 * ks; i1 = 2 * ks + 2; tt = 1; bb = bi; is_prev = true; }
                    if (conv_this_tile0 && mt == 1) {
                        const int per = (4 + ks_count - 1) / ks_count;
                        i0 = ks * per; i1 = (ks + 1) * per < 4 ? (ks + 1) * per : 4; tt = 0; bb = bi ^ 1; is_prev = false;
                    }
                    if constexpr (FINE) {
                        const int units = (i1 - i0) * 4;
#pragma unroll
                        for (int u = 0; u < 16; u++) {          // fixed bound: every index below must fold to a constant (register arrays)
                            if (u < units && (u * ng) / units == g) {
                                if constexpr (LoPrev && LoNext) conv_half(bb, tt, i0 + (u >> 2), (u >> 1) & 1, u & 1, true);
                                else if (is_prev) { if constexpr (LoPrev) conv_half(bb, tt, i0 + (u >> 2), (u >> 1) & 1, u & 1, true); else conv_half(bb, tt, i0 + (u >> 2), (u >> 1) & 1, u & 1, false); }
                                else { if constexpr (LoNext) conv_half(bb, tt, i0 + (u >> 2), (u >> 1) & 1, u & 1, true); else conv_half(bb, tt, i0 + (u >> 2), (u >> 1) & 1, u & 1, false); }
                            }
                        }
                    } else {
                        if (g == ng - 1) {
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (i >= i0 && i < i1) conv_item(bb, tt, i);
                        }
                    }
                }
            };
        };
        // ---- sigma net ----
        if constexpr (GEOIN) fr += Plan::sigma_total() * 64 * NP;          // colour net only: its fragments follow the sigma net's in the image
        half8 pre[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) pre[q] = fr[q * 64 + lane];
        f32x16 sig[PT][1];
        auto prefetch_next = [&]() {
            if constexpr (PREFETCH) {
                if (more) {
                    load_inputs(blk + gridDim.x, cols_next, bxn, bvn, kpn, bgn, sgn);             // cols_next arrived an iteration ago
                    if (blk + 2 * (int64_t)gridDim.x < nblocks) load_cols(blk + 2 * (int64_t)gridDim.x, cols_next);
                }
            }
        };
        if constexpr (GEOIN) {
        } else if constexpr (NL == 1) {
            gemm_layer<1, IN_KS, NP, IN_LO>(fr, lane, bx, sig, pre); fr += Plan::sigma_frags(0) * 64 * NP;
        } else {
            gemm_layer<2, IN_KS, NP, IN_LO, false, FINE, small_drop_of(0)>(fr, lane, bx, acc2, pre, make_job(1, false, true, IN_KS, std::true_type{}, std::bool_constant<!(small_drop_of(1) & 2)>{})); fr += Plan::sigma_frags(0) * 64 * NP;     // tile 0 -> bh[0]
            prefetch_next();
            NRF_TSTAMP(0);
            static_assert(NL <= 3, "sigma net: at most three layers");
            if constexpr (NL == 3) {
                if constexpr (!PIPE) hidden_to_b(0);
                gemm_layer<2, 4, NP, SPLIT, false, FINE, small_drop_of(1)>(fr, lane, bh[0], acc2, pre, make_job(0, true, true, 4, std::bool_constant<!(small_drop_of(1) & 2)>{}, std::bool_constant<!(small_drop_of(2) & 2)>{}));
                fr += Plan::sigma_frags(1) * 64 * NP;
                NRF_TSTAMP(1);
            }
            {
                constexpr int l = NL - 1;
                const int bi = PIPE ? ((l - 1) & 1) : 0;
                if constexpr (!PIPE) hidden_to_b(0);
                gemm_layer<1, 4, NP, SPLIT, false, FINE, small_drop_of(l)>(fr, lane, bh[bi], sig, pre, make_job(bi, true, false, 4, std::bool_constant<!(small_drop_of(l) & 2)>{}, std::true_type{}));
                fr += Plan::sigma_frags(l) * 64 * NP;
                NRF_TSTAMP(l);
            }
        }
        // ---- colour net: k-steps = [views..., geo] ----
        half8 bc[PT][V_KS + 1][NP];
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
#pragma unroll
            for (int s = 0; s < V_KS; s++)
#pragma unroll
                for (int q = 0; q < NP; q++) bc[pt][s][q] = bv[pt][s][q];
            // rows 0..15 of the sigma tile: sigma (zero weight) + geo
            if constexpr (GEOIN) { bc[pt][V_KS][0] = bg[pt][0]; bc[pt][V_KS][NP - 1] = bg[pt][NP - 1]; }
            else if constexpr (SPLIT) tile_to_frag2<false>(sig[pt][0], 0, bc[pt][V_KS][0], bc[pt][V_KS][NP - 1]);
            else bc[pt][V_KS][0] = tile_to_frag<false>(sig[pt][0], 0);
        }
        NRF_TSTAMP(3);
        f32x16 rgb[PT][1];
        if constexpr (NLC == 1) {
            gemm_layer<1, V_KS + 1, NP, SPLIT, true, false, small_drop_of(NL)>(fr, lane, bc, rgb, pre);
            if constexpr (GEOIN) prefetch_next();
        } else {
            gemm_layer<2, V_KS + 1, NP, SPLIT, false, FINE, small_drop_of(NL)>(fr, lane, bc, acc2, pre, make_job(1, false, true, V_KS + 1, std::true_type{}, std::bool_constant<!(small_drop_of(NL + 1) & 2)>{})); fr += Plan::color_frags(0) * 64 * NP;
            if constexpr (GEOIN) prefetch_next();          // the colour-only kernel's operands of the next iteration, behind its first layer
            NRF_TSTAMP(4);
            static_assert(NLC <= 4, "colour net: at most four layers");
            if constexpr (NLC >= 3) {
                if constexpr (!PIPE) hidden_to_b(0);
                gemm_layer<2, 4, NP, SPLIT, false, FINE, small_drop_of(NL + 1)>(fr, lane, bh[0], acc2, pre, make_job(0, true, true, 4, std::bool_constant<!(small_drop_of(NL + 1) & 2)>{}, std::bool_constant<!(small_drop_of(NL + 2) & 2)>{}));
                fr += Plan::color_frags(1) * 64 * NP;
                NRF_TSTAMP(5);
            }
            if constexpr (NLC >= 4) {
                if constexpr (!PIPE) hidden_to_b(0);
                gemm_layer<2, 4, NP, SPLIT, false, FINE, small_drop_of(NL + 2)>(fr, lane, bh[PIPE ? 1 : 0], acc2, pre, make_job(PIPE ? 1 : 0, true, true, 4, std::bool_constant<!(small_drop_of(NL + 2) & 2)>{}, std::bool_constant<!(small_drop_of(NL + 3) & 2)>{}));
                fr += Plan::color_frags(2) * 64 * NP;
                NRF_TSTAMP(6);
            }
            {
                constexpr int l = NLC - 1;
                const int bi = PIPE ? ((l - 1) & 1) : 0;
                if constexpr (!PIPE) hidden_to_b(0);
                gemm_layer<1, 4, NP, SPLIT, true, FINE, small_drop_of(NL + l)>(fr, lane, bh[bi], rgb, pre, make_job(bi, true, false, 4, std::bool_constant<!(small_drop_of(NL + l) & 2)>{}, std::true_type{}));
                fr += Plan::color_frags(l) * 64 * NP;
                NRF_TSTAMP(4 + l);
            }
        }
        // ---- out = (rgb, sigma): rows 0..2 of the colour tile and row 0 of the sigma tile live in registers 0..2 / 0 of lane-half 0 ----
        if (h == 0) {
#pragma unroll
            for (int pt = 0; pt < PT; pt++) {
                const int64_t p = p0 + pt * 32 + r;
                if (p < npts) {
                    float sg;
                    if constexpr (GEOIN) sg = sgv[pt]; else if constexpr (SPLIT) sg = sig[pt][0][0] * inv_sigma; else sg = sig[pt][0][0];
                    if constexpr (LM) { if (!kp[pt]) sg = 0.0f; }                                    // the embedder's keep mask (NeRFRenderer.h:187-188), fetched with the operands
                    float c0 = rgb[pt][0][0], c1 = rgb[pt][0][1], c2 = rgb[pt][0][2];
                    if constexpr (SPLIT) { c0 *= inv_rgb; c1 *= inv_rgb; c2 *= inv_rgb; }
                    if (out_stride == 4) *reinterpret_cast<float4 *>(out + p * 4) = float4{c0, c1, c2, sg};
                    else { float *o = out + p * out_stride; o[0] = c0; o[1] = c1; o[2] = c2; o[3] = sg; }
                }
            }
        }
        if constexpr (PREFETCH) {
            if (more) {
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
#pragma unroll
                    for (int s = 0; s < IN_KS; s++)
#pragma unroll
                        for (int q = 0; q < NP; q++) bx[pt][s][q] = bxn[pt][s][q];
#pragma unroll
                    for (int s = 0; s < V_KS; s++)
#pragma unroll
                        for (int q = 0; q < NP; q++) bv[pt][s][q] = bvn[pt][s][q];
                    kp[pt] = kpn[pt];
                    if constexpr (GEOIN) { bg[pt][0] = bgn[pt][0]; bg[pt][NP - 1] = bgn[pt][NP - 1]; sgv[pt] = sgn[pt]; }
                }
            }
        }
#ifdef NRF_SMALL_TRACE
        NRF_TSTAMP(8);
        tr[9] += __builtin_readcyclecounter() - titer; tr[10] += 1;
#endif
    }
#ifdef NRF_SMALL_TRACE
    if (threadIdx.x == 0 && SPLIT && LM) for (int i = 0; i < 12; i++) g_small_trace[blockIdx.x * 12 + i] += tr[i];      // no other code reads this buffer
#endif
}

#ifdef NRF_SMALL_TRACE
extern "C" NRF_API int nrf_dbg_small_trace(unsigned long long *host_out, int reset)
{
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_small_trace), sizeof(unsigned long long) * 256 * 12) != hipSuccess) return NRF_ERR_HIP;
    if (reset) { static unsigned long long z[256 * 12]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_small_trace), z, sizeof(z)) != hipSuccess) return NRF_ERR_HIP; }
    return NRF_OK;
}
#endif

// Tried and measured in round 2 (same box, A/B builds of this file, split mode, ms of this kernel per 800x800 frame's fine pass):
//   * NRF_SMALL_FINE (kept): the conversions dealt out over the matrix instructions of a step (4-6 vector instructions per MFMA gap, a fence after every MFMA,
//     value pairs interleaved so that no v_fma_mix reads the v_cvt_pk right before it) instead of one block per step: 10.88 vs 10.94, 10.61 vs 10.71 -- under 1 %.
//   * NRF_SMALL_STAGGER = 1, 2, 3 (waves 4-7 start 8 / 16 / 24 k cycles late, so that SIMD partners are not in the same phase): 11.06 / 11.15 / 11.05 vs 11.13.
//   Together with the first round's results (scheduler strategies, 16x16x32, register-resident weights, 11 % of the matrix work moved to the vector ALUs: all
//   within 1 %) the kernel's time does not respond to how its instructions are ordered or split between the pipes: per wave and iteration it issues ~6.9 k cycles
//   of instructions (MFMA 1.9 k, conversions 3.4 k, address arithmetic 1.1 k), two waves per SIMD make 13.9 k of the 24.9 k cycles an iteration takes, the matrix
//   pipe is busy 14.1 k.
//   * Ablation builds (results wrong on purpose; full kernel 11.1-11.8 ms on that box): one instruction instead of the ten conversions of a half-quad 8.9 ms;
//     one A-fragment ds_read_b128 per layer instead of one per step 11.5 ms; one matrix product per step instead of three (40 of 116 MFMAs per tile) 5.2 ms.
//     The time follows the SUM of the matrix and the vector work, not their maximum; LDS reads are free.
// Tried and measured (same box): the last colour layer (64 -> 3, one 32-row tile with 3 useful rows = 12 of the split mode's 116 matrix instructions per tile)
// moved to the vector ALUs in fp32, straight from the last hidden layer's D tiles.  Split mode: 13.63 vs 13.65 ms per frame -- nothing, although
// 11 % of the matrix work is gone: the chip is holding its clock down under this load, so cycles taken off the matrix pipe and put on the
// vector ALUs come back as clock, not as time.  Plain fp16 mode: 7.1 vs 5.5 ms (194 VGPRs instead of 146 cost the third wave per SIMD).  Not kept.

// ---------------------------------------------------------------------------------------------------
// weight image
// ---------------------------------------------------------------------------------------------------
struct Packer {
    std::vector<_Float16> img;
    // one layer: W [out][in] row-major at `w`; krow(ks, h, j) -> input index or -1 (zero)
    bool split = false;     // each 1-KB fragment followed by the fragment of the rounding residuals w - f16(w)
    template <class KMap>
    void layer(const float *w, int in, int out, int mtiles, int ksteps, KMap kmap)
    {
        for (int mt = 0; mt < mtiles; mt++)
            for (int ks = 0; ks < ksteps; ks++)
                for (int part = 0; part < (split ? 2 : 1); part++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int row = mt * 32 + (lane & 31);
                            const int k = kmap(ks, lane >> 5, j);
                            float v = 0.0f;
                            if (row < out && k >= 0 && k < in) v = w[(size_t)row * in + k];
                            const _Float16 hv = (_Float16)v;
                            img.push_back(part == 0 ? hv : (_Float16)(v - (float)hv));
                        }
    }
};

static bool small_mfma_supported(const nrf_mlp_small_desc &d)
{
    return d.input_ch == 32 && (d.input_ch_views == 16 || d.input_ch_views == 64) && d.hidden_dim == 64 && d.hidden_dim_color == 64 &&
           d.geo_feat_dim >= 0 && d.geo_feat_dim <= 15 && d.num_layers >= 2 && d.num_layers <= 3 && d.num_layers_color >= 2 && d.num_layers_color <= 4;
}

static void pack_small(const nrf_mlp_small_desc &d, const std::vector<float> &hp, Packer &pk)
{
    const int in_ks = d.input_ch / 16, v_ks = d.input_ch_views / 16;
    auto natural = [](int ks, int h, int j) { return 16 * ks + 8 * h + j; };                       // operand loaded from memory
    auto chained = [](int ks, int h, int j) { return 32 * (ks >> 1) + perm_row(ks & 1, h, j); };  // operand = previous D tiles
    size_t off = 0;
    for (int l = 0; l < d.num_layers; l++) {
        const int in = l == 0 ? d.input_ch : d.hidden_dim, out = l == d.num_layers - 1 ? 1 + d.geo_feat_dim : d.hidden_dim;
        const int mt = l == d.num_layers - 1 ? 1 : 2;
        if (l == 0) pk.layer(hp.data() + off, in, out, mt, in_ks, natural);
        else pk.layer(hp.data() + off, in, out, mt, 4, chained);
        off += (size_t)in * out;
    }
    for (int l = 0; l < d.num_layers_color; l++) {
        const int in = l == 0 ? d.input_ch_views + d.geo_feat_dim : d.hidden_dim_color, out = l == d.num_layers_color - 1 ? 3 : d.hidden_dim_color;
        const int mt = l == d.num_layers_color - 1 ? 1 : 2;
        if (l == 0) {
            const int V = d.input_ch_views, G = d.geo_feat_dim;
            // k-steps [0, v_ks): view features in natural order; k-step v_ks: rows 0..15 of the sigma tile (row 0 = sigma -> zero weight)
            auto cmap = [=](int ks, int h, int j) {
                if (ks < v_ks) return 16 * ks + 8 * h + j;
                const int row = perm_row(0, h, j);
                return (row >= 1 && row <= G) ? V + row - 1 : -1;
            };
            pk.layer(hp.data() + off, in, out, mt, v_ks + 1, cmap);
        } else pk.layer(hp.data() + off, in, out, mt, 4, chained);
        off += (size_t)in * out;
    }
}

bool mlp_small_images_host(const nrf_mlp_small_desc &d, const std::vector<float> &hp, std::vector<uint8_t> &f16_img, std::vector<uint8_t> &split_img)
{
    if (!small_mfma_supported(d)) return false;
    Packer pk, pk2;
    pk2.split = true;
    pack_small(d, hp, pk);
    pack_small(d, hp, pk2);
    f16_img.assign(reinterpret_cast<const uint8_t *>(pk.img.data()), reinterpret_cast<const uint8_t *>(pk.img.data() + pk.img.size()));
    split_img.assign(reinterpret_cast<const uint8_t *>(pk2.img.data()), reinterpret_cast<const uint8_t *>(pk2.img.data() + pk2.img.size()));
    return true;
}

int mlp_small_pack_f16(nrf_mlp *m, const std::vector<float> &hp)
{
    const auto &d = m->small;
    std::vector<uint8_t> a, b;
    if (!mlp_small_images_host(d, hp, a, b)) return NRF_OK;      // the matrix-core precisions then report NRF_ERR_UNSUPPORTED at forward time
    // re-pack after nrf_mlp_set_params (host path): same sizes, so the device images are overwritten in place
    auto upload = [](void *&dst, size_t &have, const std::vector<uint8_t> &img) -> int {
        const size_t bytes = img.size();
        if (dst && have != bytes) { (void)hipFree(dst); dst = nullptr; }
        if (!dst) NRF_HIP(hipMalloc(&dst, bytes));
        have = bytes;
        NRF_HIP(hipMemcpy(dst, img.data(), bytes, hipMemcpyHostToDevice));
        return NRF_OK;
    };
    NRF_TRY(upload(m->d_packed_f16, m->packed_f16_bytes, a));
    NRF_TRY(upload(m->d_packed_split, m->packed_split_bytes, b));
    return mlp_small_pack_bwd(m, hp);
}

template <int V_KS, int NL, int NLC>
static int launch_small(const nrf_mlp *m, const SmallInput &in, bool lm, bool split, int64_t p, float *out, int os, hipStream_t st)
{
    using Plan = SmallPlan<2, V_KS, NL, NLC>;
    const size_t lds = (size_t)Plan::total() * 1024 * (split ? 2 : 1);
    const size_t have = split ? m->packed_split_bytes : m->packed_f16_bytes;
    if (lds != have) { set_error("internal: packed weight image is %zu bytes, kernel expects %zu", have, lds); return NRF_ERR_INVALID_ARG; }
    const int64_t nblocks = ceil_div(p, block_pts_of(split));
    // persistent: 256 CUs x 3 resident 4-wave workgroups (146 VGPRs, 40-46 KB LDS), or x 1 resident 8-wave workgroup with the 80-92 KB split image
    const int64_t cap = split ? 256 : 768;
    const unsigned grid = (unsigned)(nblocks < cap ? nblocks : cap);
    const half8 *img = reinterpret_cast<const half8 *>(split ? m->d_packed_split : m->d_packed_f16);
#define NRF_GO(LM_, SP_, LO_, ...)                                                                                                            \
    do {                                                                                                                                      \
        auto kfn = k_mlp_small_mfma<2, V_KS, NL, NLC, LM_, SP_, LO_, ##__VA_ARGS__>;                                                          \
        if (lds > 64 * 1024) NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * waves_of(SP_)), lds, st, p, in, img, out, os);                                          \
    } while (0)
    // 32-bit byte offsets: points, feature planes (4 levels apart), geo fragments and the per-ray direction rows all below 4 GB
    const bool a32 = lm && split && !in.src && (p >> 26) == 0 && (in.pstride >> 27) == 0 && (in.geo_stride >> 26) == 0 && in.ray_mul != 0 &&
                     (((uint64_t)(p / (in.s > 0 ? in.s : 1)) + 1) * (uint64_t)(32 * V_KS)) >> 32 == 0;
    if (in.geo) {
        if (!lm || !split) { set_error("internal: the colour-only NeRFSmall kernel is split precision, level-major"); return NRF_ERR_INVALID_ARG; }
        if (a32) NRF_GO(true, true, false, true, true); else NRF_GO(true, true, false, true);
    } else if (lm) {
        if (split) {
            if (in.feats_lo) { if (a32) NRF_GO(true, true, true, false, true); else NRF_GO(true, true, true); }
            else { if (a32) NRF_GO(true, true, false, false, true); else NRF_GO(true, true, false); }
        } else NRF_GO(true, false, false);
    }
    else { if (split) NRF_GO(false, true, false); else NRF_GO(false, false, false); }
#undef NRF_GO
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

static int dispatch_small(const nrf_mlp *m, const SmallInput &in, bool lm, bool split, int64_t p, float *out, int os, hipStream_t st);

int mlp_small_mfma_available(const nrf_mlp *m) { return m && m->family == MLP_SMALL && m->d_packed_f16 != nullptr && m->d_packed_split != nullptr; }

// renderer fast path: level-major fp16 features + per-ray fp16 direction features + keep mask -> raw [p,4] (sigma masked).
// dirs_lo != NULL selects the split-precision kernel (NRF_PREC_F16_SPLIT).
int mlp_small_forward_mfma_lm(const nrf_mlp *m, const __half2 *feats, const __half2 *feats_lo, int64_t pstride, const __half *dirs, const __half *dirs_lo, int s,
                              const uint8_t *keep, int64_t p, float *out, hipStream_t st, const int32_t *src)
{
    if (!mlp_small_mfma_available(m)) { set_error("internal: matrix-core NeRFSmall image missing"); return NRF_ERR_UNSUPPORTED; }
    ProfScope prof(NRF_PROF_MLP, st);
    SmallInput in{nullptr, 0, m->small.input_ch, feats, pstride, dirs, s, keep, dirs_lo, dirs_lo ? feats_lo : nullptr, src, nullptr, 0, nullptr};
    return dispatch_small(m, in, true, dirs_lo != nullptr, p, out, 4, st);
}

// The colour net alone, split precision: point i takes the sigma net's output from the exact coarse kernel (mlp_small_sigma_f32_lm with geo) -- column i of the geo
// planes, sigma[i] -- and the direction features of ray i / s.  out [p, 4] = (rgb, sigma).
int mlp_small_color_from_geo_lm(const nrf_mlp *m, const void *geo, int64_t geo_stride, const float *sigma, const __half *dirs, const __half *dirs_lo, int s,
                                const uint8_t *keep, int64_t p, float *out, hipStream_t st)
{
    if (!mlp_small_mfma_available(m)) { set_error("internal: matrix-core NeRFSmall image missing"); return NRF_ERR_UNSUPPORTED; }
    if (!geo || !sigma || !dirs_lo) { set_error("internal: colour-only pass without geo planes / sigma / split direction features"); return NRF_ERR_INVALID_ARG; }
    ProfScope prof(NRF_PROF_MLP_COLOUR, st);
    SmallInput in{nullptr, 0, m->small.input_ch, nullptr, 0, dirs, s, keep, dirs_lo, nullptr, nullptr, static_cast<const half8 *>(geo), geo_stride, sigma};
    return dispatch_small(m, in, true, true, p, out, 4, st);
}

int mlp_small_forward_mfma(const nrf_mlp *m, const float *x, int xs, int64_t p, int split, float *out, int os, hipStream_t st)
{
    const auto &d = m->small;
    if (!m->d_packed_f16) {
        set_error("NRF_PREC_F16_MFMA / NRF_PREC_F16_SPLIT: NeRFSmall shape (in %d, views %d, %dx%d, geo %d, colour %dx%d) is outside the built matrix-core family; use NRF_PREC_F32",
                  d.input_ch, d.input_ch_views, d.num_layers, d.hidden_dim, d.geo_feat_dim, d.num_layers_color, d.hidden_dim_color);
        return NRF_ERR_UNSUPPORTED;
    }
    if ((xs % 4) != 0 || (reinterpret_cast<uintptr_t>(x) & 15)) { set_error("NRF_PREC_F16_MFMA: input rows must be 16-byte aligned"); return NRF_ERR_INVALID_ARG; }
    SmallInput in{x, xs, d.input_ch, nullptr, 0, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr};
    return dispatch_small(m, in, false, split != 0, p, out, os, st);
}

static int dispatch_small(const nrf_mlp *m, const SmallInput &in_, bool lm, bool split, int64_t p, float *out, int os, hipStream_t st)
{
    const auto &d = m->small;
    SmallInput in = in_;
    in.ray_mul = 0; in.ray_shift = 0;
    in.scales = m->d_scales;
    if (lm && in.s >= 1) {
        // ray = floor(p / s) for p < 2^31 as (p * M) >> (31 + L), L = ceil(log2 s), M = ceil(2^(31 + L) / s) < 2^32 (Granlund-Montgomery round-up); the kernel takes the
        // high word of the product and shifts by L - 1 (s = 1: ray = p, flagged by a negative shift)
        int L = 0;
        while ((1 << L) < in.s) L++;
        if (L == 0) { in.ray_mul = 1; in.ray_shift = -1; }
        else { in.ray_mul = (uint32_t)((((uint64_t)1 << (31 + L)) + (uint64_t)in.s - 1) / (uint64_t)in.s); in.ray_shift = L - 1; }
    }
    const int v = d.input_ch_views / 16;
#define NRF_CASE(V, NL, NLC) if (v == V && d.num_layers == NL && d.num_layers_color == NLC) return launch_small<V, NL, NLC>(m, in, lm, split, p, out, os, st);
    NRF_CASE(1, 3, 4) NRF_CASE(1, 3, 3) NRF_CASE(1, 3, 2) NRF_CASE(1, 2, 4) NRF_CASE(1, 2, 3) NRF_CASE(1, 2, 2)
    NRF_CASE(4, 3, 4) NRF_CASE(4, 3, 3) NRF_CASE(4, 3, 2) NRF_CASE(4, 2, 4) NRF_CASE(4, 2, 3) NRF_CASE(4, 2, 2)
#undef NRF_CASE
    set_error("internal: no matrix-core instantiation for this NeRFSmall shape");
    return NRF_ERR_UNSUPPORTED;
}

// classic NeRF matrix-core path: see mlp_nerf_mfma.hip
}  // namespace nrf
