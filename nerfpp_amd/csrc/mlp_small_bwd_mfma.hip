// mlp_small_bwd_mfma.hip -- backward of NeRFSmallImpl::forward (NeRF.cpp:322-412; the reference gets it from autograd,
// NeRFExecutor.h:923 loss.backward()) on the gfx950 matrix cores: ONE persistent kernel does forward, the gradient chain
// and the weight gradients of a 64-point wave tile without the activations ever being laid out row-major.
//
// The forward runs transposed, H_{l+1}^T = W_{l+1} . H_l^T (mlp_small_mfma.hip): a D tile has the point on the lane and the
// neurons in the registers, and that is the next layer's B operand.  The gradient chain has the same shape with the
// transposed weights,   G_l^T [in x points] = W_{l+1}^T . G_{l+1}^T,   so it reuses the trick unchanged: a second weight
// image holds the W^T fragments, the ReLU mask is applied to the D tile (the forward's operand fragment of that layer, kept
// in a scratch buffer, has exactly the D tile's register <-> neuron mapping), and the tile becomes the next operand.
//
// The weight gradient dW_l[o][i] = sum_points G[o][pt] H[i][pt] sums over the POINT index, which both operands carry on the
// lane -- the wrong axis for an MFMA, whose reduction index lives inside a lane's 8-element fragment.  The matrix core does
// the transposition itself: multiplying a fragment (as the A operand, M = points) by a 0/1 selector matrix (B, picks one
// neuron per column) is exact -- one non-zero product per sum -- and leaves D with the NEURON on the lane and the points in
// the registers; its two register halves are, as before, ready-made operand fragments, now with k = points.  Both factors
// go through the same map, so the permutation of points inside a k-step cancels.  dW tiles of the two point tiles of a wave
// are accumulated in the same MFMA accumulator and then added into an fp32 copy of the whole gradient blob in LDS
// (ds_add_f32, 70 KB); every workgroup adds its copy to global memory once, at the end.
//
// Arithmetic: fp16 operands, fp32 accumulation.  Gradients are multiplied by a power of two S chosen from max|g_out| (read
// on the device, no host sync) so that they sit in the middle of the fp16 range, and divided out in fp32 at the end.
#include "mlp.h"

namespace nrf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#ifdef NRF_BWD_TRACE
// diagnostic build only (tools/scratch/bwd_trace.py): cycle stamps of wave 0 of every workgroup, summed per section (names in the script)
__device__ unsigned long long g_bwd_trace[256 * 16];
#define NRF_BSTAMP(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tr[i] += t__ - tprev; tprev = t__; } while (0)
#else
#define NRF_BSTAMP(i) do { } while (0)
#endif

constexpr int BW = 4;                   // waves per workgroup: one per SIMD, 512 registers each (the dW accumulators alone are 320)
constexpr int BPT = 1;                  // 32-point tiles per wave (two would need 180 working registers on top of the 320 accumulators)
constexpr int BW_BLOCK_PTS = 32 * BPT * BW;
constexpr int IN_KS = 2, V = 16, GEO = 15;      // 32 hash features, 16 direction features, 15 geometry features (NeRF.h:215)

__host__ __device__ inline int prow(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }   // see mlp_small_mfma.hip

template <int NL, int NLC>
struct BwdPlan {
    // forward image = the one the forward kernel uses (pack_small order, mlp_small_mfma.hip); the last colour layer is not needed
    static constexpr int sigma_frags(int l) { return (l == NL - 1 ? 1 : 2) * (l == 0 ? IN_KS : 4); }
    static constexpr int color_frags(int l) { return (l == NLC - 1 ? 1 : 2) * (l == 0 ? 2 : 4); }
    static constexpr int fwd_frags()
    {
        int t = 0;
        for (int l = 0; l < NL; l++) t += sigma_frags(l);
        for (int l = 0; l < NLC; l++) t += color_frags(l);
        return t;
    }
    // backward image, consumption order: colour net last -> first, sigma net last -> first
    static constexpr int bwd_color_frags(int l) { return l == NLC - 1 ? 2 : (l == 0 ? 4 : 8); }
    static constexpr int bwd_sigma_frags(int l) { return l == NL - 1 ? 2 : (l == 0 ? 4 : 8); }
    static constexpr int bwd_frags()
    {
        int t = 0;
        for (int l = 0; l < NL; l++) t += bwd_sigma_frags(l);
        for (int l = 0; l < NLC; l++) t += bwd_color_frags(l);
        return t;
    }
    // layer-input fragments kept per 32-point tile: x, h_1..h_{NL-1}, [views, geo], c_1..c_{NLC-1}
    static constexpr int h_sigma(int l) { return l == 0 ? 0 : 2 + 4 * (l - 1); }
    static constexpr int h_color(int l) { return 2 + 4 * (NL - 1) + (l == 0 ? 0 : 2 + 4 * (l - 1)); }
    static constexpr int h_frags() { return 2 + 4 * (NL - 1) + 2 + 4 * (NLC - 1); }
};

struct LayerOffs { int s[4]; int c[4]; };       // offsets (floats) of every W in the parameter blob

// Alternative input (feats != NULL): the level-major fp16 hash features of nrf_hash_encode_lm_f16 ([16 levels][pstride] half2, already offset to this launch's
// first point) and per-RAY fp16 direction features [n][16]; point p of the launch belongs to ray (p_base + p) / s.  No [p, 48] fp32 row is ever formed.
// `src` (optional): point q of the whole call reads feature COLUMN src[q] of a table that is then NOT offset per launch -- the merge map of a renderer's feature-reusing
// fine pass (nrf_renderer_last_features): the training backward reads the features the forward render has just encoded instead of encoding the points again.
struct LmInput { const __half2 *feats; int64_t pstride; const __half *dirs; int s; int64_t p_base; const int32_t *src; };

__device__ __forceinline__ f32x16 mfma(const half8 &a, const half8 &b, const f32x16 &c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <bool RELU>
__device__ __forceinline__ half8 to_frag(const f32x16 &acc, int s)
{
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (_Float16)acc[8 * s + j];
    if (RELU) r = __builtin_elementwise_max(r, half8{0, 0, 0, 0, 0, 0, 0, 0});
    return r;
}

// acc[pt][mt] = sum_ks A[mt][ks] . b[pt][ks]; A fragments read from LDS
template <int MT, int KS>
__device__ __forceinline__ void gemm(const half8 *__restrict__ frags, int lane, const half8 (&b)[BPT][KS], f32x16 (&acc)[BPT][MT])
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the A fragment of the NEXT matrix instruction is read before this one is issued: with one wave per SIMD and a fence behind every instruction the LDS
    // latency of a fragment read where it is used stands in front of each of the ~70 matrix instructions of the two chains (NRF_BWD_A_PREFETCH=0: as before)
#ifndef NRF_BWD_A_PREFETCH
#define NRF_BWD_A_PREFETCH 0            // measured: 1.70 against 1.64 ms per call with both this and the paired transposes (r5g_*): the LDS reads were never what the wave waits for
#endif
    half8 a_next = frags[lane];
#pragma unroll
    for (int i = 0; i < MT * KS; i++) {
        const int mt = i / KS, ks = i % KS;
        half8 a = a_next;
        if (!NRF_BWD_A_PREFETCH) a = frags[i * 64 + lane];
        else if (i + 1 < MT * KS) a_next = frags[(i + 1) * 64 + lane];
#pragma unroll
        for (int pt = 0; pt < BPT; pt++) acc[pt][mt] = mfma(a, b[pt][ks], ks == 0 ? zero : acc[pt][mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// zero the gradient where the forward activation (the operand fragment of that layer) was clipped by the ReLU, then -> operand
__device__ __forceinline__ void mask_to_frags(const f32x16 (&acc)[BPT][2], const half8 (&hf)[BPT][4], half8 (&g)[BPT][4])
{
#pragma unroll
    for (int pt = 0; pt < BPT; pt++)
#pragma unroll
        for (int f = 0; f < 4; f++) {
            half8 v = to_frag<false>(acc[pt][f >> 1], f & 1);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = hf[pt][f][j] > (_Float16)0 ? v[j] : (_Float16)0;
            g[pt][f] = v;
        }
}

// dW tiles [o-group mo][i-group ni] += sum over the wave's points; GT / HT = transposed operand fragments.  The accumulators live in
// registers (AGPRs) for the whole persistent loop: with one wave per SIMD the 20 tiles of the network (320 registers) fit next to the
// working set, and nothing is added to memory until the loop is over.  (A first version added every tile to an fp32 copy of the blob in LDS
// with ds_add_f32 after each 64 points: 133 clocks per instruction and CU, 70 % of the kernel's time.)
template <int GM, int HN>
__device__ __forceinline__ void dw_update(const half8 (&gt)[GM][BPT][2], const half8 (&ht)[HN][BPT][2], f32x16 (&acc)[GM * HN])
{
#pragma unroll
    for (int mo = 0; mo < GM; mo++)
#pragma unroll
        for (int ni = 0; ni < HN; ni++) {
#pragma unroll
            for (int pt = 0; pt < BPT; pt++)
#pragma unroll
                for (int s = 0; s < 2; s++) acc[mo * HN + ni] = mfma(gt[mo][pt][s], ht[ni][pt][s], acc[mo * HN + ni]);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// end of the kernel: a wave's tiles -> the workgroup's fp32 copy of the gradient blob in LDS
template <int GM, int HN, int OUT, int IN>
__device__ __forceinline__ void dw_flush(const f32x16 (&acc)[GM * HN], float *dwl, int r, int h)
{
#pragma unroll
    for (int mo = 0; mo < GM; mo++)
#pragma unroll
        for (int ni = 0; ni < HN; ni++) {
            const int i = ni * 32 + r, ob = mo * 32 + h * 4;
            if (i < IN) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int o = ob + (q >> 2) * 8 + (q & 3);
                    if (o < OUT) __hip_atomic_fetch_add(dwl + o * IN + i, acc[mo * HN + ni][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
}

template <int NL, int NLC>
__global__ void __launch_bounds__(64 * BW, 1)
k_small_bwd(int64_t npts, const float *__restrict__ x, int xs, const float *__restrict__ g_out, int gos, const half8 *__restrict__ fimg,
            const half8 *__restrict__ bimg, half8 *__restrict__ scratch, float *__restrict__ g_params, float *__restrict__ g_x, int gxs,
            const uint32_t *__restrict__ absmax_bits, int n_params, LayerOffs lo, LmInput lm)
{
    using P = BwdPlan<NL, NLC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *wf = reinterpret_cast<half8 *>(smem);
    half8 *wb = wf + P::fwd_frags() * 64;
    float *dw = reinterpret_cast<float *>(wb + P::bwd_frags() * 64);
    for (int i = threadIdx.x; i < P::fwd_frags() * 64; i += blockDim.x) wf[i] = fimg[i];
    for (int i = threadIdx.x; i < P::bwd_frags() * 64; i += blockDim.x) wb[i] = bimg[i];
    for (int i = threadIdx.x; i < n_params; i += blockDim.x) dw[i] = 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: the scratch addresses below are SGPR base + lane
    const int r = lane & 31, h = lane >> 5;
    // loss scale: max|g_out| * S lands in [16, 32)
    const uint32_t mb = *absmax_bits;
    int e = (int)(mb >> 23) - 127;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float S = mb == 0 ? 1.0f : __builtin_ldexpf(1.0f, 4 - e), invS = 1.0f / S;
    // selector operands of the transposition: column r of the product picks ...
    half8 sel0, sel1, nat0, nat1, selgeo;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        sel0[j] = prow(0, h, j) == r ? (_Float16)1 : (_Float16)0;               // ... neuron r of a chained fragment pair (registers of a D tile)
        sel1[j] = prow(1, h, j) == r ? (_Float16)1 : (_Float16)0;
        nat0[j] = 8 * h + j == r ? (_Float16)1 : (_Float16)0;                   // ... element r of a pair loaded in natural order
        nat1[j] = 16 + 8 * h + j == r ? (_Float16)1 : (_Float16)0;
        const int row = prow(0, h, j);                                          // ... geo feature row - 1 as colour-net input V + row - 1
        selgeo[j] = (row >= 1 && row <= GEO && V + row - 1 == r) ? (_Float16)1 : (_Float16)0;
    }
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const half8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
    // fragment pair (lane = point) -> pair of fragments with lane = neuron, k = points
    auto transpose = [&](const half8 &a0, const half8 &s0, const half8 &a1, const half8 &s1, half8 (&t)[2]) {
        f32x16 d = mfma(a0, s0, zero);
        d = mfma(a1, s1, d);
        t[0] = to_frag<false>(d, 0); t[1] = to_frag<false>(d, 1);
        __builtin_amdgcn_sched_barrier(0);       // unfenced, every product of a layer is issued before the first conversion: 16 live D tiles
    };
    auto transpose1 = [&](const half8 &a0, const half8 &s0, half8 (&t)[2]) {
        const f32x16 d = mfma(a0, s0, zero);
        t[0] = to_frag<false>(d, 0); t[1] = to_frag<false>(d, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // both 32-neuron groups' products are issued before the first conversion (two live D tiles): the second pair of matrix instructions runs while the first result lands
#ifndef NRF_BWD_TRANSPOSE_PAIRS
#define NRF_BWD_TRANSPOSE_PAIRS 0
#endif
    auto transpose64 = [&](const half8 (&f)[BPT][4], half8 (&t)[2][BPT][2]) {
#pragma unroll
        for (int pt = 0; pt < BPT; pt++) {
            if (NRF_BWD_TRANSPOSE_PAIRS) {
                f32x16 d0 = mfma(f[pt][0], sel0, zero), d1 = mfma(f[pt][2], sel0, zero);
                d0 = mfma(f[pt][1], sel1, d0); d1 = mfma(f[pt][3], sel1, d1);
                __builtin_amdgcn_sched_barrier(0);
                t[0][pt][0] = to_frag<false>(d0, 0); t[0][pt][1] = to_frag<false>(d0, 1);
                t[1][pt][0] = to_frag<false>(d1, 0); t[1][pt][1] = to_frag<false>(d1, 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int g = 0; g < 2; g++) transpose(f[pt][2 * g], sel0, f[pt][2 * g + 1], sel1, t[g][pt]);
            }
        }
    };

    // weight-gradient accumulators of this wave, one 32x32 tile per (out group, in group); the hidden x hidden layers have four, the rest two
    // (a_c[l] / a_s[l] are used for the hidden layers 1 .. N-2 only)
    f32x16 a_c[NLC][4], a_s[NL][4], a_c0[2], a_s0[2], a_cl[2], a_sl[2];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int l = 1; l < NLC - 1; l++) a_c[l][t] = zero;
#pragma unroll
        for (int l = 1; l < NL - 1; l++) a_s[l][t] = zero;
    }
    a_c0[0] = a_c0[1] = a_s0[0] = a_s0[1] = a_cl[0] = a_cl[1] = a_sl[0] = a_sl[1] = zero;
    const int64_t nblocks = (npts + BW_BLOCK_PTS - 1) / BW_BLOCK_PTS;
#ifdef NRF_BWD_TRACE
    unsigned long long tr[16] = {}, tprev = __builtin_readcyclecounter();
    const unsigned long long tstart = tprev;
#endif
    // One wave per SIMD and a dependent chain: whatever a pass loads where it needs it, it waits for in full (cycle stamps of -DNRF_BWD_TRACE builds,
    // tools/scratch/bwd_trace.py, docs/history/profiles/round4/r5g_*: of 33 k cycles per pass 6.4 k in front of the inputs, 1.9 k in front of the output gradients, ~13 k in front
    // of the fragment reloads of the backward chain).  Most of that was HBM traffic of the fragment scratch, gone with the per-wave slot (below).  Requesting things
    // ahead -- bit 1: the NEXT pass's inputs (level-major form) and output gradients while this pass computes; bit 2: each backward layer requests the fragments of
    // the layer after it -- moves the waits without shortening the pass (r5j_*: training step 6.60-6.67 ms with 0, 1 or 2; the stamps show the cycles reappear at the
    // g_x stores, behind which every later wait of the in-order counter queues), and 3 crashes this compiler: off.
#ifndef NRF_BWD_PREFETCH
#define NRF_BWD_PREFETCH 0
#endif
    half8 nx[BPT][IN_KS], nc[BPT];
    float4 ng[BPT];
    auto request_inputs = [&](int64_t b) {
#pragma unroll
        for (int pt = 0; pt < BPT; pt++) {
            int64_t p = b * BW_BLOCK_PTS + wave * (32 * BPT) + pt * 32 + r;
            if (p >= npts) p = npts - 1;
            const int64_t col = lm.src ? (int64_t)lm.src[lm.p_base + p] : p;
#pragma unroll
            for (int s = 0; s < IN_KS; s++) {
                union { half8 v; __half2 q[4]; } u;
#pragma unroll
                for (int q = 0; q < 4; q++) u.q[q] = lm.feats[(int64_t)(8 * s + 4 * h + q) * lm.pstride + col];      // features 16s + 8h + 2q, +1
                nx[pt][s] = u.v;
            }
            const int64_t ray = (lm.p_base + p) / lm.s;
            nc[pt] = *reinterpret_cast<const half8 *>(lm.dirs + ray * V + 8 * h);
        }
    };
    auto request_gout = [&](int64_t b) {
#pragma unroll
        for (int pt = 0; pt < BPT; pt++) {
            const int64_t p = b * BW_BLOCK_PTS + wave * (32 * BPT) + pt * 32 + r;
            ng[pt] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (p < npts) ng[pt] = *reinterpret_cast<const float4 *>(g_out + p * gos);
        }
    };
    if ((NRF_BWD_PREFETCH & 1) && (int64_t)blockIdx.x < nblocks) {
        if (lm.feats) request_inputs(blockIdx.x);
        request_gout(blockIdx.x);
    }
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t p0 = blk * BW_BLOCK_PTS + wave * (32 * BPT);
        const bool more = blk + gridDim.x < nblocks;
        NRF_BSTAMP(15);
        // this wave's fragments: [pt][fragment][lane]; wave-uniform pointer.  The SAME 24 KB every pass: a region per tile of the launch (2.4 GB for a training
        // batch) made every fragment a line written back to HBM and, with 3 MB of them in flight per XCD, mostly re-read from there -- 4.8 GB per call, which is
        // what the kernel's 1.6 ms were (docs/history/profiles/round4/r5i_*); a wave's own slot is rewritten while it is still in the L2
        half8 *hs = scratch + ((size_t)(blockIdx.x * BW + wave) * BPT * P::h_frags()) * 64;
        // the asm pins each fragment's base as an SGPR pair where it is used: left alone, the compiler precomputes one 64-bit VGPR address per
        // 4-KB window outside the persistent loop (20 register pairs) and spills them
        auto hstore = [&](int pt, int f, const half8 &v) { half8 *sp = hs + (pt * P::h_frags() + f) * 64; asm volatile("" : "+s"(sp)); sp[lane] = v; };
        auto hload = [&](int pt, int f) { const half8 *sp = hs + (pt * P::h_frags() + f) * 64; asm volatile("" : "+s"(sp)); return sp[lane]; };
        // =============================== forward, keeping every layer's operand fragments ===============================
        half8 bh[BPT][4];
        f32x16 acc2[BPT][2];
        {
            half8 bx[BPT][IN_KS], bc[BPT][2];
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) {
                int64_t p = p0 + pt * 32 + r;
                if (p >= npts) p = npts - 1;
                if (lm.feats) {
                    if (!(NRF_BWD_PREFETCH & 1)) request_inputs(blk);
#pragma unroll
                    for (int s = 0; s < IN_KS; s++) { bx[pt][s] = nx[pt][s]; hstore(pt, P::h_sigma(0) + s, bx[pt][s]); }
                    bc[pt][0] = nc[pt];
                    hstore(pt, P::h_color(0), bc[pt][0]);
                    continue;
                }
                const float *row = x + p * xs;
#pragma unroll
                for (int s = 0; s < IN_KS + 1; s++) {
                    const float4 a = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h), b = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h + 4);
                    const half8 v = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
                    if (s < IN_KS) { bx[pt][s] = v; hstore(pt, P::h_sigma(0) + s, v); }
                    else { bc[pt][0] = v; hstore(pt, P::h_color(0), v); }
                }
            }
            if ((NRF_BWD_PREFETCH & 1) && lm.feats && more) request_inputs(blk + gridDim.x);
            NRF_BSTAMP(0);
            const half8 *fr = wf;
            gemm<2, IN_KS>(fr, lane, bx, acc2); fr += P::sigma_frags(0) * 64;
#pragma unroll
            for (int l = 1; l < NL; l++) {
#pragma unroll
                for (int pt = 0; pt < BPT; pt++)
#pragma unroll
                    for (int f = 0; f < 4; f++) { bh[pt][f] = to_frag<true>(acc2[pt][f >> 1], f & 1); hstore(pt, P::h_sigma(l) + f, bh[pt][f]); }
                if (l < NL - 1) { gemm<2, 4>(fr, lane, bh, acc2); }
                else {
                    f32x16 sig[BPT][1];
                    gemm<1, 4>(fr, lane, bh, sig);
#pragma unroll
                    for (int pt = 0; pt < BPT; pt++) { bc[pt][1] = to_frag<false>(sig[pt][0], 0); hstore(pt, P::h_color(0) + 1, bc[pt][1]); }
                }
                fr += P::sigma_frags(l) * 64;
            }
            gemm<2, 2>(fr, lane, bc, acc2); fr += P::color_frags(0) * 64;
#pragma unroll
            for (int l = 1; l < NLC; l++) {
#pragma unroll
                for (int pt = 0; pt < BPT; pt++)
#pragma unroll
                    for (int f = 0; f < 4; f++) { bh[pt][f] = to_frag<true>(acc2[pt][f >> 1], f & 1); hstore(pt, P::h_color(l) + f, bh[pt][f]); }
                if (l < NLC - 1) { gemm<2, 4>(fr, lane, bh, acc2); fr += P::color_frags(l) * 64; }
            }
        }
        NRF_BSTAMP(1);
        // ======================================= backward =======================================
        // bh = c_{NLC-1}, the input of the last colour layer, still in registers
        // stage k of the backward chain reloads: k < NLC - 2: hidden colour layer NLC - 2 - k (4 fragments); NLC - 2: colour layer 0 (2); NLC - 1: last sigma layer (4);
        // then the hidden sigma layers NL - 2 .. 1 (4 each); last: sigma layer 0 (2).  bn = the fragments requested for the next stage.
        half8 bn[BPT][4];
        auto request_stage = [&](int k) {
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) {
                if (k < NLC - 2) { for (int f = 0; f < 4; f++) bn[pt][f] = hload(pt, P::h_color(NLC - 2 - k) + f); }
                else if (k == NLC - 2) { bn[pt][0] = hload(pt, P::h_color(0)); bn[pt][1] = hload(pt, P::h_color(0) + 1); }
                else if (k == NLC - 1) { for (int f = 0; f < 4; f++) bn[pt][f] = hload(pt, P::h_sigma(NL - 1) + f); }
                else if (k < NLC + NL - 2) { for (int f = 0; f < 4; f++) bn[pt][f] = hload(pt, P::h_sigma(NL - 2 - (k - NLC)) + f); }
                else { bn[pt][0] = hload(pt, P::h_sigma(0)); bn[pt][1] = hload(pt, P::h_sigma(0) + 1); }
            }
        };
        if (NRF_BWD_PREFETCH & 2) request_stage(0);
        half8 g1[BPT][1];
        float gsig[BPT];
        if (!(NRF_BWD_PREFETCH & 1)) request_gout(blk);
#pragma unroll
        for (int pt = 0; pt < BPT; pt++) {
            const float4 g = ng[pt];
            g1[pt][0] = h == 0 ? half8{(_Float16)(g.x * S), (_Float16)(g.y * S), (_Float16)(g.z * S), 0, 0, 0, 0, 0} : hzero;
            gsig[pt] = g.w * S;
        }
        if ((NRF_BWD_PREFETCH & 1) && more) request_gout(blk + gridDim.x);
        NRF_BSTAMP(2);
        const half8 *br = wb;
        half8 gf[BPT][4];
        half8 gt[2][BPT][2], ht[2][BPT][2];
        {   // colour layer NLC-1: c -> rgb
            half8 gt1[1][BPT][2];
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) transpose1(g1[pt][0], nat0, gt1[0][pt]);
            transpose64(bh, ht);
            dw_update<1, 2>(gt1, ht, a_cl);
            gemm<2, 1>(br, lane, g1, acc2); br += P::bwd_color_frags(NLC - 1) * 64;
            mask_to_frags(acc2, bh, gf);
        }
        NRF_BSTAMP(3);
#pragma unroll
        for (int l = NLC - 2; l >= 1; l--) {            // hidden colour layers c_l -> c_{l+1}
#pragma unroll
            for (int pt = 0; pt < BPT; pt++)
#pragma unroll
                for (int f = 0; f < 4; f++) bh[pt][f] = (NRF_BWD_PREFETCH & 2) ? bn[pt][f] : hload(pt, P::h_color(l) + f);
            if (NRF_BWD_PREFETCH & 2) request_stage(NLC - 2 - l + 1);
            NRF_BSTAMP(4);
            transpose64(bh, ht); transpose64(gf, gt);
            NRF_BSTAMP(5);
            dw_update<2, 2>(gt, ht, a_c[l]);
            NRF_BSTAMP(6);
            gemm<2, 4>(br, lane, gf, acc2); br += P::bwd_color_frags(l) * 64;
            NRF_BSTAMP(7);
            mask_to_frags(acc2, bh, gf);
            NRF_BSTAMP(8);
        }
        half8 g3[BPT][1];
        {   // colour layer 0: [views, geo] -> c_1; only the geo rows propagate; sigma's own gradient joins at row 0
            half8 ht1[1][BPT][2];
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) {
                const half8 f0 = (NRF_BWD_PREFETCH & 2) ? bn[pt][0] : hload(pt, P::h_color(0)), f1 = (NRF_BWD_PREFETCH & 2) ? bn[pt][1] : hload(pt, P::h_color(0) + 1);
                if ((NRF_BWD_PREFETCH & 2) && pt == BPT - 1) request_stage(NLC - 1);
                transpose(f0, nat0, f1, selgeo, ht1[0][pt]);
            }
            transpose64(gf, gt);
            dw_update<2, 1>(gt, ht1, a_c0);
            f32x16 a1[BPT][1];
            gemm<1, 4>(br, lane, gf, a1); br += P::bwd_color_frags(0) * 64;
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) {
                if (h == 0) a1[pt][0][0] += gsig[pt];
                g3[pt][0] = to_frag<false>(a1[pt][0], 0);
            }
        }
        NRF_BSTAMP(9);
        {   // sigma layer NL-1: h -> [sigma, geo]
#pragma unroll
            for (int pt = 0; pt < BPT; pt++)
#pragma unroll
                for (int f = 0; f < 4; f++) bh[pt][f] = (NRF_BWD_PREFETCH & 2) ? bn[pt][f] : hload(pt, P::h_sigma(NL - 1) + f);
            if (NRF_BWD_PREFETCH & 2) request_stage(NLC);
            half8 gt1[1][BPT][2];
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) transpose1(g3[pt][0], sel0, gt1[0][pt]);
            transpose64(bh, ht);
            dw_update<1, 2>(gt1, ht, a_sl);
            gemm<2, 1>(br, lane, g3, acc2); br += P::bwd_sigma_frags(NL - 1) * 64;
            mask_to_frags(acc2, bh, gf);
        }
        NRF_BSTAMP(10);
#pragma unroll
        for (int l = NL - 2; l >= 1; l--) {
#pragma unroll
            for (int pt = 0; pt < BPT; pt++)
#pragma unroll
                for (int f = 0; f < 4; f++) bh[pt][f] = (NRF_BWD_PREFETCH & 2) ? bn[pt][f] : hload(pt, P::h_sigma(l) + f);
            if (NRF_BWD_PREFETCH & 2) request_stage(NLC + (NL - 2 - l) + 1);
            transpose64(bh, ht); transpose64(gf, gt);
            dw_update<2, 2>(gt, ht, a_s[l]);
            gemm<2, 4>(br, lane, gf, acc2); br += P::bwd_sigma_frags(l) * 64;
            mask_to_frags(acc2, bh, gf);
        }
        NRF_BSTAMP(11);
        {   // sigma layer 0: x -> h_1; g_x leaves in fp32
            half8 ht1[1][BPT][2];
#pragma unroll
            for (int pt = 0; pt < BPT; pt++) {
                const half8 f0 = (NRF_BWD_PREFETCH & 2) ? bn[pt][0] : hload(pt, P::h_sigma(0)), f1 = (NRF_BWD_PREFETCH & 2) ? bn[pt][1] : hload(pt, P::h_sigma(0) + 1);
                transpose(f0, nat0, f1, nat1, ht1[0][pt]);
            }
            transpose64(gf, gt);
            dw_update<2, 1>(gt, ht1, a_s0);
            if (g_x) {
                f32x16 a1[BPT][1];
                gemm<1, 4>(br, lane, gf, a1);
#pragma unroll
                for (int pt = 0; pt < BPT; pt++) {
                    const int64_t p = p0 + pt * 32 + r;
                    if (p < npts) {
                        float *gr = g_x + p * gxs;
#pragma unroll
                        for (int q = 0; q < 4; q++)      // registers 4q..4q+3 = rows 8q + 4h + 0..3
                            *reinterpret_cast<float4 *>(gr + 8 * q + 4 * h) = float4{a1[pt][0][4 * q] * invS, a1[pt][0][4 * q + 1] * invS, a1[pt][0][4 * q + 2] * invS, a1[pt][0][4 * q + 3] * invS};
                    }
                }
            }
        }
        NRF_BSTAMP(12);
    }
#ifdef NRF_BWD_TRACE
    { NRF_BSTAMP(13); }
#endif
    // ---- the wave's tiles -> LDS copy of the blob -> global (one float atomic per parameter and workgroup) ----
    {
        dw_flush<1, 2, 3, 64>(a_cl, dw + lo.c[NLC - 1], r, h);
#pragma unroll
        for (int l = NLC - 2; l >= 1; l--) dw_flush<2, 2, 64, 64>(a_c[l], dw + lo.c[l], r, h);
        dw_flush<2, 1, 64, V + GEO>(a_c0, dw + lo.c[0], r, h);
        dw_flush<1, 2, 1 + GEO, 64>(a_sl, dw + lo.s[NL - 1], r, h);
#pragma unroll
        for (int l = NL - 2; l >= 1; l--) dw_flush<2, 2, 64, 64>(a_s[l], dw + lo.s[l], r, h);
        dw_flush<2, 1, 64, 32>(a_s0, dw + lo.s[0], r, h);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_params; i += blockDim.x) {
        const float v = dw[i];
        if (v != 0.0f) unsafeAtomicAdd(g_params + i, v * invS);
    }
#ifdef NRF_BWD_TRACE
    tr[14] = __builtin_readcyclecounter() - tstart;
    if (threadIdx.x == 0) for (int i = 0; i < 16; i++) g_bwd_trace[(blockIdx.x & 255) * 16 + i] += tr[i];
#endif
}

#ifdef NRF_BWD_TRACE
}  // namespace
extern "C" NRF_API int nrf_dbg_bwd_trace(unsigned long long *host_out, int reset)
{
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bwd_trace), sizeof(unsigned long long) * 256 * 16) != hipSuccess) return NRF_ERR_HIP;
    if (reset) { static unsigned long long z[256 * 16]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace), z, sizeof(z)) != hipSuccess) return NRF_ERR_HIP; }
    return NRF_OK;
}
namespace {
#endif

__global__ void k_absmax(int64_t n, const float *__restrict__ g, uint32_t *__restrict__ out)
{
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(g[i]) & 0x7fffffffu;
        if (b < 0x7f800000u) { if (b > m) m = b; }      // the scale comes from the finite values ...
        else out[1] = 1u;                                // ... and an inf / NaN in the incoming gradient is flagged (nrf_mlp_backward_f16_flags)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// flags[2] = 1 if any accumulated parameter gradient is not finite (an fp16 operand or product of the chain overflowed)
__global__ void k_flag_nonfinite(int64_t n, const float *__restrict__ g, uint32_t *__restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if ((__float_as_uint(g[i]) & 0x7fffffffu) >= 0x7f800000u) flags[2] = 1u;
}

struct BPacker {
    std::vector<_Float16> img;
    // A[row][k] = a[row * ld + k] for row < rows, k < kn; k of element j of lane-half h in k-step ks = kmap(ks, h, j)
    template <class KMap>
    void layer(const std::vector<float> &a, int rows, int kn, int mtiles, int ksteps, KMap kmap)
    {
        for (int mt = 0; mt < mtiles; mt++)
            for (int ks = 0; ks < ksteps; ks++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int row = mt * 32 + (lane & 31), k = kmap(ks, lane >> 5, j);
                        img.push_back((_Float16)((row < rows && k >= 0 && k < kn) ? a[(size_t)row * kn + k] : 0.0f));
                    }
    }
};

bool bwd_supported(const nrf_mlp_small_desc &d)
{
    return !d.use_pred_normal && d.input_ch == 32 && d.input_ch_views == V && d.hidden_dim == 64 && d.hidden_dim_color == 64 && d.geo_feat_dim == GEO &&
           (d.num_layers == 2 || d.num_layers == 3) && (d.num_layers_color == 3 || d.num_layers_color == 4);
}

LayerOffs layer_offsets(const nrf_mlp *m)
{
    LayerOffs lo{};
    const auto &d = m->small;
    for (int l = 0; l < d.num_layers; l++) lo.s[l] = (int)m->layers[l].w_off;
    for (int l = 0; l < d.num_layers_color; l++) lo.c[l] = (int)m->layers[d.num_layers + l].w_off;
    return lo;
}

}  // namespace

// W^T fragments of every layer in the order the gradient chain consumes them
bool mlp_small_bwd_image_host(const nrf_mlp *m, const std::vector<float> &hp, std::vector<uint8_t> &img)
{
    const auto &d = m->small;
    bool ok = bwd_supported(d);
    for (auto &L : m->layers) if (L.d_bias) ok = false;
    if (!ok) return false;
    const int G = d.geo_feat_dim;
    auto natural = [](int ks, int h, int j) { return 16 * ks + 8 * h + j; };
    auto chained = [](int ks, int h, int j) { return 32 * (ks >> 1) + prow(ks & 1, h, j); };
    auto transposed = [&](const nrf::LinearLayer &L) {          // [in][out]
        std::vector<float> t((size_t)L.in * L.out);
        for (int o = 0; o < L.out; o++)
            for (int k = 0; k < L.in; k++) t[(size_t)k * L.out + o] = hp[L.w_off + (size_t)o * L.in + k];
        return t;
    };
    BPacker pk;
    for (int l = d.num_layers_color - 1; l >= 0; l--) {
        const auto &L = m->layers[d.num_layers + l];
        if (l == d.num_layers_color - 1) pk.layer(transposed(L), 64, 3, 2, 1, natural);               // rows = c neurons, k = rgb
        else if (l > 0) pk.layer(transposed(L), 64, 64, 2, 4, chained);
        else {                                                                                      // rows 1..G = geo inputs V.., k = c_1 neurons
            std::vector<float> a((size_t)32 * 64, 0.0f);
            for (int row = 1; row <= G; row++)
                for (int k = 0; k < 64; k++) a[(size_t)row * 64 + k] = hp[L.w_off + (size_t)k * L.in + V + row - 1];
            pk.layer(a, 32, 64, 1, 4, chained);
        }
    }
    for (int l = d.num_layers - 1; l >= 0; l--) {
        const auto &L = m->layers[l];
        if (l == d.num_layers - 1) pk.layer(transposed(L), 64, 1 + G, 2, 1, chained);                 // k = rows 0..15 of the [sigma, geo] tile
        else if (l > 0) pk.layer(transposed(L), 64, 64, 2, 4, chained);
        else pk.layer(transposed(L), 32, 64, 1, 4, chained);
    }
    img.assign(reinterpret_cast<const uint8_t *>(pk.img.data()), reinterpret_cast<const uint8_t *>(pk.img.data() + pk.img.size()));
    return true;
}

int mlp_small_pack_bwd(nrf_mlp *m, const std::vector<float> &hp)
{
    std::vector<uint8_t> img;
    if (!mlp_small_bwd_image_host(m, hp, img)) {
        if (m->d_packed_bwd) { (void)hipFree(m->d_packed_bwd); m->d_packed_bwd = nullptr; m->packed_bwd_bytes = 0; }
        return NRF_OK;
    }
    const size_t bytes = img.size();
    if (m->d_packed_bwd && m->packed_bwd_bytes != bytes) { (void)hipFree(m->d_packed_bwd); m->d_packed_bwd = nullptr; }
    if (!m->d_packed_bwd) NRF_HIP(hipMalloc(&m->d_packed_bwd, bytes));        // host path of nrf_mlp_set_params: overwritten in place
    m->packed_bwd_bytes = bytes;
    NRF_HIP(hipMemcpy(m->d_packed_bwd, img.data(), bytes, hipMemcpyHostToDevice));
    return NRF_OK;
}

static const int64_t BWD_MFMA_CHUNK = 1 << 22;       // points per launch; every launch ends with one float atomic per parameter and workgroup

static size_t h_frags_of(const nrf_mlp_small_desc &d) { return 2 + 4 * (d.num_layers - 1) + 2 + 4 * (d.num_layers_color - 1); }

size_t mlp_small_backward_mfma_workspace_bytes(const nrf_mlp *m, int64_t p)
{
    (void)p;                                       // one fragment slot per resident wave (256 persistent workgroups), whatever the batch
    return 256 + (size_t)256 * (BW_BLOCK_PTS / 32) * h_frags_of(m->small) * 1024;
}

static int backward_mfma_impl(const nrf_mlp *m, const float *x, int xs, const __half2 *feats_lm, const __half *dirs, int s_per_ray, const float *g_out, int gos, int64_t p,
                              float *g_params, float *g_x, int gxs, void *ws, size_t ws_bytes, hipStream_t st, int64_t lm_pstride = 0, const int32_t *lm_src = nullptr);

int mlp_small_backward_mfma(const nrf_mlp *m, const float *x, int xs, const float *g_out, int gos, int64_t p, float *g_params, float *g_x, int gxs, void *ws,
                            size_t ws_bytes, hipStream_t st)
{
    if ((xs % 4) != 0 || (reinterpret_cast<uintptr_t>(x) & 15)) { set_error("nrf_mlp_backward_f16: rows must be 16-byte aligned"); return NRF_ERR_INVALID_ARG; }
    return backward_mfma_impl(m, x, xs, nullptr, nullptr, 1, g_out, gos, p, g_params, g_x, gxs, ws, ws_bytes, st);
}

int mlp_small_backward_mfma_lm(const nrf_mlp *m, const __half2 *feats_lm, const __half *dirs, int s_per_ray, const float *g_out, int gos, int64_t p, float *g_params,
                               float *g_x, int gxs, void *ws, size_t ws_bytes, hipStream_t st, int64_t lm_pstride, const int32_t *lm_src)
{
    if ((reinterpret_cast<uintptr_t>(feats_lm) & 3) || (reinterpret_cast<uintptr_t>(dirs) & 15) || s_per_ray < 1) { set_error("nrf_mlp_backward_f16_lm: bad feature / direction buffers"); return NRF_ERR_INVALID_ARG; }
    if (lm_src && lm_pstride < 1) { set_error("nrf_mlp_backward_f16_lm_src: bad column stride"); return NRF_ERR_INVALID_ARG; }
    return backward_mfma_impl(m, nullptr, 0, feats_lm, dirs, s_per_ray, g_out, gos, p, g_params, g_x, gxs, ws, ws_bytes, st, lm_pstride, lm_src);
}

static int backward_mfma_impl(const nrf_mlp *m, const float *x, int xs, const __half2 *feats_lm, const __half *dirs, int s_per_ray, const float *g_out, int gos, int64_t p,
                              float *g_params, float *g_x, int gxs, void *ws, size_t ws_bytes, hipStream_t st, int64_t lm_pstride, const int32_t *lm_src)
{
    const auto &d = m->small;
    if (m->family != MLP_SMALL || !m->d_packed_bwd || !m->d_packed_f16) {
        set_error("nrf_mlp_backward_f16: NeRFSmall shape outside the built matrix-core family (in 32, views 16, 64-wide, geo 15, 2-3 + 3-4 layers, no bias); use nrf_mlp_backward");
        return NRF_ERR_UNSUPPORTED;
    }
    if (ws_bytes < mlp_small_backward_mfma_workspace_bytes(m, p)) { set_error("nrf_mlp_backward_f16: workspace %zu < %zu bytes", ws_bytes, mlp_small_backward_mfma_workspace_bytes(m, p)); return NRF_ERR_WORKSPACE; }
    if (g_x && ((gxs % 4) != 0 || (reinterpret_cast<uintptr_t>(g_x) & 15))) { set_error("nrf_mlp_backward_f16: d_g_x rows must be 16-byte aligned"); return NRF_ERR_INVALID_ARG; }
    uint32_t *absmax = reinterpret_cast<uint32_t *>(ws);
    half8 *scratch = reinterpret_cast<half8 *>(reinterpret_cast<unsigned char *>(ws) + 256);
    NRF_HIP(hipMemsetAsync(absmax, 0, 16, st));           // [0] max |g_out| bits, [1] non-finite input flag, [2] non-finite result flag
    {
        const int64_t n = p * gos;
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)(ceil_div(n, (int64_t)256) < 1024 ? ceil_div(n, (int64_t)256) : 1024)), dim3(256), 0, st, n, g_out, absmax);
        NRF_LAUNCH_CHECK();
    }
    const LayerOffs lo = layer_offsets(m);
    const int cus = 256;                                   // persistent: one 8-wave workgroup per CU
    for (int64_t p0 = 0; p0 < p; p0 += BWD_MFMA_CHUNK) {
        const int64_t c = (p - p0) < BWD_MFMA_CHUNK ? (p - p0) : BWD_MFMA_CHUNK;
        const int64_t nb = ceil_div(c, (int64_t)BW_BLOCK_PTS);
        const unsigned grid = (unsigned)(nb < cus ? nb : cus);
#define NRF_BW(NL_, NLC_)                                                                                                                              \
        if (d.num_layers == NL_ && d.num_layers_color == NLC_) {                                                                                          \
            using P = BwdPlan<NL_, NLC_>;                                                                                                                 \
            auto kfn = k_small_bwd<NL_, NLC_>;                                                                                                            \
            const size_t lds = (size_t)(P::fwd_frags() + P::bwd_frags()) * 1024 + (size_t)m->n_params * 4;                                                \
            if (m->packed_f16_bytes != (size_t)P::fwd_frags() * 1024 || m->packed_bwd_bytes != (size_t)P::bwd_frags() * 1024 || lds > 160 * 1024) {        \
                set_error("internal: NeRFSmall backward image sizes do not match the kernel plan"); return NRF_ERR_UNSUPPORTED;                            \
            }                                                                                                                                             \
            NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            const LmInput lmi{feats_lm ? (lm_src ? feats_lm : feats_lm + p0) : nullptr, lm_src ? lm_pstride : p, dirs, s_per_ray, p0, lm_src};                  \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * BW), lds, st, c, x ? x + p0 * xs : nullptr, xs, g_out + p0 * gos, gos, reinterpret_cast<const half8 *>(m->d_packed_f16), \
                               reinterpret_cast<const half8 *>(m->d_packed_bwd), scratch, g_params, g_x ? g_x + p0 * gxs : nullptr, gxs, absmax, (int)m->n_params, lo, lmi); \
                                                                                                     \
        }
        NRF_BW(3, 4) NRF_BW(3, 3) NRF_BW(2, 4) NRF_BW(2, 3)
#undef NRF_BW
        NRF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_flag_nonfinite, dim3(64), dim3(256), 0, st, (int64_t)m->n_params, (const float *)g_params, absmax);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf
