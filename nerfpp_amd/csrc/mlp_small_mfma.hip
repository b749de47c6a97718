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
#include "mlp.h"

namespace nrf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NRF_SMALL_PT
#define NRF_SMALL_PT 2
#endif
constexpr int PT = NRF_SMALL_PT; // 32-point tiles per wave
constexpr int WAVES = 4;
constexpr int BLOCK_PTS = 32 * PT * WAVES;

// neuron (row of a D tile / k of the next layer) held by element j of lane-half h in k-step s of a 32-row tile
__host__ __device__ inline int perm_row(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

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

// acc[pt][mt] += A[mt][ks] . B[pt][ks] over all k-steps; A fragments stream from LDS in consumption order.
template <int MT, int KS>
__device__ __forceinline__ void gemm_layer(const half8 *__restrict__ frags, int lane, const half8 (&b)[PT][KS], f32x16 (&acc)[PT][MT])
{
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[pt][mt][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const half8 a = frags[(mt * KS + ks) * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < PT; pt++) acc[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[pt][ks], acc[pt][mt], 0, 0, 0);
            // fence the scheduler every two k-steps: unfenced it hoists every ds_read_b128 of the network to the top (40 fragments =
            // 160 VGPRs), which costs the occupancy that hides the feature-load latency
            if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// number of 1-KB fragments of each layer, in consumption order
template <int IN_KS, int V_KS, int NL, int NLC>
struct SmallPlan {
    static constexpr int GEO_KS = 1;
    static constexpr int sigma_frags(int l) { return (l == NL - 1 ? 1 : 2) * (l == 0 ? IN_KS : 4); }
    static constexpr int color_frags(int l) { return (l == NLC - 1 ? 1 : 2) * (l == 0 ? (V_KS + GEO_KS) : 4); }
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
};

template <int IN_KS, int V_KS, int NL, int NLC, bool LM>
__global__ void __launch_bounds__(64 * WAVES, 2)
k_mlp_small_mfma(int64_t npts, SmallInput in, const half8 *__restrict__ packed, float *__restrict__ out, int out_stride)
{
    using Plan = SmallPlan<IN_KS, V_KS, NL, NLC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *wl = reinterpret_cast<half8 *>(smem);
    constexpr int NFRAG = Plan::total();
    for (int i = threadIdx.x; i < NFRAG * 64; i += blockDim.x) wl[i] = packed[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t nblocks = (npts + BLOCK_PTS - 1) / BLOCK_PTS;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t p0 = blk * BLOCK_PTS + wave * (32 * PT);
        // ---- layer-0 B fragments straight from the fp32 input rows: element j of k-step s is x[pt][16s + 8h + j] ----
        half8 bx[PT][IN_KS];
        half8 bv[PT][V_KS];
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            int64_t p = p0 + pt * 32 + r;
            if (p >= npts) p = npts - 1;                 // clamp loads; stores are guarded
            if constexpr (LM) {
#pragma unroll
                for (int s = 0; s < IN_KS; s++) {
                    union { half8 v; __half2 q[4]; } u;
#pragma unroll
                    for (int q = 0; q < 4; q++) u.q[q] = in.feats[(int64_t)(8 * s + 4 * h + q) * in.pstride + p];   // features 16s+8h+2q, +1
                    bx[pt][s] = u.v;
                }
                const __half *drow = in.dirs + (p / in.s) * (int64_t)(16 * V_KS);
#pragma unroll
                for (int s = 0; s < V_KS; s++) bv[pt][s] = *reinterpret_cast<const half8 *>(drow + 16 * s + 8 * h);
            } else {
                const float *row = in.x + p * in.x_stride;
#pragma unroll
                for (int s = 0; s < IN_KS; s++) {
                    const float4 lo = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h);
                    const float4 hi = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h + 4);
                    bx[pt][s] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
                }
#pragma unroll
                for (int s = 0; s < V_KS; s++) {
                    const float4 lo = *reinterpret_cast<const float4 *>(row + in.in_ch + 16 * s + 8 * h);
                    const float4 hi = *reinterpret_cast<const float4 *>(row + in.in_ch + 16 * s + 8 * h + 4);
                    bv[pt][s] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
                }
            }
        }
        const half8 *fr = wl;
        // ---- sigma net ----
        half8 bh[PT][4];
        f32x16 acc2[PT][2];
        f32x16 sig[PT][1];
        if constexpr (NL == 1) {
            gemm_layer<1, IN_KS>(fr, lane, bx, sig); fr += Plan::sigma_frags(0) * 64;
        } else {
            gemm_layer<2, IN_KS>(fr, lane, bx, acc2); fr += Plan::sigma_frags(0) * 64;
#pragma unroll
            for (int l = 1; l < NL; l++) {
#pragma unroll
                for (int pt = 0; pt < PT; pt++)
#pragma unroll
                    for (int t = 0; t < 2; t++) { bh[pt][2 * t] = tile_to_frag<true>(acc2[pt][t], 0); bh[pt][2 * t + 1] = tile_to_frag<true>(acc2[pt][t], 1); }
                if (l < NL - 1) gemm_layer<2, 4>(fr, lane, bh, acc2);
                else gemm_layer<1, 4>(fr, lane, bh, sig);
                fr += Plan::sigma_frags(l) * 64;
            }
        }
        // ---- colour net: k-steps = [views..., geo] ----
        half8 bc[PT][V_KS + 1];
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
#pragma unroll
            for (int s = 0; s < V_KS; s++) bc[pt][s] = bv[pt][s];
            bc[pt][V_KS] = tile_to_frag<false>(sig[pt][0], 0);        // rows 0..15 of the sigma tile: sigma (zero weight) + geo
        }
        f32x16 rgb[PT][1];
        if constexpr (NLC == 1) {
            gemm_layer<1, V_KS + 1>(fr, lane, bc, rgb);
        } else {
            gemm_layer<2, V_KS + 1>(fr, lane, bc, acc2); fr += Plan::color_frags(0) * 64;
#pragma unroll
            for (int l = 1; l < NLC; l++) {
#pragma unroll
                for (int pt = 0; pt < PT; pt++)
#pragma unroll
                    for (int t = 0; t < 2; t++) { bh[pt][2 * t] = tile_to_frag<true>(acc2[pt][t], 0); bh[pt][2 * t + 1] = tile_to_frag<true>(acc2[pt][t], 1); }
                if (l < NLC - 1) gemm_layer<2, 4>(fr, lane, bh, acc2);
                else gemm_layer<1, 4>(fr, lane, bh, rgb);
                fr += Plan::color_frags(l) * 64;
            }
        }
        // ---- out = (rgb, sigma): rows 0..2 of the colour tile and row 0 of the sigma tile live in registers 0..2 / 0 of lane-half 0 ----
        if (h == 0) {
#pragma unroll
            for (int pt = 0; pt < PT; pt++) {
                const int64_t p = p0 + pt * 32 + r;
                if (p < npts) {
                    float sg = sig[pt][0][0];
                    if constexpr (LM) { if (in.keep && !in.keep[p]) sg = 0.0f; }
                    if (out_stride == 4) *reinterpret_cast<float4 *>(out + p * 4) = float4{rgb[pt][0][0], rgb[pt][0][1], rgb[pt][0][2], sg};
                    else { float *o = out + p * out_stride; o[0] = rgb[pt][0][0]; o[1] = rgb[pt][0][1]; o[2] = rgb[pt][0][2]; o[3] = sg; }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// weight image
// ---------------------------------------------------------------------------------------------------
struct Packer {
    std::vector<_Float16> img;
    // one layer: W [out][in] row-major at `w`; krow(ks, h, j) -> input index or -1 (zero)
    template <class KMap>
    void layer(const float *w, int in, int out, int mtiles, int ksteps, KMap kmap)
    {
        for (int mt = 0; mt < mtiles; mt++)
            for (int ks = 0; ks < ksteps; ks++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int row = mt * 32 + (lane & 31);
                        const int k = kmap(ks, lane >> 5, j);
                        float v = 0.0f;
                        if (row < out && k >= 0 && k < in) v = w[(size_t)row * in + k];
                        img.push_back((_Float16)v);
                    }
    }
};

static bool small_mfma_supported(const nrf_mlp_small_desc &d)
{
    return d.input_ch == 32 && (d.input_ch_views == 16 || d.input_ch_views == 64) && d.hidden_dim == 64 && d.hidden_dim_color == 64 &&
           d.geo_feat_dim >= 0 && d.geo_feat_dim <= 15 && d.num_layers >= 2 && d.num_layers <= 3 && d.num_layers_color >= 2 && d.num_layers_color <= 4;
}

int mlp_small_pack_f16(nrf_mlp *m, const std::vector<float> &hp)
{
    const auto &d = m->small;
    if (!small_mfma_supported(d)) return NRF_OK;      // NRF_PREC_F16_MFMA then reports NRF_ERR_UNSUPPORTED at forward time
    Packer pk;
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
    m->packed_f16_bytes = pk.img.size() * sizeof(_Float16);
    NRF_HIP(hipMalloc(&m->d_packed_f16, m->packed_f16_bytes));
    NRF_HIP(hipMemcpy(m->d_packed_f16, pk.img.data(), m->packed_f16_bytes, hipMemcpyHostToDevice));
    return NRF_OK;
}

template <int V_KS, int NL, int NLC>
static int launch_small(const nrf_mlp *m, const SmallInput &in, bool lm, int64_t p, float *out, int os, hipStream_t st)
{
    using Plan = SmallPlan<2, V_KS, NL, NLC>;
    const size_t lds = (size_t)Plan::total() * 1024;
    if (lds != m->packed_f16_bytes) { set_error("internal: packed weight image is %zu bytes, kernel expects %zu", m->packed_f16_bytes, lds); return NRF_ERR_INVALID_ARG; }
    const int64_t nblocks = ceil_div(p, BLOCK_PTS);
    const unsigned grid = (unsigned)(nblocks < 768 ? nblocks : 768);        // persistent: 256 CUs x 3 resident workgroups (146 VGPRs, 40-46 KB LDS)
    if (lm) hipLaunchKernelGGL((k_mlp_small_mfma<2, V_KS, NL, NLC, true>), dim3(grid), dim3(64 * WAVES), lds, st, p, in, reinterpret_cast<const half8 *>(m->d_packed_f16), out, os);
    else hipLaunchKernelGGL((k_mlp_small_mfma<2, V_KS, NL, NLC, false>), dim3(grid), dim3(64 * WAVES), lds, st, p, in, reinterpret_cast<const half8 *>(m->d_packed_f16), out, os);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

static int dispatch_small(const nrf_mlp *m, const SmallInput &in, bool lm, int64_t p, float *out, int os, hipStream_t st);

int mlp_small_mfma_available(const nrf_mlp *m) { return m && m->family == MLP_SMALL && m->d_packed_f16 != nullptr; }

// renderer fast path: level-major fp16 features + per-ray fp16 direction features + keep mask -> raw [p,4] (sigma masked)
int mlp_small_forward_mfma_lm(const nrf_mlp *m, const __half2 *feats, int64_t pstride, const __half *dirs, int s, const uint8_t *keep,
                              int64_t p, float *out, hipStream_t st)
{
    if (!mlp_small_mfma_available(m)) { set_error("internal: matrix-core NeRFSmall image missing"); return NRF_ERR_UNSUPPORTED; }
    ProfScope prof(NRF_PROF_MLP, st);
    SmallInput in{nullptr, 0, m->small.input_ch, feats, pstride, dirs, s, keep};
    return dispatch_small(m, in, true, p, out, 4, st);
}

int mlp_small_forward_mfma(const nrf_mlp *m, const float *x, int xs, int64_t p, float *out, int os, hipStream_t st)
{
    const auto &d = m->small;
    if (!m->d_packed_f16) {
        set_error("NRF_PREC_F16_MFMA: NeRFSmall shape (in %d, views %d, %dx%d, geo %d, colour %dx%d) is outside the built matrix-core family; use NRF_PREC_F32",
                  d.input_ch, d.input_ch_views, d.num_layers, d.hidden_dim, d.geo_feat_dim, d.num_layers_color, d.hidden_dim_color);
        return NRF_ERR_UNSUPPORTED;
    }
    if ((xs % 4) != 0 || (reinterpret_cast<uintptr_t>(x) & 15)) { set_error("NRF_PREC_F16_MFMA: input rows must be 16-byte aligned"); return NRF_ERR_INVALID_ARG; }
    SmallInput in{x, xs, d.input_ch, nullptr, 0, nullptr, 1, nullptr};
    return dispatch_small(m, in, false, p, out, os, st);
}

static int dispatch_small(const nrf_mlp *m, const SmallInput &in, bool lm, int64_t p, float *out, int os, hipStream_t st)
{
    const auto &d = m->small;
    const int v = d.input_ch_views / 16;
#define NRF_CASE(V, NL, NLC) if (v == V && d.num_layers == NL && d.num_layers_color == NLC) return launch_small<V, NL, NLC>(m, in, lm, p, out, os, st);
    NRF_CASE(1, 3, 4) NRF_CASE(1, 3, 3) NRF_CASE(1, 3, 2) NRF_CASE(1, 2, 4) NRF_CASE(1, 2, 3) NRF_CASE(1, 2, 2)
    NRF_CASE(4, 3, 4) NRF_CASE(4, 3, 3) NRF_CASE(4, 3, 2) NRF_CASE(4, 2, 4) NRF_CASE(4, 2, 3) NRF_CASE(4, 2, 2)
#undef NRF_CASE
    set_error("internal: no matrix-core instantiation for this NeRFSmall shape");
    return NRF_ERR_UNSUPPORTED;
}

// classic NeRF matrix-core path: see mlp_nerf_mfma.hip
}  // namespace nrf
