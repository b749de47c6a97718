// sigma_small_f32.hip -- the density head of NeRFSmallImpl::forward (NeRF.cpp:372-381, sigma = h[..., 0]) in EXACT fp32 on the matrix
// cores, for the renderer's coarse pass.
//
// Why it exists.  With N_importance > 0 the coarse pass contributes nothing but its compositing weights (NeRFRenderer.h:422-428:
// outputs1.Weights -> SamplePDF), i.e. only sigma: the colour net (59 % of the network's MACs) is dead work there.  And those weights pick the
// fine pass's sample set through searchsorted on CDF plateaus, a DISCONTINUOUS function: a sigma that differs in the last bits moves
// a few samples into another bin and a pixel by up to ~5e-4, however accurate the fine pass is.  So the coarse sigma is evaluated in the
// parity arithmetic itself: v_mfma_f32_32x32x2_f32 is, bit for bit, the ascending-k fmaf chain of NRF_PREC_F32 / the oracle
// (tools/scratch/mfma_f32_probe.hip: 1024 of 1024 outputs identical, denormals included) at the fp32 vector rate without its operand traffic.
// The coarse weights, hence z_fine, then EQUAL the parity mode's, and the matrix-core precisions differ from it only by the fine pass's
// smooth rounding error.
//
// Formulation (transposed, as the fp16 kernels):  H_{l+1}^T [neurons x points] = W [neurons x k] . H_l^T [k x points],
// A = 32 neurons x 2 k (lane (i, hh): W[row i][2 ks + hh]), B = 2 k x 32 points (lane (r, hh): act[2 ks + hh] of point r), one k-step = 2 k
// in ascending order inside the instruction and ascending k-steps through the accumulator.  A D tile holds row 8(q/4) + 4 hh + (q%4) in
// register q of lane half hh; the OUTPUT neurons are free to permute (a neuron's own sum keeps its order), so row i of an m-tile carries
// neuron 2(4(i/8) + i%4) + (i/4)%2: register q of lane half hh is then neuron 2q + hh -- after the ReLU, register q of a D tile IS the B
// operand of k-step q of the next layer, in natural ascending k.  No lane movement, no LDS round trip, no permutation of any sum.
//
// The last layer needs one output (sigma; the geo features feed only the colour net): a 64-term chain per point, on the vector ALUs.  A
// point's hidden values sit in two lanes (even k in lane r, odd k in lane r + 32); one v_permlane32_swap per register PAIR of the wave's
// two point tiles hands lane l both parities of point l (lanes 0-31: tile 0, lanes 32-63: tile 1), and every lane runs the ascending chain
// for its own point: 64 fma + 32 swaps per 64 points instead of 64 matrix instructions with one useful row.
//
// GEO (the renderer's default HashNeRF mode): the kernel also leaves the sigma net's WHOLE last row block -- (sigma, geo_feat) -- for the fine pass, which then runs
// only the colour net at its S coarse depths (mlp_small_mfma.hip, GEOIN) instead of repeating the sigma net there.  The geo features feed a split-precision layer,
// so they are formed the same way: the exact hidden values split into (hi, lo) fp16 fragments, registers 8u..8u+7 of tile t = k-step (t, u) of a 32x32x16 product
// (any assignment of k inside a k-step is fine as long as the weight fragment uses the same one), three matrix instructions per k-step, 12 per 32 points beside the
// 192 fp32 ones.  Natural row order: register q of lane half h of the result is row 8(q/4) + 4h + q%4, which IS element q of the colour net's geo operand fragment
// (perm_row(0, h, q) in mlp_small_mfma.hip) -- split once more and stored as that fragment, planes [hi | lo][column][lane half] of 16 bytes.
//
// Tried and measured (round 3, docs/history/profiles/round3/r3z_sigma_two_crews_ab.log): the packed fp32 FMA of the vector ALUs has the same peak rate as the fp32 matrix
// instruction (157 TFLOP/s each) and the counters of this kernel show the matrix pipe busy 0.81 of the cycles with NO vector co-execution, so half of the waves
// were given the same chains as v_pk_fma_f32 code (lane = point, weights broadcast from LDS, units of 64 points handed out by a ticket counter; both crews
// bit-identical by construction, geo rows as exact fp32 chains).  Per frame, same call: all waves matrix 7.54 ms, all waves vector 8.7 ms, half and half 7.59 ms
// -- the two formulations do not add up, they share what limits them -- against 4.8 ms for this file (whose geo rows cost 12 fp16 matrix instructions per 32
// points instead of 512 packed FMAs per 64).  Not kept.  A registers-only probe settles why (tools/scratch/coexec_probe.hip, docs/history/profiles/round3/
// r3z_fp32_mfma_vs_pk_fma_coexec_probe.log): four matrix waves alone 3.4 ms, four packed-FMA waves alone 4.7 ms, the eight together 8.2 ms -- on this chip the fp32
// matrix instruction and the packed fp32 FMA execute on the same lanes; 157 TFLOP/s is the ceiling of their SUM.
#include "mlp.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sg_half8 __attribute__((ext_vector_type(8)));

// two fp32 values -> packed (hi, lo) fp16 pairs, v = hi + lo to 22 bits (see split_pair in mlp_small_mfma.hip)
__device__ __forceinline__ void sg_split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}
// registers q0..q0+7 of a tile -> one (hi, lo) operand fragment.  The asm reads VALU results only (the max / a copy), never a matrix result directly.
__device__ __forceinline__ void sg_frag(const f32x16 &t, int q0, sg_half8 &hi, sg_half8 &lo)
{
    union { sg_half8 v; uint32_t u[4]; } h, l;
#pragma unroll
    for (int j = 0; j < 4; j++) sg_split_pair(fmaxf(t[q0 + 2 * j], -3.402823466e38f), fmaxf(t[q0 + 2 * j + 1], -3.402823466e38f), h.u[j], l.u[j]);
    hi = h.v; lo = l.v;
}
// ... of the values times a power of two (the range scale of the split-precision operands, mlp.h): the multiply is also the VALU result the asm may read
__device__ __forceinline__ void sg_frag_scaled(const f32x16 &t, int q0, float scale, sg_half8 &hi, sg_half8 &lo)
{
    union { sg_half8 v; uint32_t u[4]; } h, l;
#pragma unroll
    for (int j = 0; j < 4; j++) sg_split_pair(t[q0 + 2 * j] * scale, t[q0 + 2 * j + 1] * scale, h.u[j], l.u[j]);
    hi = h.v; lo = l.v;
}

constexpr int SIG_WAVES = 8;                     // 2 per SIMD: one's vector work (ReLU, swaps, the last layer) under the other's matrix chain
constexpr int SIG_BLOCK_PTS = 64 * SIG_WAVES;    // two 32-point tiles per wave

// neuron carried by row i of an m-tile (see above)
__host__ __device__ inline int sigma_row_neuron(int i) { return 2 * (4 * (i >> 3) + (i & 3)) + ((i >> 2) & 1); }

// acc[pt][mt] = sum over KS k-steps; A fragments from LDS as [mt][ks / 4][lane][4] floats (one ds_read_b128 per four k-steps, shared by
// both point tiles); bfn(pt, ks) yields the B operand
template <int KS, class BFn>
__device__ __forceinline__ void sigma_layer(const f32x4 *__restrict__ img, int lane, f32x16 (&acc)[2][2], BFn bfn)
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
        for (int g = 0; g < KS / 4; g++) {
            const f32x4 a4 = img[(mt * (KS / 4) + g) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ks = 4 * g + j;
#pragma unroll
                for (int pt = 0; pt < 2; pt++) acc[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], bfn(pt, ks), ks == 0 ? zero : acc[pt][mt], 0, 0, 0);
            }
            // fence the scheduler: unfenced it hoists every ds_read_b128 of the layer to its top (96 VGPRs) and spills
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void relu_tiles(f32x16 (&acc)[2][2])
{
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[pt][mt][q] = fmaxf(acc[pt][mt][q], 0.0f);    // one v_max_f32 (this file is built with -fno-honor-nans); equals the oracle's
                                                                                          // x < 0 ? 0 : x on every value (a -0 becomes +0: no sum can tell)
}

// feats: level-major [16][pstride], half2 (CuHashEmbedder: exactly the fp16 numbers the reference's kernel outputs, CuHashEmbedder.cu:95) or
// float2 (HashEmbedder, fp32 features).  image: W0 [2][4][64][4] | (NL == 3: W1 [2][8][64][4]) | w_last [64] floats | geo fragments [4 k-steps][hi, lo][64 lanes] half8.
// GEO: geo [2][geo_stride][2] half8 receives the (sigma, geo_feat) operand fragment of every point (column = point index).
template <int NL, bool F32IN, bool GEO, bool A32>
__global__ void __launch_bounds__(64 * SIG_WAVES)
k_sigma_small_f32(int64_t npts, const void *__restrict__ feats, int64_t pstride, const uint8_t *__restrict__ keep, const float *__restrict__ image,
                  float *__restrict__ sigma, sg_half8 *__restrict__ geo, int64_t geo_stride, const float *__restrict__ scales)
{
    constexpr int W0_F4 = 2 * 4 * 64, W1_F4 = NL == 3 ? 2 * 8 * 64 : 0;
    __shared__ f32x4 wl[W0_F4 + W1_F4];
    __shared__ float wlast[64];
    __shared__ sg_half8 wg[GEO ? 4 * 2 * 64 : 1];
    for (int i = threadIdx.x; i < W0_F4 + W1_F4; i += blockDim.x) wl[i] = reinterpret_cast<const f32x4 *>(image)[i];
    if (threadIdx.x < 64) wlast[threadIdx.x] = image[(W0_F4 + W1_F4) * 4 + threadIdx.x];
    if constexpr (GEO) {
        const sg_half8 *gi = reinterpret_cast<const sg_half8 *>(image + (W0_F4 + W1_F4) * 4 + 64);
        for (int i = threadIdx.x; i < 4 * 2 * 64; i += blockDim.x) wg[i] = gi[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t nblocks = (npts + SIG_BLOCK_PTS - 1) / SIG_BLOCK_PTS;
    float hscale = 1.0f;
    if constexpr (GEO) hscale = scales[SMALL_SCALE_HIDDEN];           // uniform: one scalar load
    (void)hscale;
    // raw operand words of one block iteration: level ks of point (pt, r); lane half hh consumes feature hh
    // On this chip the fp32 matrix instruction runs on the vector ALUs' lanes (header), so every vector instruction of this kernel is time taken from the matrix pipe.
    // fp16 features: each lane half loads ITS 16-bit feature (a 2-byte load at byte offset 2 hh) instead of the packed word plus a shift and a select per value; and
    // with < 2^28 points / columns the addresses are a per-lane 32-bit byte offset beside 16 uniform level bases instead of a 64-bit sum per load.
    static_assert(!A32 || !F32IN, "the 2-byte feature loads are the fp16 input's");
    // A32: the feature planes as ONE buffer resource -- a load is a per-lane 32-bit byte offset plus the level's uniform byte offset in a scalar register, no vector
    // instruction per address (the launcher guarantees 16 planes < 4 GB)
    __amdgpu_buffer_rsrc_t frsrc = __amdgpu_buffer_rsrc_t();
    if constexpr (A32) frsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(feats), 0, (int)(uint32_t)((uint64_t)16 * (uint64_t)pstride * 4u), 0x00020000);
    auto load_inputs = [&](int64_t blk, uint32_t (&x)[2][16]) {
        const int64_t p0 = blk * SIG_BLOCK_PTS + wave * 64;
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
            int64_t p = p0 + pt * 32 + r;
            if (p >= npts) p = npts - 1;             // clamp loads; the store is guarded
            if constexpr (A32) {
                uint32_t pu = (uint32_t)p0 + (uint32_t)(pt * 32 + r);
                pu = pu < (uint32_t)npts ? pu : (uint32_t)npts - 1u;
                const uint32_t voff = pu * 4u + 2u * (uint32_t)hh;
#pragma unroll
                for (int ks = 0; ks < 16; ks++)
                    x[pt][ks] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(frsrc, (int)voff, (int)(uint32_t)((uint64_t)ks * (uint64_t)pstride * 4u), 0);
                continue;
            }
#pragma unroll
            for (int ks = 0; ks < 16; ks++) {
                if constexpr (F32IN) x[pt][ks] = reinterpret_cast<const uint32_t *>(feats)[((int64_t)ks * pstride + p) * 2 + hh];
                else { const uint32_t w = reinterpret_cast<const uint32_t *>(feats)[(int64_t)ks * pstride + p]; x[pt][ks] = hh ? (w >> 16) : (w & 0xffffu); }
            }
        }
    };
    uint32_t x[2][16];
    if ((int64_t)blockIdx.x < nblocks) load_inputs(blockIdx.x, x);
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const bool more = blk + gridDim.x < nblocks;
        f32x16 h0[2][2];
        sigma_layer<16>(wl, lane, h0, [&](int pt, int ks) -> float {
            if constexpr (F32IN) return __uint_as_float(x[pt][ks]);
            else {
                const uint16_t bits = (uint16_t)x[pt][ks];          // the lane half's own feature (load_inputs)
                _Float16 hv; __builtin_memcpy(&hv, &bits, 2);
                return (float)hv;                    // exact
            }
        });
        if (more) load_inputs(blk + gridDim.x, x);   // the next iteration's operands, into the registers layer 0 has just finished with: they land under the rest of the network
        relu_tiles(h0);
        // ---- last layer, output 0 only: lane l takes over point l of the wave's 64 ----
        auto last = [&](const f32x16 (&hl)[2][2]) -> float {
            float a = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    // vdst = tile-0 register, src = tile-1 register: lanes 32-63 of vdst <-> lanes 0-31 of src
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(hl[0][t][q]), __float_as_uint(hl[1][t][q]), false, false);
                    const float ev = __uint_as_float(sw[0]), od = __uint_as_float(sw[1]);     // k = 32t + 2q, 32t + 2q + 1 of this lane's point
                    a = __builtin_fmaf(wlast[32 * t + 2 * q], ev, a);
                    a = __builtin_fmaf(wlast[32 * t + 2 * q + 1], od, a);
                }
            }
            return a;
        };
        // ---- GEO: rows 0..15 of the last layer for both point tiles, in split precision, stored as the colour net's operand fragment ----
        // The exact hidden values enter the split product times 2^S_hidden and meet the last layer's weights times 2^e_last (the tail of the image is gathered from the
        // range-scaled blob): the result is (sigma, geo_feat) x 2^S_sigma, which is what the colour net's scaled geo columns expect (mlp.h, SMALL_MAX_GROUPS)
        auto geo_out = [&](const f32x16 (&hl)[2][2]) {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                f32x16 g = zero;
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        sg_half8 xh, xl;
                        sg_frag_scaled(hl[pt][t], 8 * u, hscale, xh, xl);
                        const sg_half8 ah = wg[((2 * t + u) * 2 + 0) * 64 + lane], al = wg[((2 * t + u) * 2 + 1) * 64 + lane];
                        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, g, 0, 0, 0);
                        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, g, 0, 0, 0);
                        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, g, 0, 0, 0);
                    }
                sg_half8 gh, gl;
                sg_frag(g, 0, gh, gl);
                const int64_t pp = blk * SIG_BLOCK_PTS + wave * 64 + pt * 32 + r;
                if (pp < npts) { geo[(pp << 1) + hh] = gh; geo[((geo_stride + pp) << 1) + hh] = gl; }
            }
        };
        float acc;
        if constexpr (NL == 3) {
            f32x16 h1[2][2];
            sigma_layer<32>(wl + W0_F4, lane, h1, [&](int pt, int ks) -> float { return h0[pt][ks >> 4][ks & 15]; });
            relu_tiles(h1);
            if constexpr (GEO) geo_out(h1);
            acc = last(h1);
        } else {
            if constexpr (GEO) geo_out(h0);
            acc = last(h0);
        }
        const int64_t p = blk * SIG_BLOCK_PTS + wave * 64 + lane;
        if (p < npts) sigma[p] = (keep && !keep[p]) ? 0.0f : acc;                               // NeRFRenderer.h:187-188
    }
}

static bool sigma_f32_supported(const nrf_mlp_small_desc &d)
{
    return d.input_ch == 32 && d.hidden_dim == 64 && (d.num_layers == 2 || d.num_layers == 3);
}

// head: the fp32 fragments of the hidden layers + the sigma row; tail: the whole last layer as split fp16 fragments of the GEO product (stored behind the head)
bool mlp_small_sigma_image_host(const nrf_mlp_small_desc &d, const std::vector<float> &hp, std::vector<uint8_t> &head_f32, std::vector<uint8_t> &tail_f16)
{
    if (!sigma_f32_supported(d)) return false;
    std::vector<float> img;
    size_t off = 0;
    for (int l = 0; l + 1 < d.num_layers; l++) {
        const int in = l == 0 ? d.input_ch : d.hidden_dim, out = d.hidden_dim;
        const float *w = hp.data() + off;
        const int ks_count = in / 2;
        for (int mt = 0; mt < 2; mt++)
            for (int g = 0; g < ks_count / 4; g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 4; j++) {
                        const int row = 32 * mt + sigma_row_neuron(lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
                        img.push_back(w[(size_t)row * in + k]);
                    }
        off += (size_t)in * out;
    }
    for (int k = 0; k < d.hidden_dim; k++) img.push_back(hp[off + k]);       // row 0 (sigma) of the last sigma-net layer [1 + geo][hidden]
    // the whole last layer (rows 0 .. geo) as split fp16 fragments of the GEO product: k-step (t, u), lane (row i, half hh), element j = W[i][32t + 16u + 2j + hh]
    const int rows = 1 + d.geo_feat_dim;
    std::vector<_Float16> frag;
    for (int ks = 0; ks < 4; ks++)
        for (int part = 0; part < 2; part++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int i = lane & 31, k = 32 * (ks >> 1) + 16 * (ks & 1) + 2 * j + (lane >> 5);
                    const float v = i < rows ? hp[off + (size_t)i * d.hidden_dim + k] : 0.0f;
                    const _Float16 hv = (_Float16)v;
                    frag.push_back(part == 0 ? hv : (_Float16)(v - (float)hv));
                }
    head_f32.assign(reinterpret_cast<const uint8_t *>(img.data()), reinterpret_cast<const uint8_t *>(img.data() + img.size()));
    tail_f16.assign(reinterpret_cast<const uint8_t *>(frag.data()), reinterpret_cast<const uint8_t *>(frag.data() + frag.size()));
    return true;
}

int mlp_small_pack_sigma_f32(nrf_mlp *m, const std::vector<float> &hp)
{
    std::vector<uint8_t> head, tail;
    if (!mlp_small_sigma_image_host(m->small, hp, head, tail)) return NRF_OK;
    const size_t bytes = head.size() + tail.size();
    if (m->d_packed_sigma_f32 && m->packed_sigma_f32_bytes != bytes) { (void)hipFree(m->d_packed_sigma_f32); m->d_packed_sigma_f32 = nullptr; }
    if (!m->d_packed_sigma_f32) NRF_HIP(hipMalloc(&m->d_packed_sigma_f32, bytes));
    m->packed_sigma_f32_bytes = bytes;
    NRF_HIP(hipMemcpy(m->d_packed_sigma_f32, head.data(), head.size(), hipMemcpyHostToDevice));
    NRF_HIP(hipMemcpy(static_cast<char *>(m->d_packed_sigma_f32) + head.size(), tail.data(), tail.size(), hipMemcpyHostToDevice));
    return NRF_OK;
}

int mlp_small_sigma_f32_available(const nrf_mlp *m) { return m && m->family == MLP_SMALL && m->d_packed_sigma_f32 != nullptr; }

// sigma [p] = keep ? NeRFSmall sigma-net output 0 : 0, bit-identical to NRF_PREC_F32.  feats: level-major [16][pstride] half2, or float2 when f32_in.
// geo != NULL: also the (sigma, geo_feat) operand fragments of the colour net, planes [hi | lo][geo_stride columns][2 lane halves] of 16 bytes, column = point index
int mlp_small_sigma_f32_lm(const nrf_mlp *m, const void *feats, int f32_in, int64_t pstride, const uint8_t *keep, int64_t p, float *sigma, hipStream_t st, void *geo,
                           int64_t geo_stride)
{
    if (!mlp_small_sigma_f32_available(m)) { set_error("internal: fp32 matrix-core sigma image missing"); return NRF_ERR_UNSUPPORTED; }
    if (p == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_SIGMA, st);
    const int64_t nblocks = ceil_div(p, SIG_BLOCK_PTS);
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);          // persistent: one 8-wave workgroup per CU
    const float *img = reinterpret_cast<const float *>(m->d_packed_sigma_f32);
    const int nl = m->small.num_layers;
    if (geo && (m->small.geo_feat_dim > 15 || m->small.hidden_dim != 64)) { set_error("internal: geo hand-over outside the built NeRFSmall family"); return NRF_ERR_UNSUPPORTED; }
    sg_half8 *g = static_cast<sg_half8 *>(geo);
    const bool a32 = (p >> 26) == 0 && (pstride >> 26) == 0;          // the 16 level planes within 4 GB: 32-bit byte offsets
#define NRF_GO(NL_, F_)                                                                                                                                              \
    do {                                                                                                                                                             \
        constexpr bool A_ = !F_;                                                                                                                                     \
        if (A_ && a32) {                                                                                                                                             \
            if (g) hipLaunchKernelGGL((k_sigma_small_f32<NL_, F_, true, A_>), dim3(grid), dim3(64 * SIG_WAVES), 0, st, p, feats, pstride, keep, img, sigma, g, geo_stride, m->d_scales); \
            else hipLaunchKernelGGL((k_sigma_small_f32<NL_, F_, false, A_>), dim3(grid), dim3(64 * SIG_WAVES), 0, st, p, feats, pstride, keep, img, sigma, g, geo_stride, m->d_scales); \
        } else if (g) hipLaunchKernelGGL((k_sigma_small_f32<NL_, F_, true, false>), dim3(grid), dim3(64 * SIG_WAVES), 0, st, p, feats, pstride, keep, img, sigma, g, geo_stride, m->d_scales); \
        else hipLaunchKernelGGL((k_sigma_small_f32<NL_, F_, false, false>), dim3(grid), dim3(64 * SIG_WAVES), 0, st, p, feats, pstride, keep, img, sigma, g, geo_stride, m->d_scales);   \
    } while (0)
    if (nl == 3) { if (f32_in) NRF_GO(3, true); else NRF_GO(3, false); }
    else { if (f32_in) NRF_GO(2, true); else NRF_GO(2, false); }
#undef NRF_GO
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf
