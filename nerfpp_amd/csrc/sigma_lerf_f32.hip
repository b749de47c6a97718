// sigma_lerf_f32.hip -- the density net of LeRFImpl::forward (LeRF.cpp:86-95: sigma_le = SigmaLENet(x)[..., 0], main.cpp:203-213 sizes 128 -> 256 -> 1 + 32)
// in EXACT fp32 on the matrix cores, for the coarse pass of the LeRF render (LeRFRenderer.cpp:139-170).
//
// Why: the coarse pass's sigma_le picks the fine sample set through searchsorted on CDF plateaus (Sampler.h:6-43), a discontinuous function; evaluated in the
// split (hi + lo fp16) arithmetic of the timed mode, the fine depths matched the fp32 stage path on only ~80 % of the rays.  v_mfma_f32_32x32x2_f32 is, bit for
// bit, the ascending-k fmaf chain of NRF_PREC_F32 / the oracle (sigma_small_f32.hip, tools/scratch/mfma_f32_probe.hip), so sigma_le here EQUALS the parity
// mode's and so does the sample set.  CuHashEmbedder features are exact fp16 numbers (CuHashEmbedder.cu:95): the level-major fp16 table loses nothing.
//
// Formulation (as sigma_small_f32.hip): layer 0 transposed, H^T [256 neurons x points] = W0 [256 x 128] . X^T, A = 32 neurons x 2 k, B = 2 k x 32 points, 64
// ascending k-steps through one accumulator per (m-tile, point tile).  Row i of an m-tile carries neuron 2(4(i/8) + i%4) + (i/4)%2 of its 32, so that register q of
// lane half hh of a finished D tile is neuron 2q + hh: after the ReLU it IS the B operand of k-step q of the next layer, in natural ascending k.
// The 256-wide hidden layer is never held whole: m-tiles are produced in ascending order and consumed at once --
//   * sigma (output row 0) is a 256-term chain per point on the vector ALUs: one v_permlane32_swap per register pair of the wave's two point tiles hands lane l
//     both parities of point l, and the lane runs fmaf over k = 32 mt .. 32 mt + 31 in ascending order before the next m-tile arrives;
//   * geo (output rows 1..32, optional) accumulates in ONE matrix tile per point tile: the 16 k-steps a finished m-tile supplies are issued right away, so the
//     tile's own sum also runs in ascending k.  Its rows are ordered [geo31, geo0 .. geo30]: registers 8s..8s+7 then hold exactly the values of operand fragment s of
//     LE0's chained input cat[sigma, geo0..30] (mlp_lerf_net.h: perm_row), with sigma patched into the slot of row 0 and geo31 moved to fragment 2 -- the (hi, lo)
//     planes kernel B of mlp_lerf_split_mfma.hip reads (Args::geo), here from exact fp32 values instead of split-arithmetic ones.
// W0 (128 KB of fp32 fragments) lives in LDS for the life of a persistent 8-wave workgroup (two waves per SIMD: one's vector work under the other's matrix
// chain); the geo tile's 32 KB of fragments are read from global memory (L1 / L2 resident) four k-steps per load.
#include "mlp_lerf_net.h"

namespace nrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace lsig {

using lerf::f32x16;
using lerf::half8;

constexpr int WAVES = 8;
constexpr int BLOCK_PTS = 64 * WAVES;
constexpr int W0_F4 = 8 * 16 * 64;                 // [mt 8][g = ks / 4, 16][lane 64] float4
constexpr int W1G_F4 = 64 * 64;                    // [g1 = ks1 / 4, 64][lane 64] float4: the geo tile
constexpr size_t LDS_BYTES = (size_t)W0_F4 * 16 + 256 * 4;

__host__ __device__ inline int row_neuron(int i) { return 2 * (4 * (i >> 3) + (i & 3)) + ((i >> 2) & 1); }

__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}

// image: W0 fragments | w1 row 0 (sigma) [256] | geo tile fragments
template <bool GEO>
__global__ void __launch_bounds__(64 * WAVES)
k_lerf_sigma_f32(int64_t npts, const _Float16 *__restrict__ x_lm, int64_t pstride, const uint8_t *__restrict__ keep, const float *__restrict__ image,
                 float *__restrict__ sigma, half8 *__restrict__ geo, int64_t geo_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *wl = reinterpret_cast<f32x4 *>(smem);
    float *wlast = reinterpret_cast<float *>(smem + (size_t)W0_F4 * 16);
    for (int i = threadIdx.x; i < W0_F4; i += blockDim.x) wl[i] = reinterpret_cast<const f32x4 *>(image)[i];
    if (threadIdx.x < 256) wlast[threadIdx.x] = image[(size_t)W0_F4 * 4 + threadIdx.x];
    __syncthreads();
    const f32x4 *w1g = reinterpret_cast<const f32x4 *>(image + (size_t)W0_F4 * 4 + 256);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t nblocks = (npts + BLOCK_PTS - 1) / BLOCK_PTS;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t p0 = blk * BLOCK_PTS + wave * 64;
        // B operands of layer 0: k-step ks = 4 level + j covers k = 8 level + 2 j + {0, 1}; lane half hh takes feature 2 j + hh of the level's 8
        float x[2][64];
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
            int64_t p = p0 + pt * 32 + r;
            if (p >= npts) p = npts - 1;             // clamp loads; the stores are guarded
#pragma unroll
            for (int lv = 0; lv < 16; lv++) {
                // word j of the 16 bytes holds features 2 j (low half) and 2 j + 1 (high half): the lane half's one is shifted down and converted -- two instructions per
                // value.  Written as `hh ? v[2 j + 1] : v[2 j]` the compiler builds a dynamic vector-element extract: seven v_cndmask per value, 870 per iteration.
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = *reinterpret_cast<const u32x4_t *>(x_lm + ((int64_t)lv * pstride + p) * 8);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t bits = v[j] >> (16 * hh);
                    x[pt][4 * lv + j] = (float)__builtin_bit_cast(_Float16, (uint16_t)bits);                  // exact
                }
            }
        }
        float a = 0.0f;
        f32x16 gacc[2] = {zero, zero};
#pragma unroll 1
        for (int mt = 0; mt < 8; mt++) {                 // a real loop: unrolled, the scheduler hoists eight m-tiles' worth of fragment reads and spills
            f32x16 acc[2];
            const f32x4 *wmt = wl + (size_t)mt * 16 * 64 + lane;
            f32x4 ga[4];                                 // the geo tile's fragments for this m-tile's 16 k-steps: requested now, consumed after layer 0's 128 matrix instructions
            if constexpr (GEO) {
#pragma unroll
                for (int g = 0; g < 4; g++) ga[g] = w1g[(size_t)(4 * mt + g) * 64 + lane];
            }
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const f32x4 a4 = wmt[g * 64];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ks = 4 * g + j;
#pragma unroll
                    for (int pt = 0; pt < 2; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], x[pt][ks], ks == 0 ? zero : acc[pt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the ds_read_b128 of later groups from being hoisted to the top (register pressure)
            }
#pragma unroll
            for (int pt = 0; pt < 2; pt++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[pt][q] = fmaxf(acc[pt][q], 0.0f);            // ReLU (one v_max_f32: built with -fno-honor-nans)
            if constexpr (GEO) {
                // the 16 k-steps of layer 1 this m-tile supplies: k = 32 mt + 2 q + hh, register q of the D tile as it stands
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int pt = 0; pt < 2; pt++) gacc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[g][j], acc[pt][4 * g + j], gacc[pt], 0, 0, 0);
            }
            // sigma: lane l runs the chain of point l of the wave's 64 over this m-tile's 32 neurons, ascending
            const f32x4 *wm = reinterpret_cast<const f32x4 *>(wlast + 32 * mt);
#pragma unroll
            for (int q4 = 0; q4 < 8; q4++) {
                const f32x4 w4 = wm[q4];                 // broadcast read: w1[0][32 mt + 4 q4 ..+3]
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int q = 2 * q4 + e;
                    // vdst = tile-0 register, src = tile-1 register: lanes 32-63 of vdst <-> lanes 0-31 of src
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][q]), __float_as_uint(acc[1][q]), false, false);
                    const float ev = __uint_as_float(sw[0]), od = __uint_as_float(sw[1]);
                    a = __builtin_fmaf(w4[2 * e], ev, a);
                    a = __builtin_fmaf(w4[2 * e + 1], od, a);
                }
            }
        }
        const int64_t p = p0 + lane;
        if (p < npts) sigma[p] = (keep && !keep[p]) ? 0.0f : a;                                  // raw_le[~keep, -1] = 0 (LeRFRenderer.cpp:22-23)
        if constexpr (GEO) {
            // sigma of tile 1's point r sits in lane 32 + r: bring it to lane r (the h = 0 lanes own fragment element (s = 0, j = 0) = row 0)
            const auto sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
            const float a_t1 = __uint_as_float(sa[1]);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int64_t q = p0 + pt * 32 + r;
                const float g31 = gacc[pt][0];                       // tile row 0 carries geo31 (h = 0 lanes)
                f32x16 t = gacc[pt];
                if (hh == 0) t[0] = pt == 0 ? a : a_t1;              // row 0 of LE0's chained operand: sigma (unmasked, as kernel A hands it over)
                union { half8 v; uint32_t u[4]; } fh[3], fl[3];
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        // through a compiler-visible vector instruction (identity on finite values): the asm of split_pair must never read a matrix result directly --
                        // the compiler's hazard padding does not look into it (the other sigma kernels do the same)
                        const float lim = -3.402823466e38f;
                        split_pair(fmaxf(t[8 * s + 2 * j], lim), fmaxf(t[8 * s + 2 * j + 1], lim), fh[s].u[j], fl[s].u[j]);
                    }
                split_pair(hh == 0 ? g31 : 0.0f, 0.0f, fh[2].u[0], fl[2].u[0]);
#pragma unroll
                for (int j = 1; j < 4; j++) { fh[2].u[j] = 0; fl[2].u[j] = 0; }
                if (q < npts) {
#pragma unroll
                    for (int f = 0; f < lerf::GEO_FRAGS; f++) {
                        geo[(((int64_t)(f * 2 + 0) * geo_stride + q) << 1) + hh] = fh[f].v;
                        geo[(((int64_t)(f * 2 + 1) * geo_stride + q) << 1) + hh] = fl[f].v;
                    }
                }
            }
        }
    }
}

}  // namespace lsig

static bool lerf_sigma_f32_supported(const nrf_mlp_small_desc &d)
{
    return d.input_ch == lerf::IN && d.hidden_dim == lerf::HID && d.num_layers == 2 && d.geo_feat_dim == lerf::GEO;
}

// fp32 fragments of the LeRF density net: W0 [8][16][64][4] | w1 row 0 [256] | geo tile [64][64][4] (rows geo31, geo0..geo30)
int mlp_lerf_pack_sigma_f32(nrf_mlp *m, const std::vector<float> &hp)
{
    const auto &d = m->small;
    if (!lerf_sigma_f32_supported(d)) return NRF_OK;
    std::vector<float> img;
    img.reserve((size_t)lsig::W0_F4 * 4 + 256 + (size_t)lsig::W1G_F4 * 4);
    const float *w0 = hp.data();                                   // sigma_le_net_0 [256][128]
    const float *w1 = hp.data() + (size_t)lerf::HID * lerf::IN;    // sigma_le_net_1 [33][256]
    for (int mt = 0; mt < 8; mt++)
        for (int g = 0; g < 16; g++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 4; j++) {
                    const int row = 32 * mt + lsig::row_neuron(lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
                    img.push_back(w0[(size_t)row * lerf::IN + k]);
                }
    for (int k = 0; k < lerf::HID; k++) img.push_back(w1[k]);
    for (int g = 0; g < 64; g++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++) {
                const int i = lane & 31, row = i == 0 ? 32 : i, k = 2 * (4 * g + j) + (lane >> 5);       // output row of cat[sigma, geo0..31]: tile row 0 <- geo31
                img.push_back(w1[(size_t)row * lerf::HID + k]);
            }
    const size_t bytes = img.size() * sizeof(float);
    if (m->d_packed_sigma_f32 && m->packed_sigma_f32_bytes != bytes) { (void)hipFree(m->d_packed_sigma_f32); m->d_packed_sigma_f32 = nullptr; }
    if (!m->d_packed_sigma_f32) NRF_HIP(hipMalloc(&m->d_packed_sigma_f32, bytes));
    m->packed_sigma_f32_bytes = bytes;
    NRF_HIP(hipMemcpy(m->d_packed_sigma_f32, img.data(), bytes, hipMemcpyHostToDevice));
    return NRF_OK;
}

// the same image from the parameter blob on the device (one thread per element; the index arithmetic of the host loops above)
__global__ void k_lsig_fill(const float *__restrict__ hp, float *__restrict__ img)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int A = lsig::W0_F4 * 4, B = 256, Cn = lsig::W1G_F4 * 4;
    if (idx >= A + B + Cn) return;
    const float *w0 = hp, *w1 = hp + (size_t)lerf::HID * lerf::IN;
    float v;
    if (idx < A) {
        const int j = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 15, mt = idx >> 12;
        const int row = 32 * mt + lsig::row_neuron(lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
        v = w0[(size_t)row * lerf::IN + k];
    } else if (idx < A + B) {
        v = w1[idx - A];
    } else {
        const int r = idx - A - B, j = r & 3, lane = (r >> 2) & 63, g = r >> 8;
        const int i = lane & 31, row = i == 0 ? 32 : i, k = 2 * (4 * g + j) + (lane >> 5);
        v = w1[(size_t)row * lerf::HID + k];
    }
    img[idx] = v;
}
static_assert(lsig::W0_F4 * 4 == 8 * 16 * 64 * 4 && lsig::W1G_F4 * 4 == 64 * 64 * 4, "k_lsig_fill's index arithmetic");

int mlp_lerf_pack_sigma_f32_device(nrf_mlp *m, hipStream_t st)
{
    const size_t n = (size_t)lsig::W0_F4 * 4 + 256 + (size_t)lsig::W1G_F4 * 4;
    if (!lerf_sigma_f32_supported(m->small) || !m->d_packed_sigma_f32 || m->packed_sigma_f32_bytes != n * sizeof(float)) return NRF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_lsig_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)m->d_params, reinterpret_cast<float *>(m->d_packed_sigma_f32));
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_lerf_sigma_exact_available(const nrf_mlp *m) { return m && m->family == MLP_LERF && m->d_packed_sigma_f32 != nullptr; }

int nrf_lerf_sigma_exact_lm_strided(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const uint8_t *d_keep, int64_t p, float *d_sigma, void *d_geo,
                                    int64_t geo_stride, void *stream)
{
    NRF_CHECK_ARG(m && d_feats_lm && d_sigma && p >= 0 && pstride >= p && (!d_geo || geo_stride >= p), "nrf_lerf_sigma_exact_lm_strided: bad argument");
    if (!nrf_lerf_sigma_exact_available(m)) {
        set_error("nrf_lerf_sigma_exact_lm_strided: the exact-fp32 matrix-core density pass is built for the LeRF of main.cpp:203-213 (in 128, hidden 256, 2 layers, geo 32)");
        return NRF_ERR_UNSUPPORTED;
    }
    NRF_CHECK_ARG(((reinterpret_cast<uintptr_t>(d_feats_lm) | reinterpret_cast<uintptr_t>(d_geo)) & 15) == 0, "nrf_lerf_sigma_exact_lm_strided: features / geo must be 16-byte aligned");
    if (p == 0) return NRF_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(NRF_PROF_SIGMA, st);
    // one-time per-device setup of the 129 KB dynamic LDS window (cheap and idempotent: done on every call, so every device a process drives is covered)
    NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsig::k_lerf_sigma_f32<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lsig::LDS_BYTES));
    NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsig::k_lerf_sigma_f32<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lsig::LDS_BYTES));
    const int64_t nblocks = ceil_div(p, lsig::BLOCK_PTS);
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);          // persistent: one 8-wave workgroup per CU
    const float *img = reinterpret_cast<const float *>(m->d_packed_sigma_f32);
    const _Float16 *x = reinterpret_cast<const _Float16 *>(d_feats_lm);
    if (d_geo) hipLaunchKernelGGL((lsig::k_lerf_sigma_f32<true>), dim3(grid), dim3(64 * lsig::WAVES), lsig::LDS_BYTES, st, p, x, pstride, d_keep, img, d_sigma,
                                  reinterpret_cast<lerf::half8 *>(d_geo), geo_stride);
    else hipLaunchKernelGGL((lsig::k_lerf_sigma_f32<false>), dim3(grid), dim3(64 * lsig::WAVES), lsig::LDS_BYTES, st, p, x, pstride, d_keep, img, d_sigma,
                            static_cast<lerf::half8 *>(nullptr), (int64_t)0);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // extern "C"
