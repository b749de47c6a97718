// hash_fast.hip -- the renderer's fast hash-grid encode for the CuHashEmbedder semantics (CuHashEmbedder.cu:8-102),
// F = 2, fp16 table.  Same arithmetic, bit for bit, as k_hash_cu in encode.hip; what changes is WHERE things live:
//
//   * output is LEVEL-MAJOR fp16: feats[level][point] as one half2 (4 B) per (point, level).  A wavefront's store is one
//     contiguous 256-B segment (the reference kernel's out_feat[p][level*2+k] is a 4-B store at a 64-B stride), and the
//     fused MLP kernel reads its layer-0 MFMA operand straight from this layout (4 coalesced dword loads per lane).
//   * levels are pinned to XCDs.  Workgroups are dealt round-robin over the 8 XCDs, so block b runs on the XCD
//     labelled b % 8; giving all blocks with the same label the same ceil(L/8) levels keeps those levels' slice of the
//     table (2 x 2 MiB, overlapping to 3 MiB through the reference's level-offset quirk) resident in that XCD's
//     4 MiB L2 instead of all 32 MiB competing for every L2.  Placement is a speed assumption only.
//   * points are formed on the fly as o + d*z from the packed ray batch (NeRFRenderer.h:419): no pts buffer.
#include "encode.h"
#include "hash_fast.h"

namespace nrf {

__global__ void k_bake_dense(HashParams hp, int l, uint32_t dim, int64_t entries, uint4 *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= entries) return;
    const uint32_t nby = (uint32_t)hp.dense_nby[l], dz = (uint32_t)hp.dense_nbz[l];
    const uint32_t w = (uint32_t)(i & 15);
    const int64_t col = i >> 4;                       // (tile * dz + z)
    const uint32_t z = (uint32_t)(col % dz), tile = (uint32_t)(col / dz);
    const uint32_t x = (tile / nby) * 4 + (w >> 2), y = (tile % nby) * 4 + (w & 3);
    const uint32_t pa = hp.primes[l * 3 + 0], pb = hp.primes[l * 3 + 1], pc = hp.primes[l * 3 + 2];
    const __half *fp = reinterpret_cast<const __half *>(hp.table) + hp.local_idx[l];
    auto row = [&](uint32_t xx, uint32_t zz) -> uint32_t {
        if (xx >= dim || y >= dim || zz >= dim) return 0u;
        const uint32_t hv = ((xx * pa) ^ (y * pb) ^ (zz * pc)) % hp.local_size[l];
        return *reinterpret_cast<const uint32_t *>(fp + (size_t)hv * 2);
    };
    dst[i] = uint4{row(x, z), row(x, z + 1), row(x + 1, z), row(x + 1, z + 1)};
}

// ------------------------------------------------------------------------------------------------
// HashEmbedder (LibTorch) semantics, NeRF.cpp:208-318: fp32 tables [L][2^T][F], floor()ed resolution, int64 hash with the Instant-NGP
// primes, nested-lerp blend with weights from the UNCLAMPED point.  Same arithmetic, bit for bit, as k_hash_ngp in encode.hip.
// Dense image: per vertex (x,y,z), 0 <= x,y,z < res + 2, the fp32 feature pairs of (x,y,z), (x,y,z+1), (x+1,y,z), (x+1,y,z+1): 32 bytes,
// same 4x4 (x,y) tiling as the CuHash image.  Output: level-major fp16 in TWO planes, hi = f16(v) and lo = f16(v - hi): the split-precision
// MLP consumes both (22 significant bits of the fp32 feature), the plain fp16 MLP only the first.
// ------------------------------------------------------------------------------------------------
__global__ void k_bake_dense_ngp(HashParams hp, int l, uint32_t dim, int64_t entries, uint4 *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= entries) return;
    const uint32_t nby = (uint32_t)hp.dense_nby[l], dz = (uint32_t)hp.dense_nbz[l];
    const uint32_t w = (uint32_t)(i & 15);
    const int64_t col = i >> 4;
    const uint32_t z = (uint32_t)(col % dz), tile = (uint32_t)(col / dz);
    const uint32_t x = (tile / nby) * 4 + (w >> 2), y = (tile % nby) * 4 + (w & 3);
    const float *tl = reinterpret_cast<const float *>(hp.table) + (int64_t)l * ((int64_t)1 << hp.log2_t) * 2;
    const uint32_t hmask = (1u << hp.log2_t) - 1u;
    auto row = [&](uint32_t xx, uint32_t zz) -> uint2 {
        if (xx >= dim || y >= dim || zz >= dim) return uint2{0u, 0u};
        const uint32_t hv = (xx ^ (y * 2654435761u) ^ (zz * 805459861u)) & hmask;
        return *reinterpret_cast<const uint2 *>(tl + (size_t)hv * 2);
    };
    const uint2 a = row(x, z), b = row(x, z + 1), c = row(x + 1, z), d = row(x + 1, z + 1);
    dst[i * 2] = uint4{a.x, a.y, b.x, b.y};
    dst[i * 2 + 1] = uint4{c.x, c.y, d.x, d.y};
}

// a / d, correctly rounded, for operands and quotients in the normal range (here: coordinates of a scene box over cell sizes of 1e-3 .. 1): the reciprocal
// refinement and the two quotient corrections of the compiler's own fp32 division, WITHOUT its range scaling and special-case fix-up (v_div_scale x 2,
// v_div_fmas, v_div_fixup) -- those are the identity on such operands.  8 instructions instead of 11; six divisions per point and level.
#ifndef NRF_NGP_FAST_DIV
#define NRF_NGP_FAST_DIV 1
#endif
__device__ __forceinline__ float ngp_div(float a, float d)
{
#if NRF_NGP_FAST_DIV
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
    return q;
#else
    return a / d;
#endif
}

__device__ __forceinline__ void encode_level_ngp(const HashParams &hp, const float (&x)[3], const float (&xc)[3], int l, float (&acc)[2])
{
    float w[3];
    uint32_t idx[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float grid = hp.bias[l * 3 + a];          // (max - min) / res, divided once on the host (wave-uniform: a scalar load instead of a vector division)
        const float fl = floorf(ngp_div(xc[a] - hp.bbox.mn[a], grid));
        idx[a] = (uint32_t)(int32_t)fl;
        const float vmin = fl * grid + hp.bbox.mn[a];
        const float vmax = vmin + grid;
        w[a] = ngp_div(x[a] - vmin, vmax - vmin);
    }
    float2 e[8];
    if (hp.dense_off[l] >= 0) {                 // wave-uniform
        // The four 16-byte loads go through a native vector type: as HIP's uint4 (a struct) they are taken apart into scalar loads, and the compiler then sinks them
        // and the hashed branch's corner loads into ONE sequence of eight 8-byte loads with selected addresses -- 14 gather instructions per level, 12 of them
        // single dwords, and the encode at a third of its speed (16.4 instead of ~5 ms per frame for the twelve coarse levels of the LibTorch-twin scene)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *dp = reinterpret_cast<const u32x4 *>(hp.dense) + hp.dense_off[l] * 2;
        const uint32_t nby = (uint32_t)hp.dense_nby[l], dz = (uint32_t)hp.dense_nbz[l];
        const uint32_t x0 = idx[0], y0 = idx[1], y1 = idx[1] + 1u, z = idx[2];
        const uint32_t tx0 = (x0 >> 2) * nby, ix0 = (x0 & 3u) << 2;
        const size_t e0 = (size_t)((((tx0 + (y0 >> 2)) * dz + z) << 4) | ix0 | (y0 & 3u)) * 2;
        const size_t e1 = (size_t)((((tx0 + (y1 >> 2)) * dz + z) << 4) | ix0 | (y1 & 3u)) * 2;
        const u32x4 q0a = dp[e0], q0b = dp[e0 + 1], q1a = dp[e1], q1b = dp[e1 + 1];
        auto f2 = [](uint32_t u, uint32_t v) { float2 r; __builtin_memcpy(&r.x, &u, 4); __builtin_memcpy(&r.y, &v, 4); return r; };
        // corner index c = 4dx + 2dy + dz; a quad holds (dx,dz) = (0,0),(0,1) | (1,0),(1,1)
        e[0] = f2(q0a.x, q0a.y); e[1] = f2(q0a.z, q0a.w); e[4] = f2(q0b.x, q0b.y); e[5] = f2(q0b.z, q0b.w);
        e[2] = f2(q1a.x, q1a.y); e[3] = f2(q1a.z, q1a.w); e[6] = f2(q1b.x, q1b.y); e[7] = f2(q1b.z, q1b.w);
    } else {
        const float *tl = reinterpret_cast<const float *>(hp.table) + (int64_t)l * ((int64_t)1 << hp.log2_t) * 2;
        const uint32_t hmask = (1u << hp.log2_t) - 1u;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t cx = idx[0] + ((c >> 2) & 1), cy = idx[1] + ((c >> 1) & 1), cz = idx[2] + (c & 1);
            const uint32_t hv = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & hmask;
            e[c] = *reinterpret_cast<const float2 *>(tl + (size_t)hv * 2);
        }
    }
    const float omx = 1.0f - w[0], omy = 1.0f - w[1], omz = 1.0f - w[2];
    // NeRF.cpp:279-298, per feature: mul, mul, add -- one rounding each (the x and y component are independent chains)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto v2 = [](const float2 &t) { return f32x2{t.x, t.y}; };
    const f32x2 c00 = v2(e[0]) * omx + v2(e[4]) * w[0];
    const f32x2 c01 = v2(e[1]) * omx + v2(e[5]) * w[0];
    const f32x2 c10 = v2(e[2]) * omx + v2(e[6]) * w[0];
    const f32x2 c11 = v2(e[3]) * omx + v2(e[7]) * w[0];
    const f32x2 c0 = c00 * omy + c10 * w[1];
    const f32x2 c1 = c01 * omy + c11 * w[1];
    const f32x2 o = c0 * omz + c1 * w[2];
    acc[0] = o.x; acc[1] = o.y;
}

// feats: plane 0 (hi) at feats, plane 1 (lo) at feats + lo_off (0: not wanted); both [L][pstride] half2.
// F32OUT: the unrounded fp32 features instead, as ONE level-major float2 plane at feats (the exact-fp32 coarse pass, sigma_small_f32.hip).
template <int LPT, bool F32OUT = false>
__global__ void __launch_bounds__(256)
k_hash_ngp_lm(HashParams hp, PointSource ps, int64_t p, __half2 *__restrict__ feats, int64_t pstride, int64_t lo_off, uint8_t *__restrict__ keep, int level0,
              float2 *__restrict__ f32_also = nullptr, int64_t f32_stride = 0)
{
    const int level = level0 + blockIdx.y * LPT;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const F3 pt = load_point(ps, i < p ? i : p - 1);
    const float x[3] = {pt.x, pt.y, pt.z};
    float xc[3];
    bool kp = true;
#pragma unroll
    for (int a = 0; a < 3; a++) { xc[a] = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]); kp = kp && (x[a] == xc[a]); }
    __half2 hi[LPT], lo[LPT];
    float2 full[LPT];
#pragma unroll
    for (int j = 0; j < LPT; j++) {
        float acc[2];
        encode_level_ngp(hp, x, xc, level + j, acc);
        const __half h0 = __float2half_rn(acc[0]), h1 = __float2half_rn(acc[1]);
        hi[j] = __halves2half2(h0, h1);
        lo[j] = __halves2half2(__float2half_rn(acc[0] - __half2float(h0)), __float2half_rn(acc[1] - __half2float(h1)));
        full[j] = float2{acc[0], acc[1]};
    }
    // A point outside the box (keep == false) gets ZERO features.  Its sigma is forced to 0 downstream (NeRFRenderer.h:187-188), so its sample weighs exactly 0 and only
    // the FINITENESS of its colour matters -- and this encoder extrapolates with weights from the unclamped point (NeRF.cpp:265-277): half a scene away from the box the
    // finest levels' weights reach 1e5, the feature leaves the fp16 range (inf), the network returns NaN and 0 * NaN poisons the pixel of a ray that merely misses the
    // box (found by tools/scratch/lane_fuzz.py; the reference's fp32 value there is finite and multiplied by a zero weight).  CuHashEmbedder clamps the point first.
    if (!kp) {
#pragma unroll
        for (int j = 0; j < LPT; j++) { hi[j] = __halves2half2(__half(0.0f), __half(0.0f)); lo[j] = hi[j]; full[j] = float2{0.0f, 0.0f}; }
    }
    if (i >= p) return;
#pragma unroll
    for (int j = 0; j < LPT; j++) {
        if constexpr (F32OUT) reinterpret_cast<float2 *>(feats)[(int64_t)(level + j) * pstride + i] = full[j];
        else {
            feats[(int64_t)(level + j) * pstride + i] = hi[j];
            if (lo_off) feats[lo_off + (int64_t)(level + j) * pstride + i] = lo[j];
            if (f32_also) f32_also[(int64_t)(level + j) * f32_stride + i] = full[j];       // the coarse pass of a feature-reusing render: both forms in one encode
        }
    }
    if (level == 0 && keep) keep[i] = kp ? 1 : 0;
}

// (Re)build the dense image of levels [0, nb) after the table or the primes changed.  nb is chosen by a byte budget.
int hash_fast_prepare(nrf_hash *h, size_t budget_bytes, hipStream_t st)
{
    if (h->fast_valid) return NRF_OK;
    HashParams &hp = h->params;
    const int L = h->desc.n_levels;
    for (int l = 0; l < L; l++) hp.dense_off[l] = -1;
    hp.dense = nullptr;
    bool zero_bias = true;
    for (int i = 0; i < L * 3; i++) zero_bias = zero_bias && hp.bias[i] == 0.0f;
    int64_t total = 0;
    int nb = 0;
    const bool ngp = h->desc.mode == NRF_HASH_NGP;
    const size_t entry_bytes = ngp ? 32 : 16;
    if (h->desc.n_features == 2 && (ngp || zero_bias)) {
        for (int l = 0; l < L; l++) {
            const int64_t dim = (int64_t)floorf(hp.level_scale[l]) + 2;
            const int64_t nbx = (dim + 3) / 4, nby = (dim + 3) / 4;
            const int64_t entries = nbx * nby * dim * 16;                 // (x..x+1, z..z+1) quads: 16 bytes (fp16 pairs) / 32 bytes (fp32 pairs)
            if ((size_t)(total + entries) * entry_bytes > budget_bytes || entries >= ((int64_t)1 << 30)) break;
            hp.dense_off[l] = total; hp.dense_nby[l] = (int32_t)nby; hp.dense_nbz[l] = (int32_t)dim;
            total += entries; nb = l + 1;
        }
    }
    if (nb == 0 && h->d_fast) {          // budget 0 (a training loop): give the image back
        NRF_HIP(hipStreamSynchronize(st));
        NRF_HIP(hipFree(h->d_fast));
        h->d_fast = nullptr; h->fast_bytes = 0;
    }
    bool fresh_image = false;
    if (nb > 0) {
        const size_t need = (size_t)total * entry_bytes;
        // (re)allocate when the image grew -- or shrank to less than half (a training loop lowers the budget to a few coarse levels: the gigabytes go back)
        if (h->fast_bytes < need || h->fast_bytes / 2 > need) {
            if (h->d_fast) { NRF_HIP(hipStreamSynchronize(st)); NRF_HIP(hipFree(h->d_fast)); }
            h->d_fast = nullptr; h->fast_bytes = 0;
            NRF_HIP(hipMalloc(&h->d_fast, need));
            h->fast_bytes = need;
            fresh_image = true;
        }
        hp.dense = h->d_fast;
        for (int l = 0; l < nb; l++) {
            const int64_t dim = (int64_t)floorf(hp.level_scale[l]) + 2;
            const int64_t entries = (l + 1 < nb ? hp.dense_off[l + 1] : total) - hp.dense_off[l];
            if (ngp) hipLaunchKernelGGL(k_bake_dense_ngp, dim3((unsigned)ceil_div(entries, 256)), dim3(256), 0, st, hp, l, (uint32_t)dim, entries,
                                        reinterpret_cast<uint4 *>(h->d_fast) + hp.dense_off[l] * 2);
            else hipLaunchKernelGGL(k_bake_dense, dim3((unsigned)ceil_div(entries, 256)), dim3(256), 0, st, hp, l, (uint32_t)dim, entries,
                                    reinterpret_cast<uint4 *>(h->d_fast) + hp.dense_off[l]);
            NRF_LAUNCH_CHECK();
        }
    }
    // A fresh image (model-load time, budget changes): the bake is complete before this returns, whatever stream reads it next.  A re-bake into the SAME image (the
    // per-step table upload of a training loop) stays asynchronous on `st` like every other call: its readers are ordered behind it on that stream (the lanes of
    // nrf_batchify_rays fork from it)
    if (fresh_image) NRF_HIP(hipStreamSynchronize(st));
    h->fast_valid = true;
    h->dense_levels = nb;
    return NRF_OK;
}

// Measured in round 2 (same box, A/B builds): trimming the kernel's vector instructions -- the three box-coordinate divisions by the (uniform) extent replaced by a
// reciprocal multiply + two FMA corrections (bit-identical: compared exhaustively on the host for 8 extents over all normal numerators), the point -> ray index
// division by a magic multiply, the tile-index multiplies by 24-bit ones: ~18 % fewer vector-ALU cycles -- changed NOTHING: 11.76 vs 11.76 ms, 11.58 vs 11.56 ms per
// frame.  The kernel is not bound by instruction issue (although SQ_INSTS_VALU x 4 cycles is 78 % of its duration) but by the gather path; not kept.
// Which point a thread encodes.  Linear (thread t -> point t) puts 64 consecutive samples of ONE ray on a wave: a segment as long as the scene, 20-64 different
// lines per gather instruction at every level but the coarsest.  NRF_HASH_LANE_TILE = R > 1: a wave takes 64 / R consecutive samples of R consecutive rays --
// neighbouring pixels at neighbouring depths, a handful of lines per gather -- while the features still go to column = point index (R runs of 256 / R bytes per
// store instruction), so nothing downstream changes and the features are the same bits.  Measured on the bench frame, same call, ms of hash encode per frame
// (docs/history/profiles/round3/r4a_hash_lane_tile_ab.log): linear 8.54 / 8.20, R = 2: 8.39 / 8.05, R = 4: 8.14 / 8.19, R = 8: 8.81 / 8.44 (9.0 vs 8.7 in another call),
// R = 16: 10.8 -- the gather path does not pay for fewer distinct lines per instruction, the stores pay for more.  Even with the features stored at the thread's own
// (linear) column -- a wrong image, but the cost a layout that followed the tiling would have -- R = 8 takes 8.8-9.1 against 8.4, R = 16 10.4, R = 64 15.8: a wave
// that spans R rays loads R rays' origins and directions instead of one broadcast row.  Linear stays.
#ifndef NRF_HASH_LANE_TILE
#define NRF_HASH_LANE_TILE 1
#endif
__device__ __forceinline__ int64_t lane_tile_index(const PointSource &ps, int64_t p, int64_t t)
{
#if NRF_HASH_LANE_TILE > 1
    constexpr uint32_t R = NRF_HASH_LANE_TILE, SPW = 64 / R;              // rays and samples per wave
    if (ps.pts || (ps.s % SPW) || (p >> 31)) return t;
    const uint32_t s = (uint32_t)ps.s, span = R * s;                       // points of a group of R rays
    const uint32_t whole = ((uint32_t)p / span) * span;                   // points in whole groups
    if ((uint32_t)t >= whole) return t;
    const uint32_t w = (uint32_t)t >> 6, lane = (uint32_t)t & 63u;         // wave of the launch, lane
    const uint32_t wpg = s / SPW;                                         // waves per group: one per run of SPW samples
    const uint32_t grp = w / wpg, jo = w - grp * wpg;
    return (int64_t)((grp * R + lane / SPW) * s + jo * SPW + (lane % SPW));
#else
    return t;
#endif
}

template <int PPT, int GATHER, int LPT, int DENSE = 0>
__device__ __forceinline__ void hash_lm_body(const HashParams &hp, const PointSource &ps, int64_t p, __half2 *__restrict__ feats, int64_t pstride, uint8_t *__restrict__ keep,
                                             int level, int64_t tile)
{
    PointPrep pp[PPT];
    int64_t idx[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) {
        idx[q] = lane_tile_index(ps, p, (tile * PPT + q) * 256 + threadIdx.x);
        const int64_t i = idx[q] < p ? idx[q] : p - 1;
        pp[q] = prep_point(hp, load_point(ps, i));
    }
    __amdgpu_buffer_rsrc_t rsrc = __amdgpu_buffer_rsrc_t();
    if constexpr (GATHER != 0)
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(hp.table), 0, (int)(((size_t)hp.n_levels << hp.log2_t) * 4), 0x00020000);
    // the feature planes as one buffer resource when they end below 4 GB (a chunk's always do; wave-uniform): a store is a 32-bit lane offset + the level's scalar offset
    const bool store32 = NRF_HASH_DENSE_A32 && (uint64_t)hp.n_levels * (uint64_t)pstride * 4u < ((uint64_t)1 << 32);
    __amdgpu_buffer_rsrc_t frs = __amdgpu_buffer_rsrc_t();
    if (store32) frs = __builtin_amdgcn_make_buffer_rsrc(feats, 0, -1, 0x00020000);
    auto store = [&](int j, int q, __half2 v) {
        if (idx[q] < p) {
#if NRF_HASH_DENSE_A32
            if (store32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), frs, (uint32_t)idx[q] << 2, (uint32_t)(level + j) * (uint32_t)pstride * 4u, 0);
            else
#endif
                feats[(int64_t)(level + j) * pstride + idx[q]] = v;
            if (level + j == 0 && keep) keep[idx[q]] = pp[q].keep ? 1 : 0;
        }
    };
    if constexpr (DENSE > 0) {
        // every level of this launch is baked and 32-bit addressable (decided on the host): BATCH = DENSE levels' loads in flight per thread, then their blends, then
        // their stores -- a batch's quads, one level's weights / products and nothing of the other batches live in registers
        constexpr int BATCH = DENSE < LPT ? DENSE : LPT;
        static_assert(PPT == 1, "the straight-line baked instance takes one point per thread");
#pragma unroll
        for (int j0 = 0; j0 < LPT; j0 += BATCH) {
            float qq[BATCH][3];
            hf_u32x4 q0[BATCH], q1[BATCH];
            __half2 o[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; b++) {
                if (j0 + b < LPT) {
                    uint32_t o0, o1; __amdgpu_buffer_rsrc_t lr;
                    dense_level_issue(hp, pp[0], level + j0 + b, qq[b], o0, o1, lr);
                    q0[b] = __builtin_bit_cast(hf_u32x4, __builtin_amdgcn_raw_buffer_load_b128(lr, o0, 0, 0));
                    q1[b] = __builtin_bit_cast(hf_u32x4, __builtin_amdgcn_raw_buffer_load_b128(lr, o1, 0, 0));
                }
            }
#pragma unroll
            for (int b = 0; b < BATCH; b++)
                if (j0 + b < LPT) {
                    o[b] = dense_level_blend(qq[b], q0[b], q1[b]);
                    __builtin_amdgcn_sched_barrier(0);   // one level's 8 weights and 16 products at a time; and the next batch's loads stay behind this batch's blends
                }
#pragma unroll
            for (int b = 0; b < BATCH; b++)
                if (j0 + b < LPT) store(j0 + b, 0, o[b]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        __half2 out[LPT][PPT];
#pragma unroll
        for (int j = 0; j < LPT; j++)
#pragma unroll
            for (int q = 0; q < PPT; q++) out[j][q] = encode_level<GATHER>(hp, pp[q], level + j, rsrc);
#pragma unroll
        for (int j = 0; j < LPT; j++)
#pragma unroll
            for (int q = 0; q < PPT; q++) store(j, q, out[j][q]);
    }
}

// LPT levels per thread (blockIdx.y indexes groups of LPT levels): the coarse levels run at a fixed per-(point, level) instruction
// cost (their lines are cached), a good part of which is forming the point and its box coordinates -- done once for LPT levels.
// the straight-line baked instances (DENSE > 0) keep DENSE levels' quads in registers: without a floor on the waves per SIMD the compiler hoists every load of the
// thread to the front (214 registers for 12 levels: two waves per SIMD)
#ifndef NRF_HASH_DENSE_WAVES
#define NRF_HASH_DENSE_WAVES 8
#endif
template <int PPT, int GATHER, int LPT = 1, int DENSE = 0>
__global__ void __launch_bounds__(256, (DENSE > 0 ? NRF_HASH_DENSE_WAVES : 1))
k_hash_cu_lm(HashParams hp, PointSource ps, int64_t p, __half2 *__restrict__ feats, int64_t pstride, uint8_t *__restrict__ keep, int lpg, int xcd_map,
             int level0)
{
    int level;
    int64_t tile;
    if (xcd_map) {
        const int g = blockIdx.x & 7;
        const int64_t j = blockIdx.x >> 3;
        const int sub = (int)(j % lpg);
        tile = j / lpg;
        // xcd_map 1: XCD g owns levels [g*lpg, (g+1)*lpg);  2: mirrored pairing (cheap coarse level with an expensive fine one)
        if (xcd_map == 1) level = g * lpg + sub;
        else { const int k = sub * 8 + g; level = (sub & 1) ? (hp.n_levels - 1 - (k - 8 * sub) - 8 * (sub >> 1)) : (k - 8 * sub) + 8 * (sub >> 1); }
        if (level >= hp.n_levels || level < 0) return;
    } else {
        level = level0 + blockIdx.y * LPT;
        tile = blockIdx.x;
    }
    hash_lm_body<PPT, GATHER, LPT, DENSE>(hp, ps, p, feats, pstride, keep, level, tile);
}

// per-ray direction features as fp16 rows [n, V] (the MLP kernel's colour-net operand): SH of the ray's view direction
__global__ void k_dirs_f16(int64_t n, int degree, int variant, const float *__restrict__ rays, int stride, __half *__restrict__ out, __half *__restrict__ out_lo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *dp = rays + i * stride + 8;
    float r[64];
    if (variant == NRF_SH_CUDA) sh_cuda(dp[0], dp[1], dp[2], degree, r);
    else sh_libtorch(dp[0], dp[1], dp[2], degree, r);
    const int od = degree * degree;
    for (int k = 0; k < od; k++) {
        const __half hv = __float2half_rn(r[k]);
        out[i * od + k] = hv;
        if (out_lo) out_lo[i * od + k] = __float2half_rn(r[k] - __half2float(hv));
    }
}

int hash_fast_supported(const nrf_hash *h)
{
    if (!h || h->desc.n_features != 2 || !h->table_set) return 0;
    return h->desc.mode == NRF_HASH_NGP ? 1 : (h->primes_set ? 1 : 0);
}

// HashEmbedder-mode level-major encode: hi plane at feats, lo plane at feats + lo_off (0 = none)
int launch_hash_ngp_lm(const nrf_hash *h, const PointSource &ps, int64_t p, __half2 *feats, int64_t pstride, int64_t lo_off, uint8_t *keep, hipStream_t st, bool f32_out,
                       float2 *f32_also, int64_t f32_stride)
{
    if (p == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_HASH, st);
    const int L = h->desc.n_levels;
    const int64_t ntiles = ceil_div(p, 256);
    const int lc = (L * 3 / 4) & ~3;
    // levels per thread of the coarse / fine launch (see launch_hash_lm).  This encoder's lookups are four 16-byte gathers per level out of an 8.7 GB fp32 image -- bench
    // frame of the LibTorch-twin scene, same call, ms of hash encode: 4 + 1 per thread 14.0-14.1, 12 + 2: 15.3, 6 + 2: 14.2, 4 + 2: 14.2, 2 + 1: 14.1
    // (docs/history/profiles/round3/r9e_ngp_levels_per_thread_ab.log; before the dense lookup's loads were repaired -- see encode_level_ngp -- every grouping took 23.6-24.1)
#ifndef NRF_NGP_COARSE_LPT
#define NRF_NGP_COARSE_LPT 4
#endif
#ifndef NRF_NGP_FINE_LPT
#define NRF_NGP_FINE_LPT 1
#endif
    const bool cg = lc > 0 && (lc % NRF_NGP_COARSE_LPT) == 0, fg = ((L - lc) % NRF_NGP_FINE_LPT) == 0;
    const dim3 gc((unsigned)ntiles, (unsigned)(cg ? lc / NRF_NGP_COARSE_LPT : lc / 4)), gf((unsigned)ntiles, (unsigned)(fg ? (L - lc) / NRF_NGP_FINE_LPT : L - lc));
    if (f32_out) {
        if (cg) hipLaunchKernelGGL((k_hash_ngp_lm<NRF_NGP_COARSE_LPT, true>), gc, dim3(256), 0, st, h->params, ps, p, feats, pstride, 0, keep, 0, nullptr, 0);
        else if (lc > 0) hipLaunchKernelGGL((k_hash_ngp_lm<4, true>), gc, dim3(256), 0, st, h->params, ps, p, feats, pstride, 0, keep, 0, nullptr, 0);
        NRF_LAUNCH_CHECK();
        if (fg) hipLaunchKernelGGL((k_hash_ngp_lm<NRF_NGP_FINE_LPT, true>), gf, dim3(256), 0, st, h->params, ps, p, feats, pstride, 0, keep, lc, nullptr, 0);
        else hipLaunchKernelGGL((k_hash_ngp_lm<1, true>), gf, dim3(256), 0, st, h->params, ps, p, feats, pstride, 0, keep, lc, nullptr, 0);
        NRF_LAUNCH_CHECK();
        return NRF_OK;
    }
    if (cg) hipLaunchKernelGGL((k_hash_ngp_lm<NRF_NGP_COARSE_LPT>), gc, dim3(256), 0, st, h->params, ps, p, feats, pstride, lo_off, keep, 0, f32_also, f32_stride);
    else if (lc > 0) hipLaunchKernelGGL((k_hash_ngp_lm<4>), gc, dim3(256), 0, st, h->params, ps, p, feats, pstride, lo_off, keep, 0, f32_also, f32_stride);
    NRF_LAUNCH_CHECK();
    if (fg) hipLaunchKernelGGL((k_hash_ngp_lm<NRF_NGP_FINE_LPT>), gf, dim3(256), 0, st, h->params, ps, p, feats, pstride, lo_off, keep, lc, f32_also, f32_stride);
    else hipLaunchKernelGGL((k_hash_ngp_lm<1>), gf, dim3(256), 0, st, h->params, ps, p, feats, pstride, lo_off, keep, lc, f32_also, f32_stride);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_hash_lm(const nrf_hash *h, const PointSource &ps, int64_t p, __half2 *feats, int64_t pstride, uint8_t *keep, int variant, hipStream_t st,
                   int level_lo, int level_hi)
{
    if (p == 0) return NRF_OK;
    ProfScope prof(NRF_PROF_HASH, st);
    const int L = h->desc.n_levels;
    const int lpg = (L + 7) / 8;
    const int ppt = (variant & 1) ? 2 : 1;
    const int xcd = (variant >> 1) & 3;
    if (level_hi < 0) level_hi = L;
    const int64_t ntiles = ceil_div(p, 256 * ppt);
    dim3 grid = xcd ? dim3((unsigned)(ntiles * lpg * 8)) : dim3((unsigned)ntiles, (unsigned)(level_hi - level_lo));
    const int gather = (variant >> 3) & 3;
    HashParams hpar = h->params;
    if (variant & 32) { for (int l = 0; l < L; l++) hpar.dense_off[l] = -1; }       // tuning: force the hashed lookup everywhere
    // default variant, all levels: the lower three quarters of the pyramid (instruction-bound: their lines are cached) in ONE thread per point, the (gather-bound) finest
    // levels two per thread
    if (variant == 0 && level_lo == 0 && level_hi == L && L >= 8) {
#ifdef NRF_HASH_LC
        const int lc = NRF_HASH_LC;
#else
        const int lc = (L * 3 / 4) & ~3;
#endif       // measured on 16 levels (ms per frame): 0 -> 11.8, 8 -> 11.4, 12 -> 11.1, 16 -> 11.9
        // Both launches as ONE (consecutive workgroups alternating between a four-coarse-levels kind and a one-fine-level kind, so that a CU holds vector-bound and
        // latency-bound waves together) was built and measured, same call: 8.71-8.74 ms per frame against 8.61-8.62 for the two launches
        // (docs/history/profiles/round3/r5d_hash_mixed_launch_ab.log) -- the two kinds wait for the same gather path.  Again with the 12-level coarse kind and two 2-level fine kinds:
        // 7.78-7.79 against 7.65-7.68.
        // ... and the coarse levels ALL in one thread when there are twelve of them (16-level grids): the point and its box coordinates are formed once, and a coarse level
        // costs instructions, not gather latency.  Round 3, same call, ms of hash encode per frame: 4 per thread 8.34-8.42, 6: 7.88-8.11, 12: 7.79-7.86; 14 + 2: 7.95-8.01,
        // 10 + 6: 7.89-7.96 (docs/history/profiles/round3/r6g_hash_levels_per_thread_ab.log)
#ifndef NRF_HASH_COARSE_LPT
#define NRF_HASH_COARSE_LPT 12
#endif
#ifndef NRF_HASH_FINE_LPT
#define NRF_HASH_FINE_LPT 2
#endif
        // ... where those levels are baked.  With the hashed lookups (a training step uploads a new table every step and bakes nothing: 8 four-byte gathers per level into 8
        // lines) the old grouping stays: training step 9.2-9.35 ms with 4 + 1 levels per thread against 9.83-9.87 with 12 + 2 (docs/history/profiles/round3/r7c_train_step_hash_grouping_ab.log)
        bool baked = !(variant & 32);
        for (int l = 0; l < lc && baked; l++) baked = hpar.dense_off[l] >= 0;
        const int clpt = (baked && (lc % NRF_HASH_COARSE_LPT) == 0) ? NRF_HASH_COARSE_LPT : 4;
        const int flpt = (baked && ((L - lc) % NRF_HASH_FINE_LPT) == 0) ? NRF_HASH_FINE_LPT : 1;
        // every level baked AND its image below 4 GB with tile indices below 2^24 (all of 16..1024): the straight-line instances (hash_fast.h, dense_level_issue)
#ifndef NRF_HASH_DENSE_BATCH_COARSE
#define NRF_HASH_DENSE_BATCH_COARSE 3
#endif
#ifndef NRF_HASH_DENSE_BATCH_FINE
#define NRF_HASH_DENSE_BATCH_FINE 2
#endif
        bool straight = baked && NRF_HASH_DENSE_A32 && NRF_HASH_DENSE_BATCH_COARSE > 0 && !(variant & 64);
        for (int l = 0; l < L && straight; l++) {
            const uint64_t nby = (uint64_t)hpar.dense_nby[l], dz = (uint64_t)hpar.dense_nbz[l];
            straight = hpar.dense_off[l] >= 0 && nby * nby * dz < ((uint64_t)1 << 24) && hpar.bias[l * 3] == 0.0f && hpar.bias[l * 3 + 1] == 0.0f && hpar.bias[l * 3 + 2] == 0.0f;
        }
        if (lc > 0) {
            if (clpt == 4) hipLaunchKernelGGL((k_hash_cu_lm<1, 0, 4>), dim3((unsigned)ntiles, (unsigned)(lc / 4)), dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, 0, 0);
            else if (straight) hipLaunchKernelGGL((k_hash_cu_lm<1, 0, NRF_HASH_COARSE_LPT, NRF_HASH_DENSE_BATCH_COARSE>), dim3((unsigned)ntiles, (unsigned)(lc / NRF_HASH_COARSE_LPT)), dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, 0, 0);
            else hipLaunchKernelGGL((k_hash_cu_lm<1, 0, NRF_HASH_COARSE_LPT>), dim3((unsigned)ntiles, (unsigned)(lc / NRF_HASH_COARSE_LPT)), dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, 0, 0);
        }
        NRF_LAUNCH_CHECK();
        if (straight && flpt == NRF_HASH_FINE_LPT) {
            hipLaunchKernelGGL((k_hash_cu_lm<1, 0, NRF_HASH_FINE_LPT, NRF_HASH_DENSE_BATCH_FINE>), dim3((unsigned)ntiles, (unsigned)((L - lc) / NRF_HASH_FINE_LPT)), dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, 0, lc);
            NRF_LAUNCH_CHECK();
            return NRF_OK;
        }
        // the finest levels two per thread (round 3, same call, ms of hash encode per frame: one per thread 8.67-8.73, two 8.47-8.49; with only 8 or 4 levels in the
        // four-per-thread launch 8.47 / 8.68; docs/history/profiles/round3/r6e_hash_fine_levels_per_thread_ab.log)
        if (flpt == 1) hipLaunchKernelGGL((k_hash_cu_lm<1, 0, 1>), dim3((unsigned)ntiles, (unsigned)(L - lc)), dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, 0, lc);
        else hipLaunchKernelGGL((k_hash_cu_lm<1, 0, NRF_HASH_FINE_LPT>), dim3((unsigned)ntiles, (unsigned)((L - lc) / NRF_HASH_FINE_LPT)), dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, 0, lc);
        NRF_LAUNCH_CHECK();
        return NRF_OK;
    }
#define NRF_LM(P, G) hipLaunchKernelGGL((k_hash_cu_lm<P, G>), grid, dim3(256), 0, st, hpar, ps, p, feats, pstride, keep, lpg, xcd, level_lo)
    if (ppt == 1) { if (gather == 0) NRF_LM(1, 0); else if (gather == 1) NRF_LM(1, 1); else if (gather == 2) NRF_LM(1, 2); else NRF_LM(1, 3); }
    else { if (gather == 0) NRF_LM(2, 0); else if (gather == 1) NRF_LM(2, 1); else if (gather == 2) NRF_LM(2, 2); else NRF_LM(2, 3); }
#undef NRF_LM
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_dirs_f16(const float *rays, int stride, int64_t n, int degree, int variant, __half *out, __half *out_lo, hipStream_t st)
{
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_dirs_f16, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, degree, variant, rays, stride, out, out_lo);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

// Debug / tuning entry (not part of the public header): level-major encode of explicit points with a kernel variant.

extern "C" NRF_API int nrf_dbg_hash_lm(const nrf_hash *h, const float *d_x, int64_t p, int variant, int level_lo, int level_hi, void *d_feats, uint8_t *d_keep, void *stream)
{
    NRF_CHECK_ARG(h && d_x && d_feats && p >= 0, "nrf_dbg_hash_lm: bad argument");
    NRF_CHECK_ARG(hash_fast_supported(h) && h->desc.mode == NRF_HASH_CU, "nrf_dbg_hash_lm: needs a CuHashEmbedder-mode grid with F = 2, table and primes set");
    PointSource ps{d_x, nullptr, nullptr, 0, 1};
    return launch_hash_lm(h, ps, p, reinterpret_cast<__half2 *>(d_feats), p, d_keep, variant, as_stream(stream), level_lo, level_hi);
}
