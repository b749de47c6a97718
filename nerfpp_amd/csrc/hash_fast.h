// hash_fast.h -- level-major fp16 hash encode + per-ray direction features (renderer fast path).
#pragma once
#include "encode.h"

namespace nrf {

int hash_fast_supported(const nrf_hash *h);
int hash_fast_prepare(nrf_hash *h, size_t budget_bytes, hipStream_t st);
// feats: [L][pstride] half2.  variant bit 0: two points per thread; bits 1-2: 0 = levels on grid.y, 1 = XCD-pinned level blocks, 2 = mirrored pairs.
int launch_hash_lm(const nrf_hash *h, const PointSource &ps, int64_t p, __half2 *feats, int64_t pstride, uint8_t *keep, int variant, hipStream_t st,
                   int level_lo = 0, int level_hi = -1);
// out_lo (optional): the rounding residuals r - f16(r), for the split-precision MLP
// f32_out: ONE float2 plane [L][pstride] of the unrounded fp32 features at feats instead (lo_off ignored)
// f32_also (with the (hi, lo) form only): the fp32 features as well, as a level-major float2 plane [L][f32_stride]
int launch_hash_ngp_lm(const nrf_hash *h, const PointSource &ps, int64_t p, __half2 *feats, int64_t pstride, int64_t lo_off, uint8_t *keep, hipStream_t st, bool f32_out = false,
                       float2 *f32_also = nullptr, int64_t f32_stride = 0);
int launch_dirs_f16(const float *rays, int stride, int64_t n, int degree, int variant, __half *out, __half *out_lo, hipStream_t st);

constexpr int HASH_LM_DEFAULT_VARIANT = 0;

// ---------------------------------------------------------------------------------------------------
// device-side lookup of the level-major encode kernel (hash_fast.hip).  The level parameters travel in a struct so that a kernel whose
// lanes work on DIFFERENT levels can use the same code: that was tried as a fused hash + NeRFSmall kernel (each lane looks up exactly the
// 8 levels its layer-0 MFMA fragment holds; no feature buffer) and REJECTED -- at the 2-3 waves per SIMD the MLP's registers allow, the
// gathers lose the memory-level parallelism the stand-alone kernel gets from 8 waves per SIMD: 30.9 ms per frame against 11.5 + 5.5.
// ---------------------------------------------------------------------------------------------------
#ifndef NRF_HASH_DENSE_A32
#define NRF_HASH_DENSE_A32 1
#endif
struct PointPrep {
    float q[3];     // (clamp(x) - min) / (max - min), level independent (CuHashEmbedder.cu:44-46 before * mul)
    bool keep;
};

__device__ __forceinline__ PointPrep prep_point(const HashParams &hp, const F3 &pt)
{
    PointPrep r;
    const float x[3] = {pt.x, pt.y, pt.z};
    r.keep = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]);
        r.keep = r.keep && (x[a] == c);
        r.q[a] = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]);
    }
    return r;
}

// Everything a lookup needs to know about its level (wave-uniform in the encode kernel: scalar registers).
struct LevelParams {
    float mul, bias[3];
    int64_t dense_off;          // >= 0: baked dense image (offset in 16-byte quads), -1: hashed table
    uint32_t nby, dz;
    uint32_t pa, pb, pc, lsz;
    int32_t local_idx;
};

__device__ __forceinline__ LevelParams level_params(const HashParams &hp, int l)
{
    LevelParams p;
    p.mul = hp.level_scale[l];
    p.bias[0] = hp.bias[l * 3]; p.bias[1] = hp.bias[l * 3 + 1]; p.bias[2] = hp.bias[l * 3 + 2];
    p.dense_off = hp.dense_off[l];
    p.nby = (uint32_t)hp.dense_nby[l]; p.dz = (uint32_t)hp.dense_nbz[l];
    p.pa = hp.primes[l * 3 + 0]; p.pb = hp.primes[l * 3 + 1]; p.pc = hp.primes[l * 3 + 2];
    p.lsz = hp.local_size[l];
    p.local_idx = hp.local_idx[l];
    return p;
}

__device__ __forceinline__ LevelParams select_params(bool second, const LevelParams &a, const LevelParams &b)
{
    LevelParams p;
    p.mul = second ? b.mul : a.mul;
#pragma unroll
    for (int i = 0; i < 3; i++) p.bias[i] = second ? b.bias[i] : a.bias[i];
    p.dense_off = second ? b.dense_off : a.dense_off;
    p.nby = second ? b.nby : a.nby; p.dz = second ? b.dz : a.dz;
    p.pa = second ? b.pa : a.pa; p.pb = second ? b.pb : a.pb; p.pc = second ? b.pc : a.pc;
    p.lsz = second ? b.lsz : a.lsz;
    p.local_idx = second ? b.local_idx : a.local_idx;
    return p;
}

// One (point, level): the voxel's 8 table values, then the fp32 blend in the reference's order.
// Dense ("baked") image of a level: for every lattice vertex (x,y,z), 0 <= x,y,z < D = floor(mul)+2, the QUAD of table values of
// (x,y,z), (x,y,z+1), (x+1,y,z), (x+1,y,z+1), copied out of the hashed table once at model load (16 bytes: four half2).  A voxel's
// 8 corners are then 2 sixteen-byte loads (rows y and y+1) instead of 8 four-byte gathers into 8 lines.  Vertices are grouped in
// 4x4 (x,y) tiles (256 B at one z), tiles of one (x,y) column stacked along z, so the two loads of a lookup mostly share a line
// and consecutive samples along a ray walk adjacent ones.  Each lookup returns the same table entries the hash would have selected:
// outputs are bit-identical.  Costs 4x the vertices in memory (4.4 GB at 16..512) -- HBM is 288 GB.
// Measured ladder (ms per 800x800x256 frame, same kernel otherwise): hashed table 30.6 -> (z,z+1) pairs, 4 x 8-B loads 14.8 ->
// quads, 2 x 16-B loads 13.6 -> all 8 corners in one 32-B entry 22.2 (REJECTED: 8.8 GB of image, every lookup its own line, the kernel
// turns HBM-bound).  The kernel sits where the vector-memory request rate and the line traffic balance.
template <int GATHER>
__device__ __forceinline__ __half2 encode_level(const HashParams &hp, const PointPrep &pp, const LevelParams &lp, __amdgpu_buffer_rsrc_t rsrc)
{
    float fr[3];
    uint32_t pos[3];
    float acc[2];
    if (lp.dense_off >= 0) {
        // a baked level has zero bias (hash_fast_prepare) and the point is clamped into the box: q >= 0, so the cell index is the truncating conversion itself and the
        // fraction q - floor(q) -- exact in fp32 -- is v_fract_f32: two instructions per axis instead of floor, convert, subtract (same bits; the level's 3 of ~89)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float q = pp.q[a] * lp.mul;
            pos[a] = (uint32_t)q;
            fr[a] = __builtin_amdgcn_fractf(q);
        }
        const uint4 *dp = reinterpret_cast<const uint4 *>(hp.dense) + lp.dense_off;
        const uint32_t nby = lp.nby, dz = lp.dz;
        const float a = fr[0], b = fr[1], c = fr[2];
        const float oma = 1.0f - a, omb = 1.0f - b, omc = 1.0f - c;
        const uint32_t x0 = pos[0], y0 = pos[1], y1 = pos[1] + 1u, z = pos[2];
        const uint32_t tx0 = (x0 >> 2) * nby, ty0 = y0 >> 2, ty1 = y1 >> 2;
        const uint32_t ix0 = (x0 & 3u) << 2, iy0 = y0 & 3u, iy1 = y1 & 3u;
        const uint32_t i0 = (((tx0 + ty0) * dz + z) << 4) | ix0 | iy0;       // corners (x..x+1, y0, z..z+1)
        const uint32_t i1 = (((tx0 + ty1) * dz + z) << 4) | ix0 | iy1;       // corners (x..x+1, y1, z..z+1)
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t q0, q1;
#if NRF_HASH_DENSE_A32
        if ((uint64_t)nby * nby * dz < ((uint64_t)1 << 24)) {
            // the level's image is below 4 GB (every level of 16..512; wave-uniform): the two loads take a 32-bit byte offset off a buffer resource of the level --
            // one shift each instead of a 64-bit per-lane address (v_lshl_add_u64)
            const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(dp), 0, -1, 0x00020000);
            q0 = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(lr, i0 << 4, 0, 0));
            q1 = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(lr, i1 << 4, 0, 0));
        } else
#endif
        {
            q0 = reinterpret_cast<const u32x4_t *>(dp)[i0];
            q1 = reinterpret_cast<const u32x4_t *>(dp)[i1];
        }
        // blend order k = 4dx + 2dy + dz; a quad holds (dx,dz) = (0,0),(0,1),(1,0),(1,1)
        const uint32_t wv[8] = {q0.x, q0.y, q1.x, q1.y, q0.z, q0.w, q1.z, q1.w};
        // both features of a corner go through the same multiply and the same add: float2 vector arithmetic lets the compiler use the
        // packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, one rounding per lane and op: bit-identical to the scalar form)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        float ws[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float wx = (k & 4) ? a : oma, wy = (k & 2) ? b : omb, wz = (k & 1) ? c : omc;
            ws[k] = wx * wy * wz;
        }
        // NOTE the weights ws[] feeding the asm below are results of PLAIN v_mul_f32 on purpose: written as six packed v_pk_mul_f32 (same bits, six instructions fewer) the
        // kernel is right alone and wrong beside matrix-core kernels on another stream -- packed fp32 runs in the matrix data path, its result latency then depends on
        // other waves, and the compiler's hazard padding does not look into the asm that reads it (tools/scratch/pk_debug3.py)
        // product of a corner's fp16 feature and its weight: v_fma_mix_f32 converts and multiplies in one instruction (fma(f32(h), w, -0) = RN(f32(h) * w), sign of a zero
        // product included) at the issue cost of a plain multiply, where v_cvt_f32_f16 alone costs 1.6 of them (tools/scratch/valu_cost_probe.hip) -- 16 per level
        f32x2 pr[8];
        const float negzero = -0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float p0, p1;
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(p0) : "v"(wv[k]), "v"(ws[k]), "s"(negzero));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(wv[k]), "v"(ws[k]), "s"(negzero));
            pr[k] = f32x2{p0, p1};
        }
        f32x2 s2 = pr[0];
#pragma unroll
        for (int k = 1; k < 8; k++) s2 = s2 + pr[k];
        acc[0] = s2.x; acc[1] = s2.y;
    } else {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float q = pp.q[a] * lp.mul;
            q = q + lp.bias[a];
            const float fl = floorf(q);
            pos[a] = (uint32_t)fl;
            fr[a] = q - fl;
        }
        const __half *fp = reinterpret_cast<const __half *>(hp.table) + lp.local_idx;
        cu_blend<2, GATHER>(fp, pos, fr, lp.pa, lp.pb, lp.pc, lp.lsz, acc, rsrc, (uint32_t)lp.local_idx * 2u);
    }
    return __halves2half2(__float2half_rn(acc[0]), __float2half_rn(acc[1]));
}

// ---- the baked lookup of SEVERAL levels as straight-line code (round 5) ----
// encode_level above decides per level, at run time, between the baked image and the hashed table and between 32- and 64-bit addressing: two wave-uniform branches per
// level, i.e. every level is its own basic block -- its two 16-byte loads are issued, ~20 weight instructions later they are waited for, and nothing of the next level
// can start before.  A thread's levels ran one after the other and the only memory-level parallelism was the SIMD's other waves (2 loads x 8 waves in flight).  When the
// HOST knows that every level of a launch is baked and below 4 GB (hash_fast_prepare's layout; the render path always), the kernel instance below has no such branches:
// the loads of BATCH levels are issued back to back, then their blends follow -- BATCH x 2 loads per wave in flight, same arithmetic, same bits.
struct DenseAddr { uint32_t o0, o1; };

__device__ __forceinline__ void dense_level_issue(const HashParams &hp, const PointPrep &pp, int l, float q[3], uint32_t &off0, uint32_t &off1, __amdgpu_buffer_rsrc_t &lr)
{
    const float mul = hp.level_scale[l];
    uint32_t pos[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { q[a] = pp.q[a] * mul; pos[a] = (uint32_t)q[a]; }
    const uint32_t nby = (uint32_t)hp.dense_nby[l], dz = (uint32_t)hp.dense_nbz[l];
    const uint32_t x0 = pos[0], y0 = pos[1], y1 = pos[1] + 1u, z = pos[2];
    // tile indices stay below 2^24 (checked on the host with the 4 GB bound): the 24-bit multiply is a full-rate instruction, v_mul_lo_u32 is not
    auto mul24 = [](uint32_t v, uint32_t s_uniform) {
        uint32_t r;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(s_uniform), "v"(v));       // inputs: an integer shift / add result and a scalar: no matrix / packed / transcendental producer
        return r;
    };
    const uint32_t tx0 = mul24(x0 >> 2, nby), ty0 = y0 >> 2, ty1 = y1 >> 2;
    const uint32_t ix0 = (x0 & 3u) << 2, iy0 = y0 & 3u, iy1 = y1 & 3u;
    const uint32_t i0 = ((mul24(tx0 + ty0, dz) + z) << 4) | ix0 | iy0;
    const uint32_t i1 = ((mul24(tx0 + ty1, dz) + z) << 4) | ix0 | iy1;
    off0 = i0 << 4; off1 = i1 << 4;
    lr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(reinterpret_cast<const uint4 *>(hp.dense) + hp.dense_off[l]), 0, -1, 0x00020000);
}

typedef uint32_t hf_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __half2 dense_level_blend(const float q[3], const hf_u32x4 q0, const hf_u32x4 q1)
{
    const float a = __builtin_amdgcn_fractf(q[0]), b = __builtin_amdgcn_fractf(q[1]), c = __builtin_amdgcn_fractf(q[2]);
    const float oma = 1.0f - a, omb = 1.0f - b, omc = 1.0f - c;
    const uint32_t wv[8] = {q0.x, q0.y, q1.x, q1.y, q0.z, q0.w, q1.z, q1.w};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float ws[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float wx = (k & 4) ? a : oma, wy = (k & 2) ? b : omb, wz = (k & 1) ? c : omc;
        ws[k] = wx * wy * wz;
    }
    f32x2 pr[8];
    const float negzero = -0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        float p0, p1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(p0) : "v"(wv[k]), "v"(ws[k]), "s"(negzero));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(wv[k]), "v"(ws[k]), "s"(negzero));
        pr[k] = f32x2{p0, p1};
    }
    f32x2 s2 = pr[0];
#pragma unroll
    for (int k = 1; k < 8; k++) s2 = s2 + pr[k];
    return __halves2half2(__float2half_rn(s2.x), __float2half_rn(s2.y));
}

template <int GATHER>
__device__ __forceinline__ __half2 encode_level(const HashParams &hp, const PointPrep &pp, int l, __amdgpu_buffer_rsrc_t rsrc)
{
    return encode_level<GATHER>(hp, pp, level_params(hp, l), rsrc);
}


}  // namespace nrf
