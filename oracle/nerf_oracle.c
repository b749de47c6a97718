/*
 * nerf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C (C99 + optional OpenMP) restatement of the reference's volume-rendering hot
 * path (DeliriumV01D/NeRFpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (nerfpp_amd/, libnerfpp_hip.so)
 * never links, imports or calls it and has no CPU fallback.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  Evaluation rule: fp32, one IEEE rounding per source-level
 * operation, in source order -- the way LibTorch's CPU eager ops evaluate a chain of
 * tensor ops.  Build with -ffp-contract=off (oracle/Makefile does) so the compiler never
 * fuses a*b+c.  Where ATen itself is not sequential-fp32 it is stated:
 *   - cumsum(float) on CPU accumulates in double and rounds each prefix to float
 *     (ATen/native/cpu/ReduceOpsKernel.cpp cumsum -> acc_type<float,false> == double);
 *   - sum(float) on CPU is a vectorised multi-lane cascade whose order depends on the
 *     host ISA (AVX2 vs AVX-512): it has no machine-independent bit pattern.  The oracle
 *     accumulates such sums in double and rounds once, which is within 1 ulp of any of
 *     ATen's orders;
 *   - sin/cos/exp/log/sigmoid on CPU are SLEEF u10 vector kernels; libm and the GPU's OCML each differ
 *     from it (and from each other) in the last ulp.  The oracle and the HIP path therefore share ONE
 *     portable definition, include/nrf_math.h (FMA/mul/add/div + bit casts only, <= 1.5 ulp), so that
 *     they agree bit for bit; against the reference these stages are checked to a few ulp.
 *
 * PINNING: oracle/_ref (the reference's own sources compiled against LibTorch here) emits
 * tests/golden/ (npz files); tests/test_oracle_golden.py checks every function below against
 * them (bit-exact for integer/index outputs and for stages made only of + - * / floor;
 * tolerance stated per test otherwise).  The two CUDA-only units (CuHashEmbedder.cu,
 * CuSHEncoder.cu) cannot be compiled here: orc_hash_cu / orc_sh_cu are
 * RESTATEMENT-PINNED (hand-computed known answers + cross-checks), parity unpinned by
 * any reference run.
 */
#include <math.h>
#include <stdint.h>
#include "nrf_math.h"
#include "nrf_rng.h"   /* portable exp/log/sin/cos: identical bits on CPU and GPU (see the header) */
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#define OMP_FOR _Pragma("omp parallel for schedule(static)")
#else
#define OMP_FOR
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * helpers
 * ------------------------------------------------------------------------------------------ */
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }

/* fp32 -> fp16 bits, round-to-nearest-even (what tensor.to(kFloat16) and CUDA (__half)x do) */
static uint16_t f32_to_f16_bits(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t exp = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
    int32_t e = exp - 127 + 15;
    if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
    return (uint16_t)(sign | half);
}

static float f16_bits_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t mant = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) x = sign;
        else {
            int e = -1;
            do { e++; mant <<= 1; } while (!(mant & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13);
        }
    } else if (exp == 0x1f) x = sign | 0x7f800000u | (mant << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}

ORC_API void orc_f32_to_f16(const float *in, uint16_t *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = f32_to_f16_bits(in[i]);
}
ORC_API void orc_f16_to_f32(const uint16_t *in, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = f16_bits_to_f32(in[i]);
}

/* torch::linspace(start, end, steps, kFloat) as ATen's CPU kernel evaluates it
 * (ATen/native/cpu/RangeFactoriesKernel.cpp linspace_kernel): step = (end-start)/(steps-1) in
 * fp32; element i is start + step*i in the first half and end - step*(steps-1-i) in the second,
 * and the kernel is built with FMA contraction, so each element is ONE fused rounding:
 * fma(step, i, start) / fma(-step, steps-1-i, end).  Verified bit for bit against torch 2.10
 * (tests/golden sample_pdf aux_t64 / aux_u_128 / aux_t192).  The renderer takes its t / u tables
 * as INPUTS (the LibTorch / PyTorch host passes torch::linspace output); this is for hosts that
 * have no torch at hand and for the oracle's own drivers. */
ORC_API void orc_linspace(float start, float end, int steps, float *out)
{
    if (steps == 1) { out[0] = start; return; }
    float step = (end - start) / (float)(steps - 1);
    int halfway = steps / 2;
    for (int i = 0; i < steps; i++)
        out[i] = (i < halfway) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - i - 1), end);
}

/* torch::sum(x, -1) over a contiguous fp32 row of n < 1024 elements as ATen's CPU kernel evaluates
 * it (ATen/native/cpu/SumKernel.cpp: vectorized_inner_sum -> row_sum -> multi_row_sum, ilp 4):
 *   nv = n / vec full vectors; lane-wise: four interleaved vector accumulators P_k = sum_i V[4i+k]
 *   (i < nv/4, sequential), then P_0 += V[j] for the nv%4 left-over vectors, then P_0 += P_1,P_2,P_3;
 *   scalar: acc = sum of the n%vec tail elements (sequential), then acc += lane[0..vec-1] of P_0.
 * vec = 8: ATen registers the AVX2 build of this kernel for AVX-512 hosts too (REGISTER_DISPATCH
 * without ALSO_REGISTER_AVX512_DISPATCH), so 8 fp32 lanes is what any AVX2-or-newer x86 host runs;
 * verified bit for bit against torch 2.10 on an AVX-512 machine.  vec = 0 selects the
 * order-free definition: accumulate in double, round once.  Used where a sum feeds an integer
 * decision (SamplePDF's pdf normaliser -> searchsorted indices). */
static float aten_row_sum_f32(const float *x, int n, int vec)
{
    if (vec <= 0 || n < vec) {
        if (vec <= 0) { double s = 0.0; for (int i = 0; i < n; i++) s += (double)x[i]; return (float)s; }
        /* n < vec: scalar_inner_sum -> row_sum over scalars with ilp 4 */
        float ps[4] = {0, 0, 0, 0};
        int q = n / 4;
        for (int i = 0; i < q; i++) for (int k = 0; k < 4; k++) ps[k] += x[i * 4 + k];
        for (int i = q * 4; i < n; i++) ps[0] += x[i];
        for (int k = 1; k < 4; k++) ps[0] += ps[k];
        return ps[0];
    }
    float P[4][64];
    memset(P, 0, sizeof(P));
    int nv = n / vec, q = nv / 4;
    for (int i = 0; i < q; i++)
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < vec; l++) P[k][l] += x[(i * 4 + k) * vec + l];
    for (int j = q * 4; j < nv; j++)
        for (int l = 0; l < vec; l++) P[0][l] += x[j * vec + l];
    for (int k = 1; k < 4; k++)
        for (int l = 0; l < vec; l++) P[0][l] += P[k][l];
    float acc = 0.0f;
    for (int i = nv * vec; i < n; i++) acc += x[i];
    for (int l = 0; l < vec; l++) acc += P[0][l];
    return acc;
}
ORC_API float orc_aten_row_sum(const float *x, int n, int vec) { return aten_row_sum_f32(x, n, vec); }

/* ------------------------------------------------------------------------------------------
 * R1  GetDirections + GetRays            RayUtils.h:5-46
 *     dir = ((x-cx)/fx, -(y-cy)/fy, -1);  d_i = sum_j dir_j * c2w[i][j] (torch::sum over 3,
 *     left to right);  o = c2w[:3,3];  cone_angle = ((1/fx + 1/fy)/2) * 1.1
 *     Pixel <-> ray index is row-major, y outer, x inner.  Pixel coordinates come from
 *     linspace(0, n-1, n) whose values are exact integers in fp32.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_get_rays(int h, int w, const float *K, const float *c2w /*[3,4]*/, int row0, int rows,
                          float *o, float *d, float *cone_angle)
{
    const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    OMP_FOR
    for (int y = row0; y < row0 + rows; y++)
        for (int x = 0; x < w; x++) {
            float dir[3];
            dir[0] = ((float)x - cx) / fx;
            dir[1] = -((float)y - cy) / fy;
            dir[2] = -1.0f;
            int64_t r = (int64_t)(y - row0) * w + x;
            for (int i = 0; i < 3; i++) {
                float acc = dir[0] * c2w[i * 4 + 0];
                acc = acc + dir[1] * c2w[i * 4 + 1];
                acc = acc + dir[2] * c2w[i * 4 + 2];
                d[r * 3 + i] = acc;
                o[r * 3 + i] = c2w[i * 4 + 3];
            }
        }
    if (cone_angle) {
        /* RayUtils.h:35-43: tensors stay fp32 (a 0-dim fp32 tensor times a double scalar stays fp32) */
        float px = 1.0f / fx, py = 1.0f / fy;
        float avg = (px + py) / 2.0f;
        *cone_angle = avg * 1.1f;
    }
}

/* N2  NeRFDataset::GetRayBatch             NeRFDataset.cpp:109-145 -- rays through the pixels (rand_h, rand_w) of one view.
 *     The same direction arithmetic as GetRays (pixel coordinates converted to fp32, (w - cx)/fx, -(h - cy)/fy, -1, three products
 *     summed left to right), so a batch drawn at grid coordinates equals the rows of GetRays bit for bit; cone_angle here is the plain
 *     mean pixel size (1/fx + 1/fy)/2 (no x1.1 factor, unlike RayUtils.h:43).  RESTATEMENT-PINNED via GetRays (NeRFDataset.cpp needs
 *     OpenCV loaders and does not build here). */
ORC_API void orc_ray_batch(const float *K, const float *c2w, const int64_t *rand_h, const int64_t *rand_w, int64_t n, float *o, float *d, float *cone_angle)
{
    const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    OMP_FOR
    for (int64_t r = 0; r < n; r++) {
        float dir[3];
        dir[0] = ((float)rand_w[r] - cx) / fx;
        dir[1] = -((float)rand_h[r] - cy) / fy;
        dir[2] = -1.0f;
        for (int i = 0; i < 3; i++) {
            float acc = dir[0] * c2w[i * 4 + 0];
            acc = acc + dir[1] * c2w[i * 4 + 1];
            acc = acc + dir[2] * c2w[i * 4 + 2];
            d[r * 3 + i] = acc;
            o[r * 3 + i] = c2w[i * 4 + 3];
        }
    }
    if (cone_angle) {
        /* 1.0 / fx with fx a 0-dim fp32 tensor: fp32 */
        float px = 1.0f / fx, py = 1.0f / fy;
        *cone_angle = (px + py) / 2.0f;
    }
}

/* target_s = CurrentImage.index({rand_h, rand_w})        NeRFDataset.cpp:156 ; image [H, W, C] */
ORC_API void orc_gather_pixels(const float *image, int h, int w, int c, const int64_t *rand_h, const int64_t *rand_w, int64_t n, float *out)
{
    for (int64_t r = 0; r < n; r++)
        memcpy(out + r * c, image + ((int64_t)rand_h[r] * w + rand_w[r]) * c, sizeof(float) * c);
}

/* CalculateBounds (NeRFDataset.cpp:44-65): the centre crop used for the first PrecorpIters iterations; inclusive bounds */
ORC_API void orc_precrop_bounds(int h, int w, int iter, int precrop_iters, float precrop_frac, int *out /* h_start, h_end, w_start, w_end */)
{
    if (iter < precrop_iters) {
        int dh = (int)(h / 2 * precrop_frac), dw = (int)(w / 2 * precrop_frac);
        out[0] = h / 2 - dh; out[1] = h / 2 + dh - 1; out[2] = w / 2 - dw; out[3] = w / 2 + dw - 1;
    } else { out[0] = 0; out[1] = h - 1; out[2] = 0; out[3] = w - 1; }
}

/* Random pixel coordinates of a batch (NeRFDataset.cpp:154-155 draws torch::randint): here a pure function of (seed, iter, element):
 * lo + floor(u32 * (hi - lo + 1) / 2^32) with the counter generator of include/nrf_rng.h, streams 16 (rows) and 17 (columns). */
ORC_API void orc_rand_pixels(uint64_t seed, int64_t iter, int h_start, int h_end, int w_start, int w_end, int64_t n, int64_t *rand_h, int64_t *rand_w)
{
    const uint64_t rh = (uint64_t)(h_end - h_start + 1), rw = (uint64_t)(w_end - w_start + 1);
    for (int64_t k = 0; k < n; k++) {
        const uint64_t idx = (uint64_t)iter * (uint64_t)n + (uint64_t)k;
        rand_h[k] = h_start + (int64_t)(((uint64_t)nrf_rng_u32(seed, 16u, idx) * rh) >> 32);
        rand_w[k] = w_start + (int64_t)(((uint64_t)nrf_rng_u32(seed, 17u, idx) * rw) >> 32);
    }
}

/* R2  NDCRays                              RayUtils.h:49-83  (near = 1 in Render(), NeRFRenderer.h:567) */
ORC_API void orc_ndc_rays(int h, int w, float focal, float near_, const float *o, const float *d, int64_t n,
                          float *oo, float *od)
{
    /* the python-double constants (-1./(w/(2.*focal)), 2.*near ...) are evaluated in double and
       applied to fp32 tensors as scalars: the tensor op rounds the double scalar to fp32 first. */
    const float sx = (float)(-1. / ((double)w / (2. * (double)focal)));
    const float sy = (float)(-1. / ((double)h / (2. * (double)focal)));
    const float two_near = (float)(2. * (double)near_);
    const float m_two_near = (float)(-2. * (double)near_);
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        float ox = o[i * 3], oy = o[i * 3 + 1], oz = o[i * 3 + 2];
        float dx = d[i * 3], dy = d[i * 3 + 1], dz = d[i * 3 + 2];
        float t = -(near_ + oz) / dz;
        ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
        oo[i * 3 + 0] = sx * ox / oz;
        oo[i * 3 + 1] = sy * oy / oz;
        oo[i * 3 + 2] = 1.0f + two_near / oz;
        od[i * 3 + 0] = sx * (dx / dz - ox / oz);
        od[i * 3 + 1] = sy * (dy / dz - oy / oz);
        od[i * 3 + 2] = m_two_near / oz;
    }
}

/* R3  IntersectWithAABB                    RayUtils.h:87-126
 *     inv = 1/(d + 1e-6)  (1e-6 is a double scalar -> rounded to fp32 by the tensor op) */
ORC_API void orc_aabb(const float *o, const float *d, const float *bbox, int64_t n, float near_plane,
                      float *nears, float *fars)
{
    const float eps = (float)1e-6;
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        float tmin[3], tmax[3];
        for (int a = 0; a < 3; a++) {
            float inv = 1.0f / (d[i * 3 + a] + eps);
            float t1 = (bbox[a] - o[i * 3 + a]) * inv;
            float t2 = (bbox[3 + a] - o[i * 3 + a]) * inv;
            tmin[a] = f_min(t1, t2);
            tmax[a] = f_max(t1, t2);
        }
        float nr = f_max(f_max(tmin[0], tmin[1]), tmin[2]);
        float fr = f_min(f_min(tmax[0], tmax[1]), tmax[2]);
        nr = f_max(nr, near_plane);             /* clamp_min */
        fr = f_max(fr, nr + 1e-6f);
        nears[i] = nr; fars[i] = fr;
    }
}

/* R6  stratified depths (Perturb == 0)     NeRFRenderer.h:393-402
 *     z = near*(1-t) + far*t ; lindisp: safe_inv(safe_inv(near)*(1-t) + safe_inv(far)*t) */
static inline float safe_inv(float x) { return (fabsf(x) < 1e-8f) ? (1.0f / 1e-8f) : (1.0f / x); }

ORC_API void orc_z_vals(const float *nears, const float *fars, const float *t, int64_t n, int s, int lindisp, float *z)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < s; j++) {
            float omt = 1.0f - t[j];
            if (!lindisp) z[i * s + j] = nears[i] * omt + fars[i] * t[j];
            else z[i * s + j] = safe_inv(safe_inv(nears[i]) * omt + safe_inv(fars[i]) * t[j]);
        }
}

/* pts = o + d*z                            NeRFRenderer.h:419 */
ORC_API void orc_points(const float *o, const float *d, const float *z, int64_t n, int s, float *pts)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < s; j++)
            for (int a = 0; a < 3; a++)
                pts[(i * s + j) * 3 + a] = o[i * 3 + a] + d[i * 3 + a] * z[i * s + j];
}

/* Counter-based draws (include/nrf_rng.h): element k of the array gets index idx0 + k. */
ORC_API void orc_rng_uniform(uint64_t seed, uint32_t stream, uint64_t idx0, int64_t count, float *out)
{
    OMP_FOR
    for (int64_t k = 0; k < count; k++) out[k] = nrf_rng_uniform(seed, stream, idx0 + (uint64_t)k);
}

ORC_API void orc_rng_normal(uint64_t seed, uint32_t stream, uint64_t idx0, int64_t count, float *out)
{
    OMP_FOR
    for (int64_t k = 0; k < count; k++) out[k] = nrf_rng_normal(seed, stream, idx0 + (uint64_t)k);
}

/* R6  stratified jitter (Perturb > 0)      NeRFRenderer.h:404-417
 *     mids = .5*(z[1:]+z[:-1]) ; upper = [mids, z[-1]] ; lower = [z[0], mids] ; interval = upper-lower ;
 *     z' = lower + (interval > 1e-8 ? interval*t_rand : 0)           t_rand: the [n,s] uniform draws */
ORC_API void orc_jitter_z(const float *z, const float *t_rand, int64_t n, int s, float *out)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        const float *zi = z + i * s;
        for (int k = 0; k < s; k++) {
            float upper = (k < s - 1) ? 0.5f * (zi[k + 1] + zi[k]) : zi[s - 1];
            float lower = (k > 0) ? 0.5f * (zi[k] + zi[k - 1]) : zi[0];
            float interval = upper - lower;
            out[i * s + k] = lower + ((interval > 1e-8f) ? interval * t_rand[i * s + k] : 0.0f);
        }
    }
}

/* R7  TangentScatter                       NeRFRenderer.h:307-362
 *     dn = d / max(|d|, 1e-8) ; up = the axis dn is least aligned with (x if |dx| strictly smallest, else y if |dy|
 *     strictly smallest, else z) ; tangent = normalize(dn x up) ; bitangent = normalize(dn x tangent) ;
 *     r = sqrt(clamp(U1, 1e-8, 1-1e-8)) ; theta = fmod(U2*2*pi, 2*pi) ; offset = tangent*r*cos + bitangent*r*sin ;
 *     pts += offset * (cone_angle*z) ; clamp to the bounding box when one is given.
 *     u_r, u_theta: the [n,s] uniform draws (torch::rand at :342-343). */
static inline void normalize3(float *v)
{
    float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    nrm = f_max(nrm, 1e-8f);
    v[0] = v[0] / nrm; v[1] = v[1] / nrm; v[2] = v[2] / nrm;
}

static inline void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

ORC_API void orc_tangent_scatter(const float *pts, const float *z, float cone_angle, const float *rays_d, const float *u_r,
                                 const float *u_theta, const float *bbox /* [6] or NULL */, int64_t n, int s, float *out)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        float dn[3] = {rays_d[i * 3], rays_d[i * 3 + 1], rays_d[i * 3 + 2]};
        normalize3(dn);
        const float ax = fabsf(dn[0]), ay = fabsf(dn[1]), az = fabsf(dn[2]);
        const int mx = (ax < ay) && (ax < az), my = (ay < ax) && (ay < az);
        float up[3] = {mx ? 1.0f : 0.0f, (!mx && my) ? 1.0f : 0.0f, (!mx && !my) ? 1.0f : 0.0f};
        float tg[3], bt[3];
        cross3(dn, up, tg); normalize3(tg);
        cross3(dn, tg, bt); normalize3(bt);
        for (int k = 0; k < s; k++) {
            const int64_t q = i * s + k;
            float u1 = u_r[q];
            u1 = f_min(f_max(u1, 1e-8f), 1.0f - 1e-8f);
            const float r = sqrtf(u1);
            const float theta = fmodf(u_theta[q] * 2.0f * (float)M_PI, (float)(2.0f * M_PI));
            float sn, cs;
            nrf_sincosf(theta, &sn, &cs);
            const float ox = r * cs, oy = r * sn;
            const float radius = cone_angle * z[q];
            for (int a = 0; a < 3; a++) {
                float v = pts[q * 3 + a] + (tg[a] * ox + bt[a] * oy) * radius;
                if (bbox) v = f_min(f_max(v, bbox[a]), bbox[3 + a]);
                out[q * 3 + a] = v;
            }
        }
    }
}

/* Stochastic preconditioning + ReflectBoundary      NeRFRenderer.h:433-443, :285-304
 *     pts += noise*alpha ; x = (pts-min)/(max-min) ; x = fmod(x, 2) ; x > 1 -> 2 - x ; pts = x*(max-min)+min
 *     (fmod keeps the sign, so points pushed below the box minimum stay outside: reference behaviour). */
ORC_API void orc_precondition(const float *pts, const float *noise, float alpha, const float *bbox, int64_t p, float *out)
{
    OMP_FOR
    for (int64_t q = 0; q < p; q++)
        for (int a = 0; a < 3; a++) {
            const float ext = bbox[3 + a] - bbox[a];
            float v = pts[q * 3 + a] + noise[q * 3 + a] * alpha;
            float x = (v - bbox[a]) / ext;
            x = fmodf(x, 2.0f);
            if (x > 1.0f) x = 2.0f - x;
            out[q * 3 + a] = x * ext + bbox[a];
        }
}

/* ------------------------------------------------------------------------------------------
 * E1  sinusoidal positional encoding       NeRF.cpp:4-39
 *     freq_i = powf(2, maxlog2/(n-1)*i) with maxlog2 = n-1 (NeRF.h:22-23);
 *     out = [x, sin(x f0), cos(x f0), sin(x f1), ...], each block 3 wide.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_pe(const float *x, int64_t p, int nfreq, float *out)
{
    const int od = 3 + 6 * nfreq;
    float freqs[64];
    const float maxf = (float)(nfreq - 1);
    for (int i = 0; i < nfreq; i++) freqs[i] = powf(2.0f, maxf / (float)(nfreq - 1) * (float)i);
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float *o = out + i * od;
        for (int a = 0; a < 3; a++) o[a] = x[i * 3 + a];
        for (int f = 0; f < nfreq; f++)
            for (int a = 0; a < 3; a++) {
                float v = x[i * 3 + a] * freqs[f];
                o[3 + f * 6 + a] = nrf_sinf(v);
                o[3 + f * 6 + 3 + a] = nrf_cosf(v);
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * S2  LibTorch spherical harmonics, degree <= 5     NeRF.cpp:131-201, constants NeRF.h:83-111
 *     Python-style literals (2.0, 3, 4, 7, 35 ...) are scalars applied to fp32 tensors, each
 *     tensor-scalar op rounds once; products are evaluated left to right.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_sh_libtorch(const float *dirs, int64_t p, int degree, float *out)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f,
                         1.445305721320277f, -0.5900435899266435f};
    const float C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f, 0.10578554691520431f,
                         -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};
    const int od = degree * degree;
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        const float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
        float *r = out + i * od;
        r[0] = C0;
        if (degree <= 1) continue;
        r[1] = -C1 * y; r[2] = C1 * z; r[3] = -C1 * x;
        if (degree <= 2) continue;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r[4] = C2[0] * xy;
        r[5] = C2[1] * yz;
        r[6] = C2[2] * (2.0f * zz - xx - yy);
        r[7] = C2[3] * xz;
        r[8] = C2[4] * (xx - yy);
        if (degree <= 3) continue;
        r[9] = C3[0] * y * (3.0f * xx - yy);
        r[10] = C3[1] * xy * z;
        r[11] = C3[2] * y * (4.0f * zz - xx - yy);
        r[12] = C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        r[13] = C3[4] * x * (4.0f * zz - xx - yy);
        r[14] = C3[5] * z * (xx - yy);
        r[15] = C3[6] * x * (xx - 3.0f * yy);
        if (degree <= 4) continue;
        r[16] = C4[0] * xy * (xx - yy);
        r[17] = C4[1] * yz * (3.0f * xx - yy);
        r[18] = C4[2] * xy * (7.0f * zz - 1.0f);
        r[19] = C4[3] * yz * (7.0f * zz - 3.0f);
        r[20] = C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
        r[21] = C4[5] * xz * (7.0f * zz - 3.0f);
        r[22] = C4[6] * (xx - yy) * (7.0f * zz - 1.0f);
        r[23] = C4[7] * xz * (xx - 3.0f * yy);
        r[24] = C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
    }
}

/* ------------------------------------------------------------------------------------------
 * S1  CUDA spherical harmonics, degree <= 8         CuSHEncoder.cu:15-104 (restatement-pinned)
 *     Closed-form polynomials for UNIT directions.  Coefficient tables are laid out per band;
 *     the evaluation order of every expression follows the kernel text.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_sh_cu(const float *dirs, int64_t p, int degree, float *out)
{
    const int od = degree * degree;
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        const float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
        const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
        const float x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
        float *r = out + i * od;
        r[0] = 0.28209479177387814f;
        if (degree <= 1) continue;
        r[1] = -0.48860251190291987f * y;
        r[2] = 0.48860251190291987f * z;
        r[3] = -0.48860251190291987f * x;
        if (degree <= 2) continue;
        r[4] = 1.0925484305920792f * xy;
        r[5] = -1.0925484305920792f * yz;
        r[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        r[7] = -1.0925484305920792f * xz;
        r[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        if (degree <= 3) continue;
        r[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        r[10] = 2.8906114426405538f * xy * z;
        r[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        r[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        r[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        r[14] = 1.4453057213202769f * z * (x2 - y2);
        r[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
        if (degree <= 4) continue;
        r[16] = 2.5033429417967046f * xy * (x2 - y2);
        r[17] = 1.7701307697799304f * yz * (-3.0f * x2 + y2);
        r[18] = 0.94617469575756008f * xy * (7.0f * z2 - 1.0f);
        r[19] = 0.66904654355728921f * yz * (3.0f - 7.0f * z2);
        r[20] = -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f;
        r[21] = 0.66904654355728921f * xz * (3.0f - 7.0f * z2);
        r[22] = 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f);
        r[23] = 1.7701307697799304f * xz * (-x2 + 3.0f * y2);
        r[24] = -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4;
        if (degree <= 5) continue;
        r[25] = 0.65638205684017015f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4);
        r[26] = 8.3026492595241645f * xy * z * (x2 - y2);
        r[27] = -0.48923829943525038f * y * (3.0f * x2 - y2) * (9.0f * z2 - 1.0f);
        r[28] = 4.7935367849733241f * xy * z * (3.0f * z2 - 1.0f);
        r[29] = 0.45294665119569694f * y * (14.0f * z2 - 21.0f * z4 - 1.0f);
        r[30] = 0.1169503224534236f * z * (-70.0f * z2 + 63.0f * z4 + 15.0f);
        r[31] = 0.45294665119569694f * x * (14.0f * z2 - 21.0f * z4 - 1.0f);
        r[32] = 2.3967683924866621f * z * (x2 - y2) * (3.0f * z2 - 1.0f);
        r[33] = -0.48923829943525038f * x * (x2 - 3.0f * y2) * (9.0f * z2 - 1.0f);
        r[34] = 2.0756623148810411f * z * (-6.0f * x2 * y2 + x4 + y4);
        r[35] = 0.65638205684017015f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4);
        if (degree <= 6) continue;
        r[36] = 1.3663682103838286f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
        r[37] = 2.3666191622317521f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4);
        r[38] = 2.0182596029148963f * xy * (x2 - y2) * (11.0f * z2 - 1.0f);
        r[39] = -0.92120525951492349f * yz * (3.0f * x2 - y2) * (11.0f * z2 - 3.0f);
        r[40] = 0.92120525951492349f * xy * (-18.0f * z2 + 33.0f * z4 + 1.0f);
        r[41] = 0.58262136251873131f * yz * (30.0f * z2 - 33.0f * z4 - 5.0f);
        r[42] = 6.6747662381009842f * z2 - 20.024298714302954f * z4 + 14.684485723822165f * z6 - 0.31784601133814211f;
        r[43] = 0.58262136251873131f * xz * (30.0f * z2 - 33.0f * z4 - 5.0f);
        r[44] = 0.46060262975746175f * (x2 - y2) * (11.0f * z2 * (3.0f * z2 - 1.0f) - 7.0f * z2 + 1.0f);
        r[45] = -0.92120525951492349f * xz * (x2 - 3.0f * y2) * (11.0f * z2 - 3.0f);
        r[46] = 0.50456490072872406f * (11.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4);
        r[47] = 2.3666191622317521f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4);
        r[48] = 10.247761577878714f * x2 * y4 - 10.247761577878714f * x4 * y2 + 0.6831841051919143f * x6 - 0.6831841051919143f * y6;
        if (degree <= 7) continue;
        r[49] = 0.70716273252459627f * y * (-21.0f * x2 * y4 + 35.0f * x4 * y2 - 7.0f * x6 + y6);
        r[50] = 5.2919213236038001f * xy * z * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
        r[51] = -0.51891557872026028f * y * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + 5.0f * x4 + y4);
        r[52] = 4.1513246297620823f * xy * z * (x2 - y2) * (13.0f * z2 - 3.0f);
        r[53] = -0.15645893386229404f * y * (3.0f * x2 - y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f);
        r[54] = 0.44253269244498261f * xy * z * (-110.0f * z2 + 143.0f * z4 + 15.0f);
        r[55] = 0.090331607582517306f * y * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f);
        r[56] = 0.068284276912004949f * z * (315.0f * z2 - 693.0f * z4 + 429.0f * z6 - 35.0f);
        r[57] = 0.090331607582517306f * x * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f);
        r[58] = 0.07375544874083044f * z * (x2 - y2) * (143.0f * z2 * (3.0f * z2 - 1.0f) - 187.0f * z2 + 45.0f);
        r[59] = -0.15645893386229404f * x * (x2 - 3.0f * y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f);
        r[60] = 1.0378311574405206f * z * (13.0f * z2 - 3.0f) * (-6.0f * x2 * y2 + x4 + y4);
        r[61] = -0.51891557872026028f * x * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + x4 + 5.0f * y4);
        r[62] = 2.6459606618019f * z * (15.0f * x2 * y4 - 15.0f * x4 * y2 + x6 - y6);
        r[63] = 0.70716273252459627f * x * (-35.0f * x2 * y4 + 21.0f * x4 * y2 - x6 + 7.0f * y6);
    }
}

/* ------------------------------------------------------------------------------------------
 * H1  LibTorch hash-grid encoder           NeRF.cpp:208-318, NeRF.h:137-147
 *     resolution_l = floor(float(base * pow(b, l)))  with b = float(exp((ln finest - ln base)/(L-1)))
 *       (NeRF.cpp:251,309: the product is a double, torch::tensor(double) makes an fp32 tensor)
 *     grid = (max-min)/res ; idx = floor((clamp(x) - min)/grid) as int64 ; vmin = idx*grid + min ;
 *     vmax = vmin + grid ; hash = (ix*1 ^ iy*2654435761 ^ iz*805459861) & (2^T - 1) in int64 ;
 *     corner order z fastest ; trilinear weights use the UNCLAMPED x (NeRF.cpp:311 passes x) ;
 *     keep_mask = all(x == clamp(x)) (NeRF.cpp:213,316).
 *     table layout: [L][2^T][F] fp32 (L independent nn::Embedding, NeRF.cpp:255-256)
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_hash_ngp_resolutions(int n_levels, int base, int finest, float *res_out)
{
    float b = (float)exp((log((double)finest) - log((double)base)) / (double)(n_levels - 1));
    for (int l = 0; l < n_levels; l++) {
        double prod = (double)base * pow((double)b, (double)l);
        res_out[l] = floorf((float)prod);
    }
}

ORC_API void orc_hash_ngp(const float *x, int64_t p, const float *table, const float *bbox,
                          int n_levels, int n_feat, int log2_t, int base, int finest,
                          float *out /*[p, L*F]*/, uint8_t *mask /*[p]*/)
{
    float res[64];
    orc_hash_ngp_resolutions(n_levels, base, finest, res);
    const int64_t tsize = (int64_t)1 << log2_t;
    const int64_t hmask = tsize - 1;
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float xc[3];
        int keep = 1;
        for (int a = 0; a < 3; a++) {
            float v = x[i * 3 + a];
            float c = f_max(f_min(v, bbox[3 + a]), bbox[a]);
            if (!(v == c)) keep = 0;
            xc[a] = c;
        }
        if (mask) mask[i] = (uint8_t)keep;
        for (int l = 0; l < n_levels; l++) {
            float w[3];
            int64_t idx[3];
            for (int a = 0; a < 3; a++) {
                float grid = (bbox[3 + a] - bbox[a]) / res[l];
                idx[a] = (int64_t)floorf((xc[a] - bbox[a]) / grid);
                float vmin = (float)idx[a] * grid + bbox[a];
                float vmax = vmin + grid;           /* vmin + 1.0*grid */
                w[a] = (x[i * 3 + a] - vmin) / (vmax - vmin);
            }
            const float *tl = table + (int64_t)l * tsize * n_feat;
            const float *e[8];
            for (int c = 0; c < 8; c++) {
                int64_t cx = idx[0] + ((c >> 2) & 1), cy = idx[1] + ((c >> 1) & 1), cz = idx[2] + (c & 1);
                int64_t hsh = ((cx * 1LL) ^ (cy * 2654435761LL) ^ (cz * 805459861LL)) & hmask;
                e[c] = tl + hsh * n_feat;
            }
            const float omx = 1.0f - w[0], omy = 1.0f - w[1], omz = 1.0f - w[2];
            for (int f = 0; f < n_feat; f++) {
                float c00 = e[0][f] * omx + e[4][f] * w[0];
                float c01 = e[1][f] * omx + e[5][f] * w[0];
                float c10 = e[2][f] * omx + e[6][f] * w[0];
                float c11 = e[3][f] * omx + e[7][f] * w[0];
                float c0 = c00 * omy + c10 * w[1];
                float c1 = c01 * omy + c11 * w[1];
                out[i * n_levels * n_feat + l * n_feat + f] = c0 * omz + c1 * w[2];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * H2  CUDA hash-grid encoder               CuHashEmbedder.cpp:85-103 (host: mask + clamp),
 *                                          CuHashEmbedder.cu:27-101 (kernel), :221-275 (launch)
 *     RESTATEMENT-PINNED (CUDA-only unit).  fp16 table (fp32 master cast per call, .cu:257),
 *     uint32 hash with per-level random primes, position scale mul_l = exp2f(...) NOT floored,
 *     level base offset in ELEMENTS = feat_local_idx[l] while rows are F wide (the overlap
 *     quirk, .cu:54 vs :96), fp32 blend rounded once to fp16 (.cu:95), returned as fp32 (.cu:274).
 *     mul_l is computed on the host in fp32 libm (orc_hash_cu_scales) and handed to both this
 *     oracle and the HIP kernel, so the two agree bit for bit on voxel indices.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_hash_cu_scales(int n_levels, int base, int finest, float *mul_out)
{
    for (int l = 0; l < n_levels; l++)
        mul_out[l] = exp2f((log2f((float)finest) - log2f((float)base)) * (float)l / (float)(n_levels - 1) + log2f((float)base));
}

/* What a real CUDA build could do differently, as a switchable MODEL (test infrastructure for the sensitivity study in
 * tests/test_oracle_golden.py; default 0 = the expressions as written, one rounding per operation):
 *   bit 0: nvcc's default -fmad=true contracts the blend  w000*f000 + w001*f001 + ...  (.cu:95-100) into a chain of FMAs
 *          (first product rounded, every following product fused into its addition);
 *   bit 1: the same contraction of  pt * mul  followed by  pt += bias  (.cu:44-46, :61-63).
 * Which contractions ptxas really performs cannot be observed here (no nvcc); the study bounds their effect. */
static int g_cuda_fma_model = 0;
ORC_API void orc_set_cuda_fma_model(int flags) { g_cuda_fma_model = flags; }

ORC_API void orc_hash_cu(const float *x, int64_t p, const uint16_t *table_f16, const int32_t *primes /*[L,1,3]*/,
                         const int32_t *local_idx /*[L]*/, const int32_t *local_size /*[L]*/,
                         const float *bias /*[L,3]*/, const float *bbox, const float *mul /*[L]*/,
                         int n_levels, int n_feat,
                         float *out /*[p, L*F] fp32 holding fp16-rounded values*/, uint8_t *mask)
{
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float xc[3];
        int keep = 1;
        for (int a = 0; a < 3; a++) {
            float v = x[i * 3 + a];
            float c = f_max(f_min(v, bbox[3 + a]), bbox[a]);
            if (!(v == c)) keep = 0;
            xc[a] = c;
        }
        if (mask) mask[i] = (uint8_t)keep;
        for (int l = 0; l < n_levels; l++) {
            float pt[3], fl[3];
            uint32_t pos[3];
            for (int a = 0; a < 3; a++) {
                if (g_cuda_fma_model & 2) pt[a] = fmaf((xc[a] - bbox[a]) / (bbox[3 + a] - bbox[a]), mul[l], bias[l * 3 + a]);
                else {
                    pt[a] = (xc[a] - bbox[a]) / (bbox[3 + a] - bbox[a]) * mul[l];
                    pt[a] = pt[a] + bias[l * 3 + a];
                }
                fl[a] = floorf(pt[a]);
                pos[a] = (uint32_t)fl[a];
            }
            const uint32_t pa = (uint32_t)primes[l * 3 + 0], pb = (uint32_t)primes[l * 3 + 1], pc = (uint32_t)primes[l * 3 + 2];
            const uint32_t lsz = (uint32_t)local_size[l];
            const uint16_t *fp = table_f16 + local_idx[l];
            const float a = pt[0] - fl[0], b = pt[1] - fl[1], c = pt[2] - fl[2];
            float ws[8];
            uint32_t ps[8];
            for (int k = 0; k < 8; k++) {        /* k = (dx dy dz) bits, same order as pos_000..pos_111 */
                uint32_t dx = (k >> 2) & 1u, dy = (k >> 1) & 1u, dz = k & 1u;
                ps[k] = (((pos[0] + dx) * pa) ^ ((pos[1] + dy) * pb) ^ ((pos[2] + dz) * pc)) % lsz;
                float wx = dx ? a : (1.0f - a), wy = dy ? b : (1.0f - b), wz = dz ? c : (1.0f - c);
                ws[k] = wx * wy * wz;
            }
            for (int f = 0; f < n_feat; f++) {
                float acc = ws[0] * f16_bits_to_f32(fp[ps[0] * n_feat + f]);
                for (int k = 1; k < 8; k++) {
                    if (g_cuda_fma_model & 1) acc = fmaf(ws[k], f16_bits_to_f32(fp[ps[k] * n_feat + f]), acc);
                    else acc = acc + ws[k] * f16_bits_to_f32(fp[ps[k] * n_feat + f]);
                }
                out[i * n_levels * n_feat + l * n_feat + f] = f16_bits_to_f32(f32_to_f16_bits(acc));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * MLPs.  Parameters arrive as ONE fp32 blob in the reference's named_parameters() order
 * (== checkpoint order, NeRFExecutor.h:1055-1070), Linear weights [out, in] row-major.
 * Dot products: an fp32 FMA chain in ascending k starting from 0, bias added last -- the order the
 * HIP path's NRF_PREC_F32 mode reproduces bit for bit.  (LibTorch uses MKL sgemm whose blocking /
 * FMA order is unknowable; agreement is to ~1e-6 relative, tolerance stated in the tests.)
 * ------------------------------------------------------------------------------------------ */
static void linear(const float *w, const float *b, const float *x, int in, int out, float *y, int relu)
{
    for (int o = 0; o < out; o++) {
        float acc = 0.0f;
        const float *wr = w + (int64_t)o * in;
        for (int k = 0; k < in; k++) acc = fmaf(wr[k], x[k], acc);     /* one rounding per MAC, ascending k */
        if (b) acc += b[o];
        y[o] = (relu && acc < 0.0f) ? 0.0f : acc;
    }
}

/* M2  NeRFSmallImpl::forward               NeRF.cpp:322-412 (bias-free; use_pred_normal=false)
 *     sigma net: in -> H .. -> 1+geo (ReLU between); colour net: cat[views, geo] -> Hc .. -> 3;
 *     out = cat[colour, sigma] */
ORC_API int64_t orc_mlp_small_param_count(int in_ch, int in_views, int n_layers, int hidden, int geo, int n_layers_c, int hidden_c)
{
    int64_t n = 0;
    for (int l = 0; l < n_layers; l++) n += (int64_t)((l == 0) ? in_ch : hidden) * ((l == n_layers - 1) ? (1 + geo) : hidden);
    for (int l = 0; l < n_layers_c; l++) n += (int64_t)((l == 0) ? in_views + geo : hidden_c) * ((l == n_layers_c - 1) ? 3 : hidden_c);
    return n;
}

ORC_API void orc_mlp_small(const float *params, const float *x /*[p, in_ch+in_views]*/, int64_t p,
                           int in_ch, int in_views, int n_layers, int hidden, int geo, int n_layers_c, int hidden_c,
                           float *out /*[p,4]*/)
{
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float a[512], b2[512];
        const float *xi = x + i * (in_ch + in_views);
        const float *w = params;
        const float *cur = xi;
        float *bufs[2] = {a, b2};
        int cur_dim = in_ch;
        for (int l = 0; l < n_layers; l++) {
            int od = (l == n_layers - 1) ? (1 + geo) : hidden;
            linear(w, NULL, cur, cur_dim, od, bufs[l & 1], l != n_layers - 1);
            w += (int64_t)cur_dim * od;
            cur = bufs[l & 1]; cur_dim = od;
        }
        float sigma = cur[0];
        float cin[512];
        for (int k = 0; k < in_views; k++) cin[k] = xi[in_ch + k];
        for (int k = 0; k < geo; k++) cin[in_views + k] = cur[1 + k];
        cur = cin; cur_dim = in_views + geo;
        for (int l = 0; l < n_layers_c; l++) {
            int od = (l == n_layers_c - 1) ? 3 : hidden_c;
            linear(w, NULL, cur, cur_dim, od, bufs[l & 1], l != n_layers_c - 1);
            w += (int64_t)cur_dim * od;
            cur = bufs[l & 1]; cur_dim = od;
        }
        out[i * 4 + 0] = cur[0]; out[i * 4 + 1] = cur[1]; out[i * 4 + 2] = cur[2]; out[i * 4 + 3] = sigma;
    }
}

/* M2 with the predicted-normals head (NeRF.cpp:343-347, :393-407; built by the executor only when n_importance == 0 && use_pred_normal, NeRFExecutor.h:487):
 *     a third bias-free net on cat[sigma, geo_feat, input_pts] -> Hn .. -> 3, no final activation; out = cat[colour, sigma, normals] [p, 7].
 *     blob: sigma net, colour net, normals net (registration order, NeRF.cpp:349-359) */
ORC_API void orc_mlp_small_pred_normal(const float *params, const float *x, int64_t p, int in_ch, int in_views, int n_layers, int hidden, int geo, int n_layers_c,
                                       int hidden_c, int n_layers_n, int hidden_n, float *out /*[p,7]*/)
{
    int64_t off_n = 0;
    { int cd = in_ch; for (int l = 0; l < n_layers; l++) { int od = (l == n_layers - 1) ? (1 + geo) : hidden; off_n += (int64_t)cd * od; cd = od; }
      cd = in_views + geo; for (int l = 0; l < n_layers_c; l++) { int od = (l == n_layers_c - 1) ? 3 : hidden_c; off_n += (int64_t)cd * od; cd = od; } }
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float a[1024], b2[1024], h33[512], cin[1024];
        const float *xi = x + i * (in_ch + in_views);
        float o4[4];
        orc_mlp_small(params, xi, 1, in_ch, in_views, n_layers, hidden, geo, n_layers_c, hidden_c, o4);
        /* the sigma net's output again (orc_mlp_small keeps it local): same chain, same bits */
        { const float *w = params; const float *cur = xi; int cd = in_ch; float *bufs[2] = {a, b2};
          for (int l = 0; l < n_layers; l++) { int od = (l == n_layers - 1) ? (1 + geo) : hidden; linear(w, NULL, cur, cd, od, bufs[l & 1], l != n_layers - 1); w += (int64_t)cd * od; cur = bufs[l & 1]; cd = od; }
          memcpy(h33, cur, sizeof(float) * (1 + geo)); }
        for (int k = 0; k < 1 + geo; k++) cin[k] = h33[k];                 /* cat[sigma.unsqueeze(-1), geo_feat, input_pts]  (NeRF.cpp:396) */
        for (int k = 0; k < in_ch; k++) cin[1 + geo + k] = xi[k];
        const float *w = params + off_n; const float *cur = cin; int cd = 1 + geo + in_ch; float *bufs[2] = {a, b2};
        for (int l = 0; l < n_layers_n; l++) { int od = (l == n_layers_n - 1) ? 3 : hidden_n; linear(w, NULL, cur, cd, od, bufs[l & 1], l != n_layers_n - 1); w += (int64_t)cd * od; cur = bufs[l & 1]; cd = od; }
        for (int k = 0; k < 4; k++) out[i * 7 + k] = o4[k];
        for (int k = 0; k < 3; k++) out[i * 7 + 4 + k] = cur[k];
    }
}

/* M1  NeRFImpl::forward                    NeRF.cpp:41-126
 *     D Linear(+bias)+ReLU; after layer index `skip` h = cat[input_pts, h] (NeRF.cpp:103-104);
 *     viewdirs: alpha = Linear(W,1)(h); feat = Linear(W,W)(h) (no ReLU); h = cat[feat, views];
 *     ReLU(Linear(W+views, W/2)); rgb = Linear(W/2,3); out = cat[rgb, alpha]
 *     no viewdirs: out = Linear(W+in, out_ch)(cat[h, input_pts]) (NeRF.cpp:121-124)
 *     blob order: pts_linears_i.{weight,bias}..., then (viewdirs) views_linears_0.{w,b},
 *     feature_linear.{w,b}, alpha_linear.{w,b}, rgb_linear.{w,b}  |  (else) output_linear.{w,b} */
ORC_API int64_t orc_mlp_nerf_param_count(int d, int w, int in_ch, int in_views, int out_ch, int skip, int use_viewdirs)
{
    int64_t n = (int64_t)in_ch * w + w;
    for (int i = 0; i < d - 1; i++) n += (int64_t)((i == skip) ? (w + in_ch) : w) * w + w;
    if (use_viewdirs) n += (int64_t)(in_views + w) * (w / 2) + w / 2 + (int64_t)w * w + w + w + 1 + (int64_t)(w / 2) * 3 + 3;
    else n += (int64_t)(w + in_ch) * out_ch + out_ch;
    return n;
}

ORC_API void orc_mlp_nerf(const float *params, const float *x, int64_t p, int d, int w, int in_ch, int in_views,
                          int out_ch, int skip, int use_viewdirs, float *out)
{
    const int xd = in_ch + (use_viewdirs ? in_views : 0);
    const int od = use_viewdirs ? 4 : out_ch;
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float h0[1024], h1[1024];
        const float *xi = x + i * xd;
        const float *wp = params;
        float *bufs[2] = {h0, h1};
        const float *cur = xi;
        int cur_dim = in_ch;
        int which = 0;
        for (int l = 0; l < d; l++) {
            float *dst = bufs[which];
            int off = (l == skip) ? in_ch : 0;       /* after layer `skip`: cat[input_pts, h] */
            linear(wp, wp + (int64_t)cur_dim * w, cur, cur_dim, w, dst + off, 1);
            wp += (int64_t)cur_dim * w + w;
            if (off) memcpy(dst, xi, sizeof(float) * in_ch);
            cur = dst; cur_dim = w + off; which ^= 1;
        }
        if (use_viewdirs) {
            const float *vw = wp; wp += (int64_t)(in_views + w) * (w / 2) + w / 2;
            const float *fw = wp; wp += (int64_t)w * w + w;
            const float *aw = wp; wp += w + 1;
            const float *rw = wp;
            float alpha;
            linear(aw, aw + w, cur, w, 1, &alpha, 0);
            float *feat = bufs[which];
            linear(fw, fw + (int64_t)w * w, cur, w, w, feat, 0);
            memcpy(feat + w, xi + in_ch, sizeof(float) * in_views);
            float *hv = bufs[which ^ 1];
            linear(vw, vw + (int64_t)(in_views + w) * (w / 2), feat, w + in_views, w / 2, hv, 1);
            linear(rw, rw + (int64_t)(w / 2) * 3, hv, w / 2, 3, out + i * od, 0);
            out[i * od + 3] = alpha;
        } else {
            float *hc = bufs[which];
            memcpy(hc, cur, sizeof(float) * w);
            memcpy(hc + w, xi, sizeof(float) * in_ch);
            linear(wp, wp + (int64_t)(w + in_ch) * out_ch, hc, w + in_ch, out_ch, out + i * od, 0);
        }
    }
}

/* L1  LeRFImpl::forward                    LeRF.cpp:28-111 (bias-free)
 *     sigma net: in -> H.. -> 1+geo ; LE net: cat[geo, in] -> H.. -> E ; L2-normalise (eps 1e-8:
 *     x / max(||x||, eps)) ; out = cat[le, sigma] */
ORC_API void orc_lerf(const float *params, const float *x, int64_t p, int in_ch, int n_layers, int hidden, int geo, int embed,
                      float *out /*[p, embed+1]*/)
{
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float *a = (float *)malloc(sizeof(float) * 4096), *b2 = (float *)malloc(sizeof(float) * 4096), *cin = (float *)malloc(sizeof(float) * 4096);
        float *bufs[2] = {a, b2};
        const float *xi = x + i * in_ch;
        const float *w = params;
        const float *cur = xi;
        int cur_dim = in_ch;
        for (int l = 0; l < n_layers; l++) {
            int od = (l == n_layers - 1) ? (1 + geo) : hidden;
            linear(w, NULL, cur, cur_dim, od, bufs[l & 1], l != n_layers - 1);
            w += (int64_t)cur_dim * od; cur = bufs[l & 1]; cur_dim = od;
        }
        float sigma = cur[0];
        for (int k = 0; k < geo; k++) cin[k] = cur[1 + k];
        for (int k = 0; k < in_ch; k++) cin[geo + k] = xi[k];
        cur = cin; cur_dim = geo + in_ch;
        for (int l = 0; l < n_layers; l++) {
            int od = (l == n_layers - 1) ? embed : hidden;
            linear(w, NULL, cur, cur_dim, od, bufs[l & 1], l != n_layers - 1);
            w += (int64_t)cur_dim * od; cur = bufs[l & 1]; cur_dim = od;
        }
        double ss = 0.0;
        for (int k = 0; k < embed; k++) ss += (double)cur[k] * (double)cur[k];
        float nrm = f_max((float)sqrt(ss), 1e-8f);
        for (int k = 0; k < embed; k++) out[i * (embed + 1) + k] = cur[k] / nrm;
        out[i * (embed + 1) + embed] = sigma;
        free(a); free(b2); free(cin);
    }
}

/* the density net of LeRFImpl::forward alone (LeRF.cpp:86-95): h = SigmaLENet(x) -> out [p, 1+geo] (row 0 = sigma_le, rows 1.. = geo_feat_le) */
ORC_API void orc_lerf_sigma_net(const float *params, const float *x, int64_t p, int in_ch, int n_layers, int hidden, int geo, float *out /*[p, 1+geo]*/)
{
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        float *a = (float *)malloc(sizeof(float) * 4096), *b2 = (float *)malloc(sizeof(float) * 4096);
        float *bufs[2] = {a, b2};
        const float *w = params;
        const float *cur = x + i * in_ch;
        int cur_dim = in_ch;
        for (int l = 0; l < n_layers; l++) {
            int od = (l == n_layers - 1) ? (1 + geo) : hidden;
            linear(w, NULL, cur, cur_dim, od, bufs[l & 1], l != n_layers - 1);
            w += (int64_t)cur_dim * od; cur = bufs[l & 1]; cur_dim = od;
        }
        memcpy(out + i * (1 + geo), cur, sizeof(float) * (1 + geo));
        free(a); free(b2);
    }
}

/* ------------------------------------------------------------------------------------------
 * C1  RawToOutputs                         NeRFRenderer.h:199-282, TruncExp fwd CustomOps.cpp:5-9
 *     dists = [z[i+1]-z[i], 1e10] * ||d|| ; rgb = sigmoid(raw[:3]) ;
 *     alpha = -exp(-relu(sigma)*dists) + 1 ;
 *     T_i = exp(sum_{j<i} log(max(1-alpha_j, 1e-10)))  -- cumsum accumulates in DOUBLE, each prefix
 *     rounded to fp32 (ATen CPU cumsum) ; w = alpha*T ; rgb_map = sum w*rgb ;
 *     depth = sum(w*z)/max(sum w, 1e-10) ; disp = 1/max(1e-10, depth) ; acc = sum w ;
 *     white: rgb_map + (1 - acc).   Reductions over samples: double accumulate, one rounding.
 *     raw has `c` channels per sample with rgb at 0..2 and sigma at 3 (c = 4 or 5).
 * ------------------------------------------------------------------------------------------ */
static void raw2outputs_impl(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int sigma_ch, int white_bkgr,
                             float *rgb_map, float *disp, float *acc_map, float *weights, float *depth, const float *noise, float noise_std);

ORC_API void orc_raw2outputs(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int white_bkgr,
                             float *rgb_map, float *disp, float *acc_map, float *weights, float *depth)
{
    raw2outputs_impl(raw, z, d, n, s, c, 3, white_bkgr, rgb_map, disp, acc_map, weights, depth, NULL, 0.0f);
}

/* RawNoiseStd > 0 (training): sigma + noise*std before the relu (NeRFRenderer.h:251-252); noise: the [n,s] normal draws */
ORC_API void orc_raw2outputs_noise(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int white_bkgr, const float *noise,
                                   float noise_std, float *rgb_map, float *disp, float *acc_map, float *weights, float *depth)
{
    raw2outputs_impl(raw, z, d, n, s, c, 3, white_bkgr, rgb_map, disp, acc_map, weights, depth, noise, noise_std);
}

/* L2  LeRFRenderer::RawToLEOutputs, weights part   LeRFRenderer.cpp:27-76 (sigma_le at channel lang_embed_dim, no colour) */
ORC_API void orc_raw2weights(const float *raw, int c, int sigma_ch, const float *z, const float *d, int64_t n, int s,
                             float *weights, float *depth, float *disp, float *acc_map)
{
    raw2outputs_impl(raw, z, d, n, s, c, sigma_ch, 0, NULL, disp, acc_map, weights, depth, NULL, 0.0f);
}

/* L2  RenderCLIPEmbedding                          LeRFRenderer.h:45-54: normalize(sum_s w*e, eps 1e-8) */
ORC_API void orc_render_clip_embedding(const float *embeds, int stride, int dim, const float *w, int64_t n, int s, float *out)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        double ss = 0.0;
        for (int k = 0; k < dim; k++) {
            double acc = 0.0;
            for (int j = 0; j < s; j++) acc += (double)(w[i * s + j] * embeds[(i * s + j) * (int64_t)stride + k]);
            out[i * dim + k] = (float)acc;
            ss += (double)out[i * dim + k] * (double)out[i * dim + k];
        }
        float nrm = f_max((float)sqrt(ss), 1e-8f);
        for (int k = 0; k < dim; k++) out[i * dim + k] = out[i * dim + k] / nrm;
    }
}

static void raw2outputs_impl(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int sigma_ch, int white_bkgr,
                             float *rgb_map, float *disp, float *acc_map, float *weights, float *depth, const float *noise, float noise_std)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        const float *dv = d + i * 3;
        float nrm = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        double logt = 0.0;        /* running double prefix */
        float tprev = 0.0f;       /* exclusive prefix rounded to fp32 */
        double sr = 0, sg = 0, sb = 0, sw = 0, swz = 0;
        for (int j = 0; j < s; j++) {
            const float *r = raw + (i * s + j) * c;
            float dist = (j + 1 < s) ? (z[i * s + j + 1] - z[i * s + j]) : 1e10f;
            dist = dist * nrm;
            float sraw = r[sigma_ch];
            if (noise) sraw = sraw + noise[i * s + j] * noise_std;
            float sig = sraw > 0.0f ? sraw : 0.0f;
            float alpha = -nrf_expf(-sig * dist) + 1.0f;
            float trans = nrf_expf(tprev);
            float w = alpha * trans;
            float one_m = 1.0f - alpha;
            float lg = nrf_logf(one_m > 1e-10f ? one_m : 1e-10f);
            logt += (double)lg;
            tprev = (float)logt;
            if (weights) weights[i * s + j] = w;
            float cr = 0.0f, cg = 0.0f, cb = 0.0f;
            if (rgb_map) { cr = nrf_sigmoidf(r[0]); cg = nrf_sigmoidf(r[1]); cb = nrf_sigmoidf(r[2]); }
            sr += (double)(w * cr); sg += (double)(w * cg); sb += (double)(w * cb);
            sw += (double)w; swz += (double)(w * z[i * s + j]);
        }
        float acc = (float)sw;
        float dep = (float)swz / (acc > 1e-10f ? acc : 1e-10f);
        float rr = (float)sr, gg = (float)sg, bb = (float)sb;
        if (white_bkgr) { float bg = 1.0f - acc; rr = rr + bg; gg = gg + bg; bb = bb + bg; }
        if (rgb_map) { rgb_map[i * 3] = rr; rgb_map[i * 3 + 1] = gg; rgb_map[i * 3 + 2] = bb; }
        if (depth) depth[i] = dep;
        if (disp) disp[i] = 1.0f / (dep > 1e-10f ? dep : 1e-10f);
        if (acc_map) acc_map[i] = acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * R8  SamplePDF (det)                      Sampler.h:6-43
 *     w += 1e-8 ; pdf = w / sum(w) (sum in ATen's order for a `sum_vec`-wide host, see
 *     aten_row_sum_f32; 0 = double accumulate) ; cdf = [0, cumsum(pdf)] (double accumulate, fp32 prefixes) ;
 *     inds = searchsorted(cdf, u, right=True) = #{cdf_k <= u} ; below = max(0, inds-1) ;
 *     above = min(nb-1, inds) ; denom = cdf[above]-cdf[below], <1e-5 -> 1 ;
 *     t = (u - cdf[below]) / denom ; sample = bins[below] + t*(bins[above]-bins[below])
 *     bins: [n, nb], weights: [n, nb-1], u: [ns] (torch::linspace(0,1,ns) supplied by the host).
 * ------------------------------------------------------------------------------------------ */
static void sample_pdf_impl(const float *bins, const float *weights, int64_t n, int nb, const float *u_all, int64_t u_stride, int ns, int sum_vec,
                            float *samples, int64_t *inds_out, float *cdf_out);

ORC_API void orc_sample_pdf(const float *bins, const float *weights, int64_t n, int nb, const float *u, int ns, int sum_vec,
                            float *samples, int64_t *inds_out, float *cdf_out)
{
    sample_pdf_impl(bins, weights, n, nb, u, 0, ns, sum_vec, samples, inds_out, cdf_out);
}

/* det = false (Sampler.h:22-24): u = torch::rand([n, ns]), one row per ray, NOT sorted; everything else is unchanged */
ORC_API void orc_sample_pdf_rand(const float *bins, const float *weights, int64_t n, int nb, const float *u /* [n,ns] */, int ns, int sum_vec,
                                 float *samples, int64_t *inds_out)
{
    sample_pdf_impl(bins, weights, n, nb, u, ns, ns, sum_vec, samples, inds_out, NULL);
}

static void sample_pdf_impl(const float *bins, const float *weights, int64_t n, int nb, const float *u_all, int64_t u_stride, int ns, int sum_vec,
                            float *samples, int64_t *inds_out, float *cdf_out)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        float cdf[1024], wv[1024];
        const int nw = nb - 1;
        for (int k = 0; k < nw; k++) wv[k] = weights[i * nw + k] + (float)1e-8;
        float fsum = aten_row_sum_f32(wv, nw, sum_vec);
        cdf[0] = 0.0f;
        double run = 0.0;
        for (int k = 0; k < nw; k++) { float pdf = wv[k] / fsum; run += (double)pdf; cdf[k + 1] = (float)run; }
        if (cdf_out) memcpy(cdf_out + i * nb, cdf, sizeof(float) * nb);
        const float *b = bins + i * nb;
        const float *u = u_all + i * u_stride;
        for (int j = 0; j < ns; j++) {
            int lo = 0, hi = nb;                 /* first index with cdf[idx] > u */
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] <= u[j]) lo = mid + 1; else hi = mid; }
            int ind = lo;
            int below = ind - 1 > 0 ? ind - 1 : 0;
            int above = ind < nb - 1 ? ind : nb - 1;
            float denom = cdf[above] - cdf[below];
            if (denom < 1e-5f) denom = 1.0f;
            float t = (u[j] - cdf[below]) / denom;
            samples[i * ns + j] = b[below] + t * (b[above] - b[below]);
            if (inds_out) inds_out[i * ns + j] = ind;
        }
    }
}

/* z_mid = .5*(z[1:] + z[:-1])              NeRFRenderer.h:427 */
ORC_API void orc_z_mid(const float *z, int64_t n, int s, float *mid)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j + 1 < s; j++) mid[i * (s - 1) + j] = 0.5f * (z[i * s + j + 1] + z[i * s + j]);
}

/* sort(cat(z_vals, z_samples))             NeRFRenderer.h:431  (stable ascending; values only are used) */
static int cmp_float(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
ORC_API void orc_merge_sorted(const float *z, int s, const float *zs, int ns, int64_t n, float *out)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        float *o = out + i * (s + ns);
        memcpy(o, z + i * s, sizeof(float) * s);
        memcpy(o + s, zs + i * ns, sizeof(float) * ns);
        qsort(o, s + ns, sizeof(float), cmp_float);
    }
}

/* ------------------------------------------------------------------------------------------
 * R4/R5/R6/R9  RenderRays end to end       NeRFRenderer.h:366-459 (+RunNetwork :164-194)
 *     ThinRay (no TangentScatter), Perturb = 0, RawNoiseStd = 0.  One network for both passes.
 *     family 0 = HashEmbedder(H1) + SHEncoder(S2) + NeRFSmall ; family 1 = PE + PE + NeRF ;
 *     family 2 = CuHashEmbedder(H2) + CuSHEncoder(S1) + NeRFSmall.
 *     rays: [n, 11] = o, d, near, far, viewdir  (NeRFRenderer.h:580-583).
 *     Used by the tests as the end-to-end checker and by bench.py's cpu_baseline ("port").
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int family;
    /* encoders */
    const float *bbox;
    int n_levels, n_feat, log2_t, base, finest;
    const float *table_f32;            /* family 0 */
    const uint16_t *table_f16;         /* family 2 */
    const int32_t *primes, *local_idx, *local_size;
    const float *bias, *mul;
    int sh_degree;
    int pe_freqs, pe_freqs_views;
    /* MLP */
    const float *params;
    int n_layers, hidden, geo, n_layers_c, hidden_c;   /* small */
    int depth, width, skip;                             /* classic */
} orc_model;

static void run_network(const orc_model *m, const float *pts, const float *viewdirs, int64_t n, int s, float *raw)
{
    const int64_t p = n * s;
    int in_ch, in_views;
    float *emb, *embd;
    uint8_t *mask = NULL;
    if (m->family == 1) { in_ch = 3 + 6 * m->pe_freqs; in_views = 3 + 6 * m->pe_freqs_views; }
    else { in_ch = m->n_levels * m->n_feat; in_views = m->sh_degree * m->sh_degree; }
    emb = (float *)malloc(sizeof(float) * p * in_ch);
    embd = (float *)malloc(sizeof(float) * n * in_views);
    if (m->family == 0) { mask = (uint8_t *)malloc(p); orc_hash_ngp(pts, p, m->table_f32, m->bbox, m->n_levels, m->n_feat, m->log2_t, m->base, m->finest, emb, mask); orc_sh_libtorch(viewdirs, n, m->sh_degree, embd); }
    else if (m->family == 2) { mask = (uint8_t *)malloc(p); orc_hash_cu(pts, p, m->table_f16, m->primes, m->local_idx, m->local_size, m->bias, m->bbox, m->mul, m->n_levels, m->n_feat, emb, mask); orc_sh_cu(viewdirs, n, m->sh_degree, embd); }
    else { orc_pe(pts, p, m->pe_freqs, emb); orc_pe(viewdirs, n, m->pe_freqs_views, embd); }
    /* view-direction features are identical for every sample of a ray (NeRFRenderer.h:179 expands them) */
    float *xin = (float *)malloc(sizeof(float) * p * (in_ch + in_views));
    OMP_FOR
    for (int64_t i = 0; i < p; i++) {
        memcpy(xin + i * (in_ch + in_views), emb + i * in_ch, sizeof(float) * in_ch);
        memcpy(xin + i * (in_ch + in_views) + in_ch, embd + (i / s) * in_views, sizeof(float) * in_views);
    }
    if (m->family == 1) orc_mlp_nerf(m->params, xin, p, m->depth, m->width, in_ch, in_views, 4, m->skip, 1, raw);
    else orc_mlp_small(m->params, xin, p, in_ch, in_views, m->n_layers, m->hidden, m->geo, m->n_layers_c, m->hidden_c, raw);
    if (mask) {
        for (int64_t i = 0; i < p; i++) if (!mask[i]) raw[i * 4 + 3] = 0.0f;      /* NeRFRenderer.h:187-188 */
        free(mask);
    }
    free(emb); free(embd); free(xin);
}

/* The stochastic branches of RenderRays.  Every draw array may be given explicitly (the reference's own torch::rand / randn draws,
 * replayed: that is how tests pin this code against the reference) or left NULL, in which case it is generated from
 * (seed, stream, global element index) with include/nrf_rng.h -- the definition the HIP renderer uses. */
typedef struct {
    float perturb;                 /* > 0: stratified jitter + SamplePDF(det = false) */
    int has_cone; float cone_angle;/* cone rays (ThinRay = false): TangentScatter on both passes */
    float raw_noise_std, precond_alpha;
    uint64_t seed; int64_t ray_base;   /* index of rays[0] in the whole image (Chunk / shard independence) */
    const float *t_rand, *u_r1, *u_theta1, *noise1, *u_pdf, *precond, *u_r2, *u_theta2, *noise2;
} orc_stoch;

static float *draws(const float *given, int normal, uint64_t seed, uint32_t stream, int64_t idx0, int64_t count)
{
    float *a = (float *)malloc(sizeof(float) * (count > 0 ? count : 1));
    if (given) memcpy(a, given, sizeof(float) * count);
    else if (normal) orc_rng_normal(seed, stream, (uint64_t)idx0, count, a);
    else orc_rng_uniform(seed, stream, (uint64_t)idx0, count, a);
    return a;
}

ORC_API void orc_render_rays_stoch(const orc_model *m, const float *rays, int64_t n, int n_samples, int n_importance,
                             const float *t_coarse, const float *u_fine, int lindisp, int white_bkgr, int sum_vec, const orc_stoch *st,
                             float *rgb, float *disp, float *acc, float *depth, float *weights_fine,
                             float *z_coarse_out, float *z_fine_out, float *raw_coarse_out, float *raw_fine_out,
                             float *weights_coarse_out, float *pts_coarse_out, float *pts_fine_out);

ORC_API void orc_render_rays(const orc_model *m, const float *rays, int64_t n, int n_samples, int n_importance,
                             const float *t_coarse, const float *u_fine, int lindisp, int white_bkgr, int sum_vec,
                             float *rgb, float *disp, float *acc, float *depth, float *weights_fine,
                             float *z_coarse_out, float *z_fine_out, float *raw_coarse_out, float *raw_fine_out,
                             float *weights_coarse_out)
{
    orc_render_rays_stoch(m, rays, n, n_samples, n_importance, t_coarse, u_fine, lindisp, white_bkgr, sum_vec, NULL, rgb, disp, acc, depth,
                          weights_fine, z_coarse_out, z_fine_out, raw_coarse_out, raw_fine_out, weights_coarse_out, NULL, NULL);
}

ORC_API void orc_render_rays_stoch(const orc_model *m, const float *rays, int64_t n, int n_samples, int n_importance,
                             const float *t_coarse, const float *u_fine, int lindisp, int white_bkgr, int sum_vec, const orc_stoch *st,
                             float *rgb, float *disp, float *acc, float *depth, float *weights_fine,
                             float *z_coarse_out, float *z_fine_out, float *raw_coarse_out, float *raw_fine_out,
                             float *weights_coarse_out, float *pts_coarse_out, float *pts_fine_out)
{
    const int jitter = st && st->perturb > 0.0f, cone = st && st->has_cone;
    const float nstd = st ? st->raw_noise_std : 0.0f, palpha = st ? st->precond_alpha : 0.0f;
    const int64_t rb = st ? st->ray_base : 0;
    const uint64_t seed = st ? st->seed : 0;
    const int s = n_samples, sf = n_samples + n_importance;
    float *o = (float *)malloc(sizeof(float) * n * 3), *d = (float *)malloc(sizeof(float) * n * 3), *vd = (float *)malloc(sizeof(float) * n * 3);
    float *nears = (float *)malloc(sizeof(float) * n), *fars = (float *)malloc(sizeof(float) * n);
    for (int64_t i = 0; i < n; i++) {
        memcpy(o + i * 3, rays + i * 11, 12); memcpy(d + i * 3, rays + i * 11 + 3, 12); memcpy(vd + i * 3, rays + i * 11 + 8, 12);
        nears[i] = rays[i * 11 + 6]; fars[i] = rays[i * 11 + 7];
    }
    float *z = (float *)malloc(sizeof(float) * n * s), *pts = (float *)malloc(sizeof(float) * n * sf * 3);
    float *raw = (float *)malloc(sizeof(float) * n * sf * 4), *wc = (float *)malloc(sizeof(float) * n * s);
    float *c_rgb = (float *)malloc(sizeof(float) * n * 3);
    orc_z_vals(nears, fars, t_coarse, n, s, lindisp, z);
    if (jitter) {                                                                       /* :404-417 */
        float *tr = draws(st->t_rand, 0, seed, NRF_RNG_T_RAND, rb * s, n * s), *zj = (float *)malloc(sizeof(float) * n * s);
        orc_jitter_z(z, tr, n, s, zj);
        memcpy(z, zj, sizeof(float) * n * s);
        free(tr); free(zj);
    }
    orc_points(o, d, z, n, s, pts);
    if (cone) {                                                                         /* :420 */
        float *ur = draws(st->u_r1, 0, seed, NRF_RNG_R_COARSE, rb * s, n * s), *ut = draws(st->u_theta1, 0, seed, NRF_RNG_THETA_COARSE, rb * s, n * s);
        orc_tangent_scatter(pts, z, st->cone_angle, d, ur, ut, m->bbox, n, s, pts);
        free(ur); free(ut);
    }
    if (pts_coarse_out) memcpy(pts_coarse_out, pts, sizeof(float) * n * s * 3);
    run_network(m, pts, vd, n, s, raw);
    {
        float *nz = nstd > 0.0f ? draws(st->noise1, 1, seed, NRF_RNG_NOISE_COARSE, rb * s, n * s) : NULL;
        if (n_importance <= 0) raw2outputs_impl(raw, z, d, n, s, 4, 3, white_bkgr, rgb, disp, acc, weights_fine, depth, nz, nstd);
        else raw2outputs_impl(raw, z, d, n, s, 4, 3, white_bkgr, c_rgb, NULL, NULL, wc, NULL, nz, nstd);
        free(nz);
    }
    if (z_coarse_out) memcpy(z_coarse_out, z, sizeof(float) * n * s);
    if (raw_coarse_out) memcpy(raw_coarse_out, raw, sizeof(float) * n * s * 4);
    if (weights_coarse_out && n_importance > 0) memcpy(weights_coarse_out, wc, sizeof(float) * n * s);
    if (n_importance > 0) {
        float *mid = (float *)malloc(sizeof(float) * n * (s - 1)), *wmid = (float *)malloc(sizeof(float) * n * (s - 2));
        float *zs = (float *)malloc(sizeof(float) * n * n_importance), *zf = (float *)malloc(sizeof(float) * n * sf);
        orc_z_mid(z, n, s, mid);
        for (int64_t i = 0; i < n; i++) memcpy(wmid + i * (s - 2), wc + i * s + 1, sizeof(float) * (s - 2));   /* weights[..., 1:-1] */
        if (jitter) {                                                                   /* SamplePDF(det = (perturb == 0)) :428 */
            float *up = draws(st->u_pdf, 0, seed, NRF_RNG_U_PDF, rb * n_importance, n * n_importance);
            orc_sample_pdf_rand(mid, wmid, n, s - 1, up, n_importance, sum_vec, zs, NULL);
            free(up);
        } else orc_sample_pdf(mid, wmid, n, s - 1, u_fine, n_importance, sum_vec, zs, NULL, NULL);
        orc_merge_sorted(z, s, zs, n_importance, n, zf);
        orc_points(o, d, zf, n, sf, pts);
        if (palpha > 0.0f) {                                                            /* :433-443 */
            float *pn = draws(st->precond, 1, seed, NRF_RNG_PRECOND, rb * sf * 3, n * sf * 3);
            orc_precondition(pts, pn, palpha, m->bbox, n * sf, pts);
            free(pn);
        }
        if (cone) {                                                                     /* :445 */
            float *ur = draws(st->u_r2, 0, seed, NRF_RNG_R_FINE, rb * sf, n * sf), *ut = draws(st->u_theta2, 0, seed, NRF_RNG_THETA_FINE, rb * sf, n * sf);
            orc_tangent_scatter(pts, zf, st->cone_angle, d, ur, ut, m->bbox, n, sf, pts);
            free(ur); free(ut);
        }
        if (pts_fine_out) memcpy(pts_fine_out, pts, sizeof(float) * n * sf * 3);
        run_network(m, pts, vd, n, sf, raw);
        {
            float *nz = nstd > 0.0f ? draws(st->noise2, 1, seed, NRF_RNG_NOISE_FINE, rb * sf, n * sf) : NULL;
            raw2outputs_impl(raw, zf, d, n, sf, 4, 3, white_bkgr, rgb, disp, acc, weights_fine, depth, nz, nstd);
            free(nz);
        }
        if (z_fine_out) memcpy(z_fine_out, zf, sizeof(float) * n * sf);
        if (raw_fine_out) memcpy(raw_fine_out, raw, sizeof(float) * n * sf * 4);
        free(mid); free(wmid); free(zs); free(zf);
    }
    free(o); free(d); free(vd); free(nears); free(fars); free(z); free(pts); free(raw); free(wc); free(c_rgb);
}

/* viewdirs = d / ||d||                     NeRFRenderer.h:559 ; rays_ = cat[o, d, near, far, viewdirs] :580-583 */
ORC_API void orc_pack_rays(const float *o, const float *d, const float *bbox, int64_t n, float *rays /*[n,11]*/)
{
    float *nears = (float *)malloc(sizeof(float) * n), *fars = (float *)malloc(sizeof(float) * n);
    orc_aabb(o, d, bbox, n, 0.0f, nears, fars);
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        const float *dv = d + i * 3;
        float nrm = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        float *r = rays + i * 11;
        for (int a = 0; a < 3; a++) { r[a] = o[i * 3 + a]; r[3 + a] = dv[a]; r[8 + a] = dv[a] / nrm; }
        r[6] = nears[i]; r[7] = fars[i];
    }
    free(nears); free(fars);
}

/* ==========================================================================================
 * N1  training step: backward of the render path + loss + Adam.   NeRFExecutor.h:862-995, :539
 *     Gradients flow only through the FINE pass: z_samples are detached (NeRFRenderer.h:429), rays and depths carry no
 *     parameters, so d loss / d raw of the coarse pass is identically zero (golden train_hash.s1_coarse_raw_has_grad).
 *     Autograd's own summation orders are not reproduced (the GPU accumulates with atomics anyway): tests compare with a
 *     relative tolerance.  Accumulations here are in double.
 * ========================================================================================== */

/* torch::nn::functional::huber_loss (delta = 1, mean) and mse_loss; grad = d huber / d pred     NeRFExecutor.h:882-887 */
ORC_API void orc_huber_loss(const float *pred, const float *target, int64_t count, float *loss, float *mse, float *grad)
{
    double acc = 0.0, acc2 = 0.0;
    const float norm = 1.0f / (float)count;
    for (int64_t i = 0; i < count; i++) {
        const float d = pred[i] - target[i];
        const float z = fabsf(d);
        acc += (z < 1.0f) ? 0.5 * (double)z * (double)z : (double)z - 0.5;
        acc2 += (double)d * (double)d;
        if (grad) grad[i] = (d < -1.0f) ? -norm : (d > 1.0f ? norm : norm * d);
    }
    if (loss) *loss = (float)(acc / (double)count);
    if (mse) *mse = (float)(acc2 / (double)count);
}

/* Backward of RawToOutputs (NeRFRenderer.h:199-282) w.r.t. raw, given d loss / d rgb_map [n,3] (the only output the
 * training loss reads).  TruncExp::backward = grad * exp(clamp(x, -100, 5)) (CustomOps.cpp:11-15); clamp_min passes the
 * gradient where 1 - alpha >= 1e-10; relu where sigma > 0. */
static void raw2outputs_backward_impl(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int white_bkgr,
                                       const float *noise, float noise_std, const float *g_rgb, float *g_raw);

ORC_API void orc_raw2outputs_backward(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int white_bkgr,
                                      const float *g_rgb /*[n,3]*/, float *g_raw /*[n,s,c]*/)
{
    raw2outputs_backward_impl(raw, z, d, n, s, c, white_bkgr, NULL, 0.0f, g_rgb, g_raw);
}

/* ... of a forward with raw_noise_std > 0: the density that went through relu / alpha was sigma + noise*std (NeRFRenderer.h:251-252) */
ORC_API void orc_raw2outputs_backward_noise(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int white_bkgr,
                                            const float *noise, float noise_std, const float *g_rgb, float *g_raw)
{
    raw2outputs_backward_impl(raw, z, d, n, s, c, white_bkgr, noise, noise_std, g_rgb, g_raw);
}

static void raw2outputs_backward_impl(const float *raw, const float *z, const float *d, int64_t n, int s, int c, int white_bkgr,
                                       const float *noise, float noise_std, const float *g_rgb, float *g_raw)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        const float *dv = d + i * 3;
        const float nrm = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        float alpha[1024], trans[1024], x_[1024], lt[1024], col[1024][3];
        double logt = 0.0;
        float tprev = 0.0f;
        for (int j = 0; j < s; j++) {            /* forward, as orc_raw2outputs */
            const float *r = raw + (i * s + j) * c;
            float dist = (j + 1 < s) ? (z[i * s + j + 1] - z[i * s + j]) : 1e10f;
            dist = dist * nrm;
            float sraw = r[3];
            if (noise) sraw = sraw + noise[i * s + j] * noise_std;
            const float sig = sraw > 0.0f ? sraw : 0.0f;
            x_[j] = -sig * dist;
            alpha[j] = -nrf_expf(x_[j]) + 1.0f;
            lt[j] = tprev;
            trans[j] = nrf_expf(tprev);
            const float om = 1.0f - alpha[j];
            logt += (double)nrf_logf(om > 1e-10f ? om : 1e-10f);
            tprev = (float)logt;
            for (int k = 0; k < 3; k++) col[j][k] = nrf_sigmoidf(r[k]);
        }
        const float gsum = g_rgb[i * 3] + g_rgb[i * 3 + 1] + g_rgb[i * 3 + 2];
        double suffix = 0.0;                      /* sum_{j > k} g_L[j] */
        for (int j = s - 1; j >= 0; j--) {
            const float *r = raw + (i * s + j) * c;
            float *g = g_raw + (i * s + j) * c;
            for (int k = 0; k < c; k++) g[k] = 0.0f;
            float gw = g_rgb[i * 3] * col[j][0] + g_rgb[i * 3 + 1] * col[j][1] + g_rgb[i * 3 + 2] * col[j][2];
            if (white_bkgr) gw -= gsum;           /* rgb += 1 - acc */
            const float w = alpha[j] * trans[j];
            for (int k = 0; k < 3; k++) g[k] = g_rgb[i * 3 + k] * w * (col[j][k] * (1.0f - col[j][k]));
            float g_alpha = gw * trans[j];
            const float om = 1.0f - alpha[j];
            if (om >= 1e-10f) g_alpha -= (float)suffix / om;                                   /* l_j = log(clamp_min(1 - alpha_j)) feeds every later T */
            const float cl = lt[j] < -100.0f ? -100.0f : (lt[j] > 5.0f ? 5.0f : lt[j]);
            suffix += (double)(gw * alpha[j] * nrf_expf(cl));                                   /* g_L[j] = g_T[j] * exp(clamp(L_j)) */
            const float cx = x_[j] < -100.0f ? -100.0f : (x_[j] > 5.0f ? 5.0f : x_[j]);
            const float g_x = -g_alpha * nrf_expf(cx);
            float dist = (j + 1 < s) ? (z[i * s + j + 1] - z[i * s + j]) : 1e10f;
            dist = dist * nrm;
            float sraw = r[3];
            if (noise) sraw = sraw + noise[i * s + j] * noise_std;
            g[3] = (sraw > 0.0f) ? -g_x * dist : 0.0f;
        }
    }
}

/* Backward of NeRFSmallImpl::forward (bias-free).  g_out [p,4] = d loss / d (rgb, sigma).  Accumulates d loss / d params into
 * g_params (same blob layout; caller zeroes) and writes d loss / d x[:, :in_ch] (the position features) to g_x [p, in_ch]. */
ORC_API void orc_mlp_small_backward(const float *params, const float *x, const float *g_out, int64_t p, int in_ch, int in_views, int n_layers,
                                    int hidden, int geo, int n_layers_c, int hidden_c, float *g_params, float *g_x)
{
    const int64_t np_ = orc_mlp_small_param_count(in_ch, in_views, n_layers, hidden, geo, n_layers_c, hidden_c);
    double *gacc = (double *)calloc((size_t)np_, sizeof(double));
    /* sequential over points: a deterministic double accumulation (the oracle is a checker, not a fast path) */
    for (int64_t i = 0; i < p; i++) {
        float act[16][512];                 /* inputs of every layer */
        int dims[17];
        const float *xi = x + i * (in_ch + in_views);
        const float *w = params;
        const float *wl[16];
        int nl = 0;
        memcpy(act[0], xi, sizeof(float) * in_ch); dims[0] = in_ch;
        for (int l = 0; l < n_layers; l++) {
            const int od = (l == n_layers - 1) ? (1 + geo) : hidden;
            wl[nl] = w;
            linear(w, NULL, act[nl], dims[nl], od, act[nl + 1], l != n_layers - 1);
            w += (int64_t)dims[nl] * od; dims[nl + 1] = od; nl++;
        }
        const int sig_layer_out = nl;       /* act[nl] = h (1 + geo) */
        float cin[512];
        for (int k = 0; k < in_views; k++) cin[k] = xi[in_ch + k];
        for (int k = 0; k < geo; k++) cin[in_views + k] = act[sig_layer_out][1 + k];
        const int c0 = nl + 1;              /* colour layers use act[c0 ...] */
        memcpy(act[c0], cin, sizeof(float) * (in_views + geo)); dims[c0] = in_views + geo;
        int cl = c0;
        for (int l = 0; l < n_layers_c; l++) {
            const int od = (l == n_layers_c - 1) ? 3 : hidden_c;
            wl[cl] = w;
            linear(w, NULL, act[cl], dims[cl], od, act[cl + 1], l != n_layers_c - 1);
            w += (int64_t)dims[cl] * od; dims[cl + 1] = od; cl++;
        }
        /* ---- backward ---- */
        float g[512], gin[512];
        g[0] = g_out[i * 4]; g[1] = g_out[i * 4 + 1]; g[2] = g_out[i * 4 + 2];
        for (int l = cl - 1; l >= c0; l--) {                 /* colour net, last layer first */
            const int id = dims[l], od = dims[l + 1];
            if (l != cl - 1) for (int o = 0; o < od; o++) if (!(act[l + 1][o] > 0.0f)) g[o] = 0.0f;     /* ReLU */
            double *ga = gacc + (wl[l] - params);
            for (int o = 0; o < od; o++) for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)g[o] * (double)act[l][k];
            for (int k = 0; k < id; k++) { float a = 0.0f; for (int o = 0; o < od; o++) a += wl[l][(int64_t)o * id + k] * g[o]; gin[k] = a; }
            memcpy(g, gin, sizeof(float) * id);
        }
        /* g now holds d / d cat[views, geo]; the sigma net's output gradient = (g_sigma, g_geo) */
        float gh[512];
        gh[0] = g_out[i * 4 + 3];
        for (int k = 0; k < geo; k++) gh[1 + k] = g[in_views + k];
        memcpy(g, gh, sizeof(float) * (1 + geo));
        for (int l = sig_layer_out - 1; l >= 0; l--) {
            const int id = dims[l], od = dims[l + 1];
            if (l != sig_layer_out - 1) for (int o = 0; o < od; o++) if (!(act[l + 1][o] > 0.0f)) g[o] = 0.0f;
            double *ga = gacc + (wl[l] - params);
            for (int o = 0; o < od; o++) for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)g[o] * (double)act[l][k];
            for (int k = 0; k < id; k++) { float a = 0.0f; for (int o = 0; o < od; o++) a += wl[l][(int64_t)o * id + k] * g[o]; gin[k] = a; }
            memcpy(g, gin, sizeof(float) * id);
        }
        if (g_x) memcpy(g_x + i * in_ch, g, sizeof(float) * in_ch);
    }
    for (int64_t k = 0; k < np_; k++) g_params[k] += (float)gacc[k];
    free(gacc);
}

/* Backward of NeRFImpl::forward (NeRF.cpp:92-126: pts_linears with the skip concat cat[input_pts, h] after layer `skip`, biases everywhere; with view directions
 * alpha_linear(h), feature_linear(h) (no ReLU), relu(views_linears_0(cat[feature, views])), rgb_linear; without: output_linear(cat[h, input_pts])).
 * g_out [p, od] = d loss / d output (viewdirs: (rgb, alpha), od = 4).  Accumulates into g_params (blob layout of orc_mlp_nerf; caller zeroes); g_x [p, in_ch] =
 * d loss / d input_pts (may be NULL; the positional encodings have no parameters).  Sequential over points, double accumulation. */
ORC_API void orc_mlp_nerf_backward(const float *params, const float *x, const float *g_out, int64_t p, int d, int w, int in_ch, int in_views, int out_ch, int skip,
                                   int use_viewdirs, float *g_params, float *g_x)
{
    const int64_t np_ = orc_mlp_nerf_param_count(d, w, in_ch, in_views, out_ch, skip, use_viewdirs);
    double *gacc = (double *)calloc((size_t)np_, sizeof(double));
    const int xd = in_ch + (use_viewdirs ? in_views : 0);
    const int od = use_viewdirs ? 4 : out_ch;
    const int maxw = w + in_ch + in_views + 8;
    float *act = (float *)malloc(sizeof(float) * (size_t)(d + 1) * maxw);      /* act[l] = input of pts layer l; act[d] = h after the last one (cat form where it applies) */
    for (int64_t i = 0; i < p; i++) {
        const float *xi = x + i * xd;
        const float *wl[64]; int dims[65];
        const float *wp = params;
        memcpy(act, xi, sizeof(float) * in_ch); dims[0] = in_ch;
        for (int l = 0; l < d; l++) {
            float *dst = act + (size_t)(l + 1) * maxw;
            const int off = (l == skip) ? in_ch : 0;
            wl[l] = wp;
            linear(wp, wp + (int64_t)dims[l] * w, act + (size_t)l * maxw, dims[l], w, dst + off, 1);
            wp += (int64_t)dims[l] * w + w;
            if (off) memcpy(dst, xi, sizeof(float) * in_ch);
            dims[l + 1] = w + off;
        }
        const float *hl = act + (size_t)d * maxw + (dims[d] - w);             /* the last layer's own w outputs (behind a cat, if the last layer is the skip layer) */
        float gh[2048];                                                        /* d loss / d (that layer's w post-ReLU outputs) */
        float gx_extra[1024]; memset(gx_extra, 0, sizeof(float) * in_ch);
        if (use_viewdirs) {
            const float *vw = wp; wp += (int64_t)(in_views + w) * (w / 2) + w / 2;
            const float *fw = wp; wp += (int64_t)w * w + w;
            const float *aw = wp; wp += w + 1;
            const float *rw = wp;
            float feat[2048], hv[1024];
            linear(fw, fw + (int64_t)w * w, hl, w, w, feat, 0);
            memcpy(feat + w, xi + in_ch, sizeof(float) * in_views);
            linear(vw, vw + (int64_t)(in_views + w) * (w / 2), feat, w + in_views, w / 2, hv, 1);
            const float *go = g_out + i * od;
            /* rgb_linear */
            float ghv[1024];
            { double *ga = gacc + (rw - params);
              for (int o = 0; o < 3; o++) { for (int k = 0; k < w / 2; k++) ga[(int64_t)o * (w / 2) + k] += (double)go[o] * (double)hv[k]; ga[(int64_t)3 * (w / 2) + o] += (double)go[o]; }
              for (int k = 0; k < w / 2; k++) { float a = 0.0f; for (int o = 0; o < 3; o++) a += rw[(int64_t)o * (w / 2) + k] * go[o]; ghv[k] = (hv[k] > 0.0f) ? a : 0.0f; } }
            /* views_linears_0 on cat[feature, views] */
            float gfeat[2048];
            { const int id = w + in_views; double *ga = gacc + (vw - params);
              for (int o = 0; o < w / 2; o++) { for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)ghv[o] * (double)feat[k]; ga[(int64_t)(w / 2) * id + o] += (double)ghv[o]; }
              for (int k = 0; k < w; k++) { float a = 0.0f; for (int o = 0; o < w / 2; o++) a += vw[(int64_t)o * id + k] * ghv[o]; gfeat[k] = a; } }
            /* feature_linear (no ReLU) and alpha_linear, both on h */
            { double *gf = gacc + (fw - params), *gal = gacc + (aw - params);
              for (int o = 0; o < w; o++) { for (int k = 0; k < w; k++) gf[(int64_t)o * w + k] += (double)gfeat[o] * (double)hl[k]; gf[(int64_t)w * w + o] += (double)gfeat[o]; }
              for (int k = 0; k < w; k++) gal[k] += (double)go[3] * (double)hl[k];
              gal[w] += (double)go[3];
              for (int k = 0; k < w; k++) { float a = 0.0f; for (int o = 0; o < w; o++) a += fw[(int64_t)o * w + k] * gfeat[o]; gh[k] = a + aw[k] * go[3]; } }
        } else {
            const float *ow = wp;
            const float *go = g_out + i * od;
            const int id = w + in_ch;
            float hc[2048];
            memcpy(hc, hl, sizeof(float) * w); memcpy(hc + w, xi, sizeof(float) * in_ch);
            double *ga = gacc + (ow - params);
            for (int o = 0; o < out_ch; o++) { for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)go[o] * (double)hc[k]; ga[(int64_t)out_ch * id + o] += (double)go[o]; }
            for (int k = 0; k < w; k++) { float a = 0.0f; for (int o = 0; o < out_ch; o++) a += ow[(int64_t)o * id + k] * go[o]; gh[k] = a; }
            for (int k = 0; k < in_ch; k++) { float a = 0.0f; for (int o = 0; o < out_ch; o++) a += ow[(int64_t)o * id + w + k] * go[o]; gx_extra[k] += a; }
        }
        /* pts_linears, last first; gh = gradient of layer l's w post-ReLU outputs */
        for (int l = d - 1; l >= 0; l--) {
            const int id = dims[l];
            const float *in_l = act + (size_t)l * maxw;
            const float *out_l = act + (size_t)(l + 1) * maxw + (dims[l + 1] - w);
            float g[2048], gin[2048];
            for (int o = 0; o < w; o++) g[o] = (out_l[o] > 0.0f) ? gh[o] : 0.0f;
            double *ga = gacc + (wl[l] - params);
            for (int o = 0; o < w; o++) { for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)g[o] * (double)in_l[k]; ga[(int64_t)w * id + o] += (double)g[o]; }
            for (int k = 0; k < id; k++) { float a = 0.0f; for (int o = 0; o < w; o++) a += wl[l][(int64_t)o * id + k] * g[o]; gin[k] = a; }
            if (l == 0) { for (int k = 0; k < in_ch; k++) gx_extra[k] += gin[k]; }
            else if (id == w + in_ch) { for (int k = 0; k < in_ch; k++) gx_extra[k] += gin[k]; memcpy(gh, gin + in_ch, sizeof(float) * w); }      /* input was cat[input_pts, h] */
            else memcpy(gh, gin, sizeof(float) * w);
        }
        if (g_x) memcpy(g_x + i * in_ch, gx_extra, sizeof(float) * in_ch);
    }
    for (int64_t k = 0; k < np_; k++) g_params[k] += (float)gacc[k];
    free(gacc); free(act);
}

/* N1, LeRF branch (NeRFExecutor.h:955-982): lang_loss = huber_loss(RenderedLangEmbedding, target, reduction none, delta).sum(-1).nanmean()  (:970-974).
 * ATen: huber(reduction none) z = |d|: z < delta ? 0.5 z^2 : delta (z - 0.5 delta); nanmean = nansum / (count of non-NaN rows); the backward of nansum passes a ZERO
 * to a NaN row, and huber's backward multiplies it by its own derivative -- NaN where the difference is NaN: such a ray's gradient row is 0 except NaN at the NaN
 * elements (golden train_lerf_nan).  loss: 1 float; grad [n, e] = d loss / d pred. */
ORC_API void orc_huber_rows_nanmean(const float *pred, const float *target, int64_t n, int e, float delta, float *loss, float *grad)
{
    double acc = 0.0;
    int64_t cnt = 0;
    unsigned char *isnan_row = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int64_t i = 0; i < n; i++) {
        double row = 0.0;
        for (int k = 0; k < e; k++) {
            const float d = pred[i * e + k] - target[i * e + k];
            const float z = fabsf(d);
            row += (z < delta) ? 0.5 * (double)z * (double)z : (double)delta * ((double)z - 0.5 * (double)delta);
        }
        const float rowf = (float)row;
        if (rowf != rowf) isnan_row[i] = 1; else { acc += (double)rowf; cnt++; }
    }
    if (loss) *loss = (float)(acc / (double)cnt);                       /* 0 / 0 = NaN when every row is NaN, as nanmean */
    if (grad) {
        const float norm = 1.0f / (float)cnt;
        for (int64_t i = 0; i < n; i++) {
            const float go = isnan_row[i] ? 0.0f : norm;
            for (int k = 0; k < e; k++) {
                const float d = pred[i * e + k] - target[i * e + k];
                grad[i * e + k] = (d < -delta) ? -delta * go : (d > delta ? delta * go : d * go);        /* NaN d: the last branch, NaN * 0 = NaN */
            }
        }
    }
    free(isnan_row);
}

/* Backward of the FINE pass of LeRFRenderer::RenderRays downstream of the language grid, w.r.t. the LeRF head's parameters and the grid's features:
 *   emb [n*s, in] -> LeRFImpl::forward (LeRF.cpp:86-108) -> sigma_le[~keep] = 0 (LeRFRenderer.cpp:37-38) -> RawToLEOutputs' weights (:38-66, = C1's) ->
 *   RenderCLIPEmbedding (LeRFRenderer.h:45-54) -> rendered [n, E];  given g_rendered = d loss / d rendered.
 * normalize(x, eps) = x / clamp_min(||x||, eps): its backward is g / c - [||x|| >= eps] (g . x / c^2) x / ||x||, c = max(||x||, eps).
 * Weights backward as raw2outputs_backward_impl above with d loss / d w given directly (TruncExp::backward = grad * exp(clamp(x, -100, 5)), CustomOps.cpp:11-15).
 * Accumulates into g_params (blob layout of orc_lerf; caller zeroes); writes g_emb [n*s, in] (may be NULL), rendered [n, E] and weights [n, s] (may be NULL). */
ORC_API void orc_lerf_head_backward(const float *params, const float *emb, const unsigned char *keep, const float *z, const float *d, int64_t n, int s,
                                    int in_ch, int n_layers, int hidden, int geo, int embed, const float *noise /*[n,s] or NULL: the RawNoiseStd draws (LeRFRenderer.cpp:50-51)*/,
                                    float noise_std, const float *g_rendered, float *g_params, float *g_emb, float *rendered, float *weights_out)
{
    int64_t np_ = 0;
    { int cd = in_ch; for (int l = 0; l < n_layers; l++) { int od = (l == n_layers - 1) ? (1 + geo) : hidden; np_ += (int64_t)cd * od; cd = od; }
      cd = geo + in_ch; for (int l = 0; l < n_layers; l++) { int od = (l == n_layers - 1) ? embed : hidden; np_ += (int64_t)cd * od; cd = od; } }
    double *gacc = (double *)calloc((size_t)np_, sizeof(double));
    const int NL = 2 * n_layers;
    const int maxw = (hidden > embed ? hidden : embed) > (geo + in_ch + 1) ? (hidden > embed ? hidden : embed) : (geo + in_ch + 1);
    for (int64_t i = 0; i < n; i++) {                                     /* sequential over rays: deterministic double accumulation */
        /* ---- forward of the ray's s samples, every layer input kept ---- */
        float *act = (float *)calloc((size_t)s * (NL + 1) * maxw, sizeof(float));       /* act[j][l] = input of layer l (l = NL: the last layer's output h) */
        float *le = (float *)malloc(sizeof(float) * (size_t)s * embed), *nrm = (float *)malloc(sizeof(float) * s), *sig = (float *)malloc(sizeof(float) * s);
        const float *wl[32]; int dims[33];
        for (int j = 0; j < s; j++) {
            const float *xi = emb + (i * s + j) * in_ch;
            float *A = act + (size_t)j * (NL + 1) * maxw;
            const float *w = params;
            memcpy(A, xi, sizeof(float) * in_ch); dims[0] = in_ch;
            for (int l = 0; l < n_layers; l++) {
                const int od = (l == n_layers - 1) ? (1 + geo) : hidden;
                wl[l] = w;
                linear(w, NULL, A + (size_t)l * maxw, dims[l], od, A + (size_t)(l + 1) * maxw, l != n_layers - 1);
                w += (int64_t)dims[l] * od; dims[l + 1] = od;
            }
            /* act[n_layers] = (sigma, geo): the LE net's input replaces it in a separate slot: layer n_layers reads cat[geo, in] */
            float *h33 = A + (size_t)n_layers * maxw;
            sig[j] = (keep && !keep[i * s + j]) ? 0.0f : h33[0];
            if (noise) sig[j] = sig[j] + noise[i * s + j] * noise_std;     /* the density that goes through relu / alpha */
            float cin[4096];
            for (int k = 0; k < geo; k++) cin[k] = h33[1 + k];
            for (int k = 0; k < in_ch; k++) cin[geo + k] = xi[k];
            float keep33[4096]; memcpy(keep33, h33, sizeof(float) * (1 + geo));
            /* slots: [0..n_layers-1] inputs of the sigma layers; slot n_layers: cat[geo, in] (input of LE layer 0); h33 itself is not needed again (ReLU-free output) */
            memcpy(h33, cin, sizeof(float) * (geo + in_ch));
            int cd = geo + in_ch;
            for (int l = 0; l < n_layers; l++) {
                const int od = (l == n_layers - 1) ? embed : hidden;
                wl[n_layers + l] = w;
                linear(w, NULL, A + (size_t)(n_layers + l) * maxw, cd, od, A + (size_t)(n_layers + l + 1) * maxw, l != n_layers - 1);
                w += (int64_t)cd * od; cd = od;
            }
            const float *h = A + (size_t)NL * maxw;
            double ss = 0.0;
            for (int k = 0; k < embed; k++) ss += (double)h[k] * (double)h[k];
            nrm[j] = (float)sqrt(ss);
            const float c = f_max(nrm[j], 1e-8f);
            for (int k = 0; k < embed; k++) le[(size_t)j * embed + k] = h[k] / c;
        }
        /* ---- weights (as raw2outputs_impl) ---- */
        const float *dv = d + i * 3;
        const float dn = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        float *alpha = (float *)malloc(sizeof(float) * s), *trans = (float *)malloc(sizeof(float) * s), *x_ = (float *)malloc(sizeof(float) * s), *lt = (float *)malloc(sizeof(float) * s),
              *wgt = (float *)malloc(sizeof(float) * s), *gw = (float *)malloc(sizeof(float) * s);
        double logt = 0.0; float tprev = 0.0f;
        for (int j = 0; j < s; j++) {
            float dist = (j + 1 < s) ? (z[i * s + j + 1] - z[i * s + j]) : 1e10f;
            dist = dist * dn;
            const float sg = sig[j] > 0.0f ? sig[j] : 0.0f;
            x_[j] = -sg * dist;
            alpha[j] = -nrf_expf(x_[j]) + 1.0f;
            lt[j] = tprev; trans[j] = nrf_expf(tprev);
            const float om = 1.0f - alpha[j];
            logt += (double)nrf_logf(om > 1e-10f ? om : 1e-10f);
            tprev = (float)logt;
            wgt[j] = alpha[j] * trans[j];
            if (weights_out) weights_out[i * s + j] = wgt[j];
        }
        /* ---- RenderCLIPEmbedding forward + backward ---- */
        float *v = (float *)malloc(sizeof(float) * embed), *gv = (float *)malloc(sizeof(float) * embed);
        double vss = 0.0;
        for (int k = 0; k < embed; k++) { double a = 0.0; for (int j = 0; j < s; j++) a += (double)(wgt[j] * le[(size_t)j * embed + k]); v[k] = (float)a; vss += (double)v[k] * (double)v[k]; }
        const float vn = (float)sqrt(vss), vc = f_max(vn, 1e-8f);
        double gdot = 0.0;
        for (int k = 0; k < embed; k++) { if (rendered) rendered[i * embed + k] = v[k] / vc; gdot += (double)g_rendered[i * embed + k] * (double)v[k]; }
        for (int k = 0; k < embed; k++) {
            float g = g_rendered[i * embed + k] / vc;
            if (vn >= 1e-8f && vn > 0.0f) g -= (float)(gdot / ((double)vc * (double)vc)) * (v[k] / vn);
            gv[k] = g;
        }
        /* per sample: g_w = le . g_v ; g_le = w g_v ; g_h = normalize backward */
        float *gh = (float *)malloc(sizeof(float) * (size_t)s * embed);
        for (int j = 0; j < s; j++) {
            double t = 0.0;
            for (int k = 0; k < embed; k++) t += (double)le[(size_t)j * embed + k] * (double)gv[k];
            gw[j] = (float)t;
            const float c = f_max(nrm[j], 1e-8f);
            const float *h = act + (size_t)j * (NL + 1) * maxw + (size_t)NL * maxw;
            double hdot = 0.0;                                             /* g_le . h = w (g_v . h) */
            for (int k = 0; k < embed; k++) hdot += (double)(wgt[j] * gv[k]) * (double)h[k];
            for (int k = 0; k < embed; k++) {
                float g = wgt[j] * gv[k] / c;
                if (nrm[j] >= 1e-8f && nrm[j] > 0.0f) g -= (float)(hdot / ((double)c * (double)c)) * (h[k] / nrm[j]);
                gh[(size_t)j * embed + k] = g;
            }
        }
        /* weights backward -> g_sigma */
        float *gsig = (float *)malloc(sizeof(float) * s);
        double suffix = 0.0;
        for (int j = s - 1; j >= 0; j--) {
            float g_alpha = gw[j] * trans[j];
            const float om = 1.0f - alpha[j];
            if (om >= 1e-10f) g_alpha -= (float)suffix / om;
            const float cl = lt[j] < -100.0f ? -100.0f : (lt[j] > 5.0f ? 5.0f : lt[j]);
            suffix += (double)(gw[j] * alpha[j] * nrf_expf(cl));
            const float cx = x_[j] < -100.0f ? -100.0f : (x_[j] > 5.0f ? 5.0f : x_[j]);
            const float g_x = -g_alpha * nrf_expf(cx);
            float dist = (j + 1 < s) ? (z[i * s + j + 1] - z[i * s + j]) : 1e10f;
            dist = dist * dn;
            gsig[j] = (sig[j] > 0.0f) ? -g_x * dist : 0.0f;
            if (keep && !keep[i * s + j]) gsig[j] = 0.0f;                 /* index_put_ of a constant: no gradient to the masked sigma */
        }
        /* ---- the two nets, per sample ---- */
        for (int j = 0; j < s; j++) {
            float *A = act + (size_t)j * (NL + 1) * maxw;
            float g[4096], gin[4096];
            memcpy(g, gh + (size_t)j * embed, sizeof(float) * embed);
            int od = embed;
            for (int l = NL - 1; l >= n_layers; l--) {                     /* LE net, last layer first */
                const int id = (l == n_layers) ? (geo + in_ch) : hidden;
                const float *in_l = A + (size_t)l * maxw, *out_l = A + (size_t)(l + 1) * maxw;
                if (l != NL - 1) for (int o = 0; o < od; o++) if (!(out_l[o] > 0.0f)) g[o] = 0.0f;
                double *ga = gacc + (wl[l] - params);
                for (int o = 0; o < od; o++) for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)g[o] * (double)in_l[k];
                for (int k = 0; k < id; k++) { float a = 0.0f; for (int o = 0; o < od; o++) a += wl[l][(int64_t)o * id + k] * g[o]; gin[k] = a; }
                memcpy(g, gin, sizeof(float) * id); od = id;
            }
            float ge_a[4096];                                              /* d / d emb through the LE net's cat[geo, in] */
            memcpy(ge_a, g + geo, sizeof(float) * in_ch);
            float g33[4096];
            g33[0] = gsig[j];
            for (int k = 0; k < geo; k++) g33[1 + k] = g[k];
            memcpy(g, g33, sizeof(float) * (1 + geo)); od = 1 + geo;
            for (int l = n_layers - 1; l >= 0; l--) {
                const int id = (l == 0) ? in_ch : hidden;
                const float *in_l = A + (size_t)l * maxw, *out_l = A + (size_t)(l + 1) * maxw;
                /* slot n_layers was overwritten with the LE input: the sigma net's last layer has no ReLU, so its output is not needed for a mask */
                if (l != n_layers - 1) for (int o = 0; o < od; o++) if (!(out_l[o] > 0.0f)) g[o] = 0.0f;
                double *ga = gacc + (wl[l] - params);
                for (int o = 0; o < od; o++) for (int k = 0; k < id; k++) ga[(int64_t)o * id + k] += (double)g[o] * (double)in_l[k];
                for (int k = 0; k < id; k++) { float a = 0.0f; for (int o = 0; o < od; o++) a += wl[l][(int64_t)o * id + k] * g[o]; gin[k] = a; }
                memcpy(g, gin, sizeof(float) * id); od = id;
            }
            if (g_emb) for (int k = 0; k < in_ch; k++) g_emb[(i * s + j) * in_ch + k] = g[k] + ge_a[k];
        }
        free(act); free(le); free(nrm); free(sig); free(alpha); free(trans); free(x_); free(lt); free(wgt); free(gw); free(v); free(gv); free(gh); free(gsig);
    }
    for (int64_t k = 0; k < np_; k++) g_params[k] += (float)gacc[k];
    free(gacc);
}

/* Backward of HashEmbedderImpl::forward w.r.t. the embedding tables (NeRF.cpp:279-298 trilinear blend; nn::Embedding backward =
 * index_add of the row gradients).  g_emb [p, L*F]; g_table [L][2^T][F] accumulated (caller zeroes). */
ORC_API void orc_hash_ngp_backward(const float *x, int64_t p, const float *bbox, int n_levels, int n_feat, int log2_t, int base, int finest,
                                   const float *g_emb, float *g_table)
{
    float res[64];
    orc_hash_ngp_resolutions(n_levels, base, finest, res);
    const int64_t tsize = (int64_t)1 << log2_t, hmask = tsize - 1;
    double *acc = (double *)calloc((size_t)(n_levels * tsize * n_feat), sizeof(double));
    for (int64_t i = 0; i < p; i++) {
        float xc[3];
        for (int a = 0; a < 3; a++) xc[a] = f_max(f_min(x[i * 3 + a], bbox[3 + a]), bbox[a]);
        for (int l = 0; l < n_levels; l++) {
            float w[3];
            int64_t idx[3];
            for (int a = 0; a < 3; a++) {
                float grid = (bbox[3 + a] - bbox[a]) / res[l];
                idx[a] = (int64_t)floorf((xc[a] - bbox[a]) / grid);
                float vmin = (float)idx[a] * grid + bbox[a];
                float vmax = vmin + grid;
                w[a] = (x[i * 3 + a] - vmin) / (vmax - vmin);
            }
            for (int c = 0; c < 8; c++) {
                int64_t cx = idx[0] + ((c >> 2) & 1), cy = idx[1] + ((c >> 1) & 1), cz = idx[2] + (c & 1);
                int64_t hsh = ((cx * 1LL) ^ (cy * 2654435761LL) ^ (cz * 805459861LL)) & hmask;
                const float wx = ((c >> 2) & 1) ? w[0] : 1.0f - w[0], wy = ((c >> 1) & 1) ? w[1] : 1.0f - w[1], wz = (c & 1) ? w[2] : 1.0f - w[2];
                for (int f = 0; f < n_feat; f++) {
                    const float g = g_emb[i * n_levels * n_feat + l * n_feat + f];
                    acc[((int64_t)l * tsize + hsh) * n_feat + f] += (double)(((g * wz) * wy) * wx);     /* autograd's chain: c -> c0/c1 -> c00.. -> e */
                }
            }
        }
    }
    for (int64_t k = 0; k < (int64_t)n_levels * tsize * n_feat; k++) g_table[k] += (float)acc[k];
    free(acc);
}

/* H3  CuHashEmbedderBackwardKernel         CuHashEmbedder.cu:105-216, host :277-325   RESTATEMENT-PINNED (CUDA-only unit)
 *     grad_in = fp16(g * 128) (:297) ; corner contribution = fp16(grad_in * w) (:196-197) added to the pool at the forward's
 *     element offset (the overlap quirk again, :154) ; result = pool / 128 (:323).
 *     The reference adds with fp16 atomics into an fp16 pool (order-dependent, loses small addends); the restatement keeps its
 *     per-contribution roundings and accumulates exactly (double), which is the definition the HIP kernel implements in fp32. */
ORC_API void orc_hash_cu_backward(const float *x, int64_t p, const int32_t *primes, const int32_t *local_idx, const int32_t *local_size,
                                  const float *bias, const float *bbox, const float *mul, int n_levels, int n_feat, int64_t pool_elems,
                                  const float *g_emb /*[p, L*F]*/, float *g_table /*[pool_elems] accumulated*/)
{
    double *acc = (double *)calloc((size_t)pool_elems, sizeof(double));
    for (int64_t i = 0; i < p; i++) {
        float xc[3];
        for (int a = 0; a < 3; a++) xc[a] = f_max(f_min(x[i * 3 + a], bbox[3 + a]), bbox[a]);
        for (int l = 0; l < n_levels; l++) {
            float pt[3], fl[3];
            uint32_t pos[3];
            for (int a = 0; a < 3; a++) {
                pt[a] = (xc[a] - bbox[a]) / (bbox[3 + a] - bbox[a]) * mul[l];
                pt[a] = pt[a] + bias[l * 3 + a];
                fl[a] = floorf(pt[a]);
                pos[a] = (uint32_t)fl[a];
            }
            const uint32_t pa = (uint32_t)primes[l * 3 + 0], pb = (uint32_t)primes[l * 3 + 1], pc = (uint32_t)primes[l * 3 + 2];
            const uint32_t lsz = (uint32_t)local_size[l];
            const float a = pt[0] - fl[0], b = pt[1] - fl[1], c = pt[2] - fl[2];
            for (int k = 0; k < 8; k++) {
                uint32_t dx = (k >> 2) & 1u, dy = (k >> 1) & 1u, dz = k & 1u;
                uint32_t ps = (((pos[0] + dx) * pa) ^ ((pos[1] + dy) * pb) ^ ((pos[2] + dz) * pc)) % lsz;
                float w = (dx ? a : (1.0f - a)) * (dy ? b : (1.0f - b)) * (dz ? c : (1.0f - c));
                for (int f = 0; f < n_feat; f++) {
                    const float gin = f16_bits_to_f32(f32_to_f16_bits(g_emb[i * n_levels * n_feat + l * n_feat + f] * 128.0f));
                    if (gin == 0.0f) continue;
                    const float contrib = f16_bits_to_f32(f32_to_f16_bits(gin * w));
                    acc[(int64_t)local_idx[l] + (int64_t)ps * n_feat + f] += (double)(contrib * (1.0f / 128.0f));
                }
            }
        }
    }
    for (int64_t k = 0; k < pool_elems; k++) g_table[k] += (float)acc[k];
    free(acc);
}

/* TotalVariationLoss of the LibTorch HashEmbedder (NeRF.h:255-300; added to the training loss for the first half of the iterations,
 * NeRFExecutor.h:896-913): a cube of (cube+1)^3 lattice vertices starting at min_vertex, the level's table rows at their hashes,
 * loss = (sum of squared forward differences along x, y, z) / cube.  grad (optional) accumulates weight * d loss / d table. */
ORC_API void orc_tv_loss(const float *table_level, int log2_t, int n_feat, const int32_t *min_vertex, int cube, float weight, float *loss, float *grad)
{
    const int n = cube + 1;
    const int64_t hmask = ((int64_t)1 << log2_t) - 1;
    int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * n * n * n);
    for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) for (int z = 0; z < n; z++) {
        const int64_t cx = min_vertex[0] + x, cy = min_vertex[1] + y, cz = min_vertex[2] + z;
        rows[((int64_t)x * n + y) * n + z] = ((cx * 1LL) ^ (cy * 2654435761LL) ^ (cz * 805459861LL)) & hmask;
    }
    double tv = 0.0;
    const int64_t stride[3] = {(int64_t)n * n, n, 1};
    for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) for (int z = 0; z < n; z++) {
        const int64_t v = ((int64_t)x * n + y) * n + z;
        const int c[3] = {x, y, z};
        for (int a = 0; a < 3; a++) {
            if (c[a] + 1 >= n) continue;
            const int64_t u = v + stride[a];
            for (int f = 0; f < n_feat; f++) {
                const float d = table_level[rows[u] * n_feat + f] - table_level[rows[v] * n_feat + f];
                tv += (double)(d * d);
                if (grad) {
                    const float g = weight * (2.0f * d) / (float)cube;
                    grad[rows[u] * n_feat + f] += g;
                    grad[rows[v] * n_feat + f] -= g;
                }
            }
        }
    }
    if (loss) *loss = (float)(tv / (double)cube);
    free(rows);
}

/* torch::optim::Adam::step (no weight decay, no amsgrad), step count t >= 1:
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps) */
ORC_API void orc_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr, float b1, float b2, float eps, int t)
{
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const double step_size = (double)lr / bc1, bc2s = sqrt(bc2);
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        m[i] = m[i] * b1 + g[i] * (1.0f - b1);                /* exp_avg.lerp_(grad, 1 - beta1) == this up to 1 ulp */
        v[i] = v[i] * b2 + (g[i] * g[i]) * (1.0f - b2);       /* exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2) */
        const float denom = sqrtf(v[i]) / (float)bc2s + eps;   /* (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps), fp32 tensor ops */
        p[i] = p[i] - (float)step_size * (m[i] / denom);
    }
}

/* ------------------------------------------------------------------------------------------
 * N4  image-space tail of RenderPath       NeRFExecutor.h:690,698-700 ; TorchTensorToCVMat NeRFRenderer.h:58-68
 *     depth' = (depth - Near) / (Far - Near) ; u8 = (uint8)clamp(x*255, 0, 255)   (float -> u8 conversion truncates)
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_normalize_depth(const float *depth, int64_t n, float near_, float far_, float *out)
{
    const float span = far_ - near_;
    for (int64_t i = 0; i < n; i++) out[i] = (depth[i] - near_) / span;
}

ORC_API void orc_to_u8(const float *x, int64_t n, uint8_t *out)
{
    for (int64_t i = 0; i < n; i++) {
        float v = x[i] * 255.0f;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        out[i] = (uint8_t)v;
    }
}

/* ---------------------------------------------------------------------------------------------------
 * Relevancy (LeRFRenderer.cpp:79, NeRFExecutor.h:824) and the relevancy image (NeRFExecutor.h:713-719).
 *
 * PARITY UNPINNED.  `Relevancy(embeds, positives, negatives)` is defined in RuCLIPProcessor.h of the external module DeliriumV01D/RuCLIP
 * (../RuCLIP/src, CMakeLists.Files.txt:8-10; no pinned version, absent from /root/reference), and cv::applyColorMap is OpenCV's (absent from this image).
 * What is restated here is the PUBLISHED algorithm both derive from, anchored on the reference's call sites:
 *   - LERF (Kerr et al., ICCV 2023, section 3.3 "relevancy score"; nerfstudio lerf `get_relevancy`, which RuCLIP's function mirrors statement for statement):
 *       logits = embeds @ cat(positives, negatives)^T                       [N, P + Q]
 *       sims   = stack(repeat(logits[:, positive_id], Q), logits[:, P:])    [N, Q, 2]
 *       smx    = softmax(10 * sims, -1)                                     temperature 10
 *       best   = argmin_q smx[:, q, 0]                                      the canonical phrase the positive loses most against (first on ties)
 *       out    = smx[:, best, :]                                            [N, 2] = (p_positive, p_negative)
 *     The call sites fix the shapes: embeds [N, 768] (L2-normalised rendered embedding), positives [1, 768], negatives [Q, 768] (NeRFExecutor.h:744-752: "[3, 768]"),
 *     output [N, 2] of which column 0 is consumed (LeRFRenderer.h:18: "[num_rays, 2]", take the zeroth; NeRFExecutor.h:714, :825).
 *   - COLORMAP_JET: OpenCV's table (see orc_colormap_jet_lut); cv::Mat channel order is B, G, R.  The input byte is `pos_probs.mul(255).to(torch::kU8)` (NeRFExecutor.h:715): truncation toward zero.
 * ------------------------------------------------------------------------------------------------- */
ORC_API void orc_relevancy(const float *embeds, int64_t n, int e, const float *pos, int p, const float *neg, int q, int positive_id, float *out)
{
    OMP_FOR
    for (int64_t i = 0; i < n; i++) {
        const float *x = embeds + i * (int64_t)e;
        float lp = 0.0f;
        for (int k = 0; k < e; k++) lp += x[k] * pos[(int64_t)positive_id * e + k];
        float best0 = 0.0f, best1 = 0.0f;
        for (int j = 0; j < q; j++) {
            float ln = 0.0f;
            for (int k = 0; k < e; k++) ln += x[k] * neg[(int64_t)j * e + k];
            /* softmax over the pair (10 lp, 10 ln), max-subtracted as ATen's softmax does */
            const float a = 10.0f * lp, b = 10.0f * ln, m = a > b ? a : b;
            const float ea = expf(a - m), eb = expf(b - m), sum = ea + eb;
            const float s0 = ea / sum, s1 = eb / sum;
            if (j == 0 || s0 < best0) { best0 = s0; best1 = s1; }
        }
        out[i * 2 + 0] = best0; out[i * 2 + 1] = best1;
        (void)p;
    }
}

/* lut[256][3] in B, G, R order.  OpenCV's Jet table (modules/imgproc/src/colormap.cpp, 256 float entries per channel; first non-zero red entry 0.00588235294117645 =
   4 * 96 / 255 - 1.5) is Octave's jet(256) evaluated at x = k / 255:
       r = clamp(min(4x - 1.5, -4x + 4.5)),  g = clamp(min(4x - 0.5, -4x + 3.5)),  b = clamp(min(4x + 0.5, -4x + 2.5))       (clamp to [0, 1])
   stored as float literals and converted with Mat::convertTo(CV_8U, 255.): the float product entry * 255.f, rounded to nearest (ties to even), saturated.
   On the ramps entry * 255 = 4k - 382.5 etc. sits on a rounding tie before the float roundings, so without OpenCV at hand single entries may differ by one
   count from its table: part of what "parity unpinned" covers for this function. */
ORC_API void orc_colormap_jet_lut(uint8_t *lut)
{
    for (int k = 0; k < 256; k++) {
        const double x = (double)k / 255.0;
        const double c[3] = {fmin(4.0 * x - 1.5, -4.0 * x + 4.5), fmin(4.0 * x - 0.5, -4.0 * x + 3.5), fmin(4.0 * x + 0.5, -4.0 * x + 2.5)};   /* r, g, b */
        for (int ch = 0; ch < 3; ch++) {
            const float entry = (float)(c[ch] < 0.0 ? 0.0 : (c[ch] > 1.0 ? 1.0 : c[ch]));
            const float scaled = entry * 255.0f;
            long v = lrintf(scaled);
            lut[k * 3 + (2 - ch)] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

/* rel[:, 0] * 255 -> u8 (truncation, saturated) -> JET, [n, 3] B, G, R   (NeRFExecutor.h:713-719) */
ORC_API void orc_relevancy_image(const float *rel, int64_t n, int rel_stride, uint8_t *bgr)
{
    uint8_t lut[256 * 3];
    orc_colormap_jet_lut(lut);
    for (int64_t i = 0; i < n; i++) {
        float v = rel[i * rel_stride] * 255.0f;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        const int b = (int)(uint8_t)v;
        bgr[i * 3 + 0] = lut[b * 3 + 0]; bgr[i * 3 + 1] = lut[b * 3 + 1]; bgr[i * 3 + 2] = lut[b * 3 + 2];
    }
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
