// rays.hip -- ray generation, AABB clipping, ray-batch packing, stratified depths.
// Reference: RayUtils.h (GetDirections/GetRays/NDCRays/IntersectWithAABB), NeRFRenderer.h:393-419, :549-603.
// fp32, one rounding per source-level op (the library is built with -ffp-contract=off) so that every
// value here equals the LibTorch CPU result bit for bit.
#include "common.h"
#include "nrf_rng.h"

namespace nrf {

// ------------------------------------------------------------------------------------------------
// RayUtils.h:5-46.  One thread per pixel; ray index = (y-row0)*w + x.
// ------------------------------------------------------------------------------------------------
__global__ void k_get_rays(int w, int row0, int64_t n, float fx, float cx, float fy, float cy,
                           float r00, float r01, float r02, float r10, float r11, float r12, float r20, float r21, float r22,
                           float t0, float t1, float t2, float *__restrict__ o, float *__restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y = row0 + (int)(i / w);
    const int x = (int)(i % w);
    const float dx = ((float)x - cx) / fx;
    const float dy = -((float)y - cy) / fy;
    const float dz = -1.0f;
    // torch::sum(dirs[..., None, :] * c2w[:3,:3], -1): three products summed left to right
    float a0 = dx * r00; a0 = a0 + dy * r01; a0 = a0 + dz * r02;
    float a1 = dx * r10; a1 = a1 + dy * r11; a1 = a1 + dz * r12;
    float a2 = dx * r20; a2 = a2 + dy * r21; a2 = a2 + dz * r22;
    d[i * 3 + 0] = a0; d[i * 3 + 1] = a1; d[i * 3 + 2] = a2;
    o[i * 3 + 0] = t0; o[i * 3 + 1] = t1; o[i * 3 + 2] = t2;
}

// RayUtils.h:49-83
__global__ void k_ndc_rays(int64_t n, float sx, float sy, float near_, float two_near, float m_two_near,
                           const float *__restrict__ o, const float *__restrict__ d, float *__restrict__ oo, float *__restrict__ od)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ox = o[i * 3], oy = o[i * 3 + 1], oz = o[i * 3 + 2];
    const float dx = d[i * 3], dy = d[i * 3 + 1], dz = d[i * 3 + 2];
    const float t = -(near_ + oz) / dz;
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
    oo[i * 3 + 0] = sx * ox / oz;
    oo[i * 3 + 1] = sy * oy / oz;
    oo[i * 3 + 2] = 1.0f + two_near / oz;
    od[i * 3 + 0] = sx * (dx / dz - ox / oz);
    od[i * 3 + 1] = sy * (dy / dz - oy / oz);
    od[i * 3 + 2] = m_two_near / oz;
}

// RayUtils.h:87-126
__device__ __forceinline__ void aabb_one(const float *o, const float *d, const Bbox &bb, float near_plane, float &nr, float &fr)
{
    float tmin[3], tmax[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float inv = 1.0f / (d[a] + 1e-6f);
        const float t1 = (bb.mn[a] - o[a]) * inv;
        const float t2 = (bb.mx[a] - o[a]) * inv;
        tmin[a] = fminf(t1, t2);
        tmax[a] = fmaxf(t1, t2);
    }
    nr = fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]);
    fr = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
    nr = fmaxf(nr, near_plane);
    fr = fmaxf(fr, nr + 1e-6f);
}

__global__ void k_aabb(int64_t n, Bbox bb, float near_plane, const float *__restrict__ o, const float *__restrict__ d,
                       float *__restrict__ nears, float *__restrict__ fars)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float oo[3] = {o[i * 3], o[i * 3 + 1], o[i * 3 + 2]};
    float dd[3] = {d[i * 3], d[i * 3 + 1], d[i * 3 + 2]};
    float nr, fr;
    aabb_one(oo, dd, bb, near_plane, nr, fr);
    nears[i] = nr; fars[i] = fr;
}

// NeRFRenderer.h:549-583
__global__ void k_pack_rays(int64_t n, Bbox bb, int use_viewdirs, const float *__restrict__ o, const float *__restrict__ d,
                            float *__restrict__ rays)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int stride = use_viewdirs ? 11 : 8;
    float oo[3] = {o[i * 3], o[i * 3 + 1], o[i * 3 + 2]};
    float dd[3] = {d[i * 3], d[i * 3 + 1], d[i * 3 + 2]};
    float nr, fr;
    aabb_one(oo, dd, bb, 0.0f, nr, fr);
    float *r = rays + i * stride;
    r[0] = oo[0]; r[1] = oo[1]; r[2] = oo[2];
    r[3] = dd[0]; r[4] = dd[1]; r[5] = dd[2];
    r[6] = nr; r[7] = fr;
    if (use_viewdirs) {
        float s = dd[0] * dd[0]; s = s + dd[1] * dd[1]; s = s + dd[2] * dd[2];
        const float nrm = sqrtf(s);
        r[8] = dd[0] / nrm; r[9] = dd[1] / nrm; r[10] = dd[2] / nrm;
    }
}

// The ray-batch assembly of NeRFRenderer::Render for a pose (NeRFRenderer.h:541-583) in one pass over the pixels of a row tile: GetRays (k_get_rays'
// arithmetic), the view directions d/||d|| taken from c2w's rays BEFORE any c2w_staticcam substitution (:549-561) and BEFORE the NDC warp (:563-568), NDCRays
// (k_ndc_rays' arithmetic), the AABB near/far (aabb_one) and the packed row (k_pack_rays' order).  Also reduces min(near) / max(far) of the tile (:602-603) into
// two order-encoded ints (nf_enc, may be null) so that the caller needs no second pass and no host synchronisation.
struct ViewCam {
    float fx, cx, fy, cy;
    float r[9];      // c2w[:3,:3] row-major
    float t[3];      // c2w[:3,3]
};
struct NdcConst { float sx, sy, near_, two_near, m_two_near; };

__device__ __forceinline__ int order_enc(float f) { const int v = __float_as_int(f); return v >= 0 ? v : (v ^ 0x7fffffff); }

constexpr int VIEW_RPT = 4;          // rays per thread of k_view_rays (256-thread workgroups)
__global__ void __launch_bounds__(256) k_view_rays(int w, int row0, int64_t n, ViewCam cam, ViewCam cam_o, int use_static, int use_viewdirs, int ndc, NdcConst nc, Bbox bb,
                            float *__restrict__ rays, int *__restrict__ nf_enc)
{
    // VIEW_RPT rays per thread, and the tile's min(near) / max(far) reduced per workgroup before it touches the two global words: one same-address atomic per WAVE
    // (20 000 per 800x800 frame, ~10 ns each) was most of this kernel's 190 us
    __shared__ float red_nr[4], red_fr[4];
    float nr_all = INFINITY, fr_all = -INFINITY;
#pragma unroll 1
    for (int rep = 0; rep < VIEW_RPT; rep++) {
    const int64_t i = ((int64_t)blockIdx.x * VIEW_RPT + rep) * blockDim.x + threadIdx.x;
    float nr = INFINITY, fr = -INFINITY;
    if (i < n) {
        const int y = row0 + (int)(i / w);
        const int x = (int)(i % w);
        const float dx = ((float)x - cam.cx) / cam.fx;
        const float dy = -((float)y - cam.cy) / cam.fy;
        const float dz = -1.0f;
        float dd[3], oo[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float v = dx * cam.r[a * 3]; v = v + dy * cam.r[a * 3 + 1]; v = v + dz * cam.r[a * 3 + 2];
            dd[a] = v; oo[a] = cam.t[a];
        }
        const int stride = use_viewdirs ? 11 : 8;
        float *r = rays + i * stride;
        if (use_viewdirs) {
            float s = dd[0] * dd[0]; s = s + dd[1] * dd[1]; s = s + dd[2] * dd[2];
            const float nrm = sqrtf(s);
            r[8] = dd[0] / nrm; r[9] = dd[1] / nrm; r[10] = dd[2] / nrm;
            if (use_static) {                       // :554-558: the camera of the rays is c2w_staticcam, the view directions stay c2w's
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    float v = dx * cam_o.r[a * 3]; v = v + dy * cam_o.r[a * 3 + 1]; v = v + dz * cam_o.r[a * 3 + 2];
                    dd[a] = v; oo[a] = cam_o.t[a];
                }
            }
        }
        if (ndc) {
            float ox = oo[0], oy = oo[1], oz = oo[2];
            const float t = -(nc.near_ + oz) / dd[2];
            ox = ox + t * dd[0]; oy = oy + t * dd[1]; oz = oz + t * dd[2];
            const float q0 = nc.sx * (dd[0] / dd[2] - ox / oz);
            const float q1 = nc.sy * (dd[1] / dd[2] - oy / oz);
            const float q2 = nc.m_two_near / oz;
            oo[0] = nc.sx * ox / oz; oo[1] = nc.sy * oy / oz; oo[2] = 1.0f + nc.two_near / oz;
            dd[0] = q0; dd[1] = q1; dd[2] = q2;
        }
        aabb_one(oo, dd, bb, 0.0f, nr, fr);
        r[0] = oo[0]; r[1] = oo[1]; r[2] = oo[2];
        r[3] = dd[0]; r[4] = dd[1]; r[5] = dd[2];
        r[6] = nr; r[7] = fr;
    }
    nr_all = fminf(nr_all, nr); fr_all = fmaxf(fr_all, fr);
    }
    if (nf_enc) {
        float nr = nr_all, fr = fr_all;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nr = fminf(nr, __shfl_xor(nr, off));
            fr = fmaxf(fr, __shfl_xor(fr, off));
        }
        if ((threadIdx.x & 63) == 0) { red_nr[threadIdx.x >> 6] = nr; red_fr[threadIdx.x >> 6] = fr; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicMin(nf_enc, order_enc(fminf(fminf(red_nr[0], red_nr[1]), fminf(red_nr[2], red_nr[3]))));
            atomicMax(nf_enc + 1, order_enc(fmaxf(fmaxf(red_fr[0], red_fr[1]), fmaxf(red_fr[2], red_fr[3]))));
        }
    }
}

__global__ void k_nf_init(int *nf_enc) { nf_enc[0] = 0x7f800000; nf_enc[1] = (int)(0xff800000u ^ 0x7fffffffu); }
__global__ void k_nf_decode(const int *nf_enc, float *out)
{
    for (int k = 0; k < 2; k++) { const int v = nf_enc[k]; out[k] = __int_as_float(v >= 0 ? v : (v ^ 0x7fffffff)); }
}

// NeRFRenderer.h:549-583 for an explicit ray batch whose view directions come from other directions than the packed ones (Ndc: the pre-warp rays_d)
__global__ void k_pack_rays_vd(int64_t n, Bbox bb, const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ vsrc, float *__restrict__ rays)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float oo[3] = {o[i * 3], o[i * 3 + 1], o[i * 3 + 2]};
    float dd[3] = {d[i * 3], d[i * 3 + 1], d[i * 3 + 2]};
    const float v0 = vsrc[i * 3], v1 = vsrc[i * 3 + 1], v2 = vsrc[i * 3 + 2];
    float nr, fr;
    aabb_one(oo, dd, bb, 0.0f, nr, fr);
    float *r = rays + i * 11;
    r[0] = oo[0]; r[1] = oo[1]; r[2] = oo[2];
    r[3] = dd[0]; r[4] = dd[1]; r[5] = dd[2];
    r[6] = nr; r[7] = fr;
    float s = v0 * v0; s = s + v1 * v1; s = s + v2 * v2;
    const float nrm = sqrtf(s);
    r[8] = v0 / nrm; r[9] = v1 / nrm; r[10] = v2 / nrm;
}

// NeRFRenderer.h:602-603: near.min(), far.max().  Exact (min/max are order independent).
__global__ void k_near_far_range(int64_t n, int stride, const float *__restrict__ rays, float *__restrict__ out /*[2], pre-set to +inf,-inf*/)
{
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        mn = fminf(mn, rays[i * stride + 6]);
        mx = fmaxf(mx, rays[i * stride + 7]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) {
        // floats >= 0 order like their bit patterns; near >= 0 by construction (clamp_min(0)), far may be
        // anything, so use the order-preserving int transform for both.
        auto enc = [](float f) { int v = __float_as_int(f); return v >= 0 ? v : (v ^ 0x7fffffff); };
        atomicMin(reinterpret_cast<int *>(out), enc(mn));
        atomicMax(reinterpret_cast<int *>(out) + 1, enc(mx));
    }
}

__device__ __forceinline__ float safe_inv(float x) { return (fabsf(x) < 1e-8f) ? (1.0f / 1e-8f) : (1.0f / x); }

// NeRFRenderer.h:393-402
__global__ void k_z_vals(int64_t total, int s, int stride, int lindisp, const float *__restrict__ rays, const float *__restrict__ t,
                         float *__restrict__ z)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t ray = i / s;
    const int j = (int)(i - ray * s);
    const float nr = rays[ray * stride + 6], fr = rays[ray * stride + 7];
    const float tj = t[j];
    const float omt = 1.0f - tj;
    float v;
    if (!lindisp) v = nr * omt + fr * tj;
    else v = safe_inv(safe_inv(nr) * omt + safe_inv(fr) * tj);
    z[i] = v;
}

// NeRFRenderer.h:419
__global__ void k_points(int64_t total, int s, int stride, const float *__restrict__ rays, const float *__restrict__ z, float *__restrict__ pts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t ray = i / s;
    const float *r = rays + ray * stride;
    const float zz = z[i];
    pts[i * 3 + 0] = r[0] + r[3] * zz;
    pts[i * 3 + 1] = r[1] + r[4] * zz;
    pts[i * 3 + 2] = r[2] + r[5] * zz;
}

static inline Bbox make_bbox(const float *b)
{
    Bbox bb;
    for (int a = 0; a < 3; a++) { bb.mn[a] = b[a]; bb.mx[a] = b[3 + a]; }
    return bb;
}

// N2: GetRayBatch (NeRFDataset.cpp:109-145): the arithmetic of k_get_rays at arbitrary integer pixel coordinates
__global__ void k_ray_batch(int64_t n, const int64_t *__restrict__ rh, const int64_t *__restrict__ rw, float fx, float cx, float fy, float cy,
                            float r00, float r01, float r02, float r10, float r11, float r12, float r20, float r21, float r22,
                            float t0, float t1, float t2, float *__restrict__ o, float *__restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = ((float)rw[i] - cx) / fx;
    const float dy = -((float)rh[i] - cy) / fy;
    const float dz = -1.0f;
    float a0 = dx * r00; a0 = a0 + dy * r01; a0 = a0 + dz * r02;
    float a1 = dx * r10; a1 = a1 + dy * r11; a1 = a1 + dz * r12;
    float a2 = dx * r20; a2 = a2 + dy * r21; a2 = a2 + dz * r22;
    d[i * 3 + 0] = a0; d[i * 3 + 1] = a1; d[i * 3 + 2] = a2;
    o[i * 3 + 0] = t0; o[i * 3 + 1] = t1; o[i * 3 + 2] = t2;
}

__global__ void k_gather_pixels(int64_t total, int w, int c, const float *__restrict__ img, const int64_t *__restrict__ rh, const int64_t *__restrict__ rw,
                                float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t r = e / c; const int k = (int)(e - r * c);
    out[e] = img[(rh[r] * w + rw[r]) * c + k];
}

__global__ void k_rand_pixels(int64_t n, uint64_t seed, uint64_t idx0, int h0, uint64_t rh_range, int w0, uint64_t rw_range, int64_t *__restrict__ rh,
                              int64_t *__restrict__ rw)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    rh[k] = h0 + (int64_t)(((uint64_t)nrf_rng_u32(seed, 16u, idx0 + (uint64_t)k) * rh_range) >> 32);
    rw[k] = w0 + (int64_t)(((uint64_t)nrf_rng_u32(seed, 17u, idx0 + (uint64_t)k) * rw_range) >> 32);
}

// N4: (depth - near) / span (NeRFExecutor.h:690)
__global__ void k_normalize_depth(int64_t n, float near_, float span, const float *__restrict__ depth, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (depth[i] - near_) / span;
}

// N4: t.mul(255).clamp(0, 255).to(kU8) (NeRFRenderer.h:62); 4 values per thread, one packed 32-bit store when aligned
__global__ void k_to_u8(int64_t n, const float *__restrict__ x, uint8_t *__restrict__ out)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    uint32_t pk = 0;
    const int cnt = (n - i0) < 4 ? (int)(n - i0) : 4;
    for (int k = 0; k < cnt; k++) {
        float v = x[i0 + k] * 255.0f;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        pk |= (uint32_t)(uint8_t)v << (8 * k);
    }
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(out) & 3) == 0)) *reinterpret_cast<uint32_t *>(out + i0) = pk;
    else for (int k = 0; k < cnt; k++) out[i0 + k] = (uint8_t)(pk >> (8 * k));
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_get_rays(int h, int w, const float *K, const float *c2w, int row0, int rows, float *d_o, float *d_d, float *cone_angle, void *stream)
{
    NRF_CHECK_ARG(K && c2w, "nrf_get_rays: null pointer");
    NRF_CHECK_ARG(h > 0 && w > 0 && row0 >= 0 && rows >= 0 && row0 + rows <= h, "nrf_get_rays: rows [%d,%d) outside image of height %d", row0, row0 + rows, h);
    NRF_CHECK_ARG((d_o && d_d) || rows == 0, "nrf_get_rays: null ray buffer");          // an empty tile (a rank of a world larger than the image is high) has no buffer
    const int64_t n = (int64_t)rows * w;
    if (cone_angle) {
        const float px = 1.0f / K[0], py = 1.0f / K[4];
        const float avg = (px + py) / 2.0f;
        *cone_angle = avg * 1.1f;
    }
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_get_rays, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), w, row0, n, K[0], K[2], K[4], K[5],
                       c2w[0], c2w[1], c2w[2], c2w[4], c2w[5], c2w[6], c2w[8], c2w[9], c2w[10], c2w[3], c2w[7], c2w[11], d_o, d_d);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_ndc_rays(int h, int w, float focal, float near_plane, const float *d_o, const float *d_d, int64_t n, float *d_o_out, float *d_d_out, void *stream)
{
    NRF_CHECK_ARG(d_o && d_d && d_o_out && d_d_out && n >= 0, "nrf_ndc_rays: bad argument");
    if (n == 0) return NRF_OK;
    // RayUtils.h:63-71: the double-precision python-style constants are rounded to fp32 when applied to fp32 tensors
    const float sx = (float)(-1. / ((double)w / (2. * (double)focal)));
    const float sy = (float)(-1. / ((double)h / (2. * (double)focal)));
    const float two_near = (float)(2. * (double)near_plane);
    const float m_two_near = (float)(-2. * (double)near_plane);
    hipLaunchKernelGGL(k_ndc_rays, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, sx, sy, near_plane, two_near, m_two_near,
                       d_o, d_d, d_o_out, d_d_out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_aabb(const float *d_o, const float *d_d, const float *bbox, int64_t n, float near_plane, float *d_near, float *d_far, void *stream)
{
    NRF_CHECK_ARG(d_o && d_d && bbox && d_near && d_far && n >= 0, "nrf_aabb: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_aabb, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, make_bbox(bbox), near_plane, d_o, d_d, d_near, d_far);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_pack_rays(const float *d_o, const float *d_d, const float *bbox, int64_t n, int use_viewdirs, float *d_rays, void *stream)
{
    NRF_CHECK_ARG(d_o && d_d && bbox && d_rays && n >= 0, "nrf_pack_rays: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_pack_rays, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, make_bbox(bbox), use_viewdirs, d_o, d_d, d_rays);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_pack_rays_viewsrc(const float *d_o, const float *d_d, const float *d_view_src, const float *bbox, int64_t n, float *d_rays, void *stream)
{
    NRF_CHECK_ARG(d_o && d_d && d_view_src && bbox && d_rays && n >= 0, "nrf_pack_rays_viewsrc: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_pack_rays_vd, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, make_bbox(bbox), d_o, d_d, d_view_src, d_rays);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

static ViewCam make_cam(const float *K, const float *c2w)
{
    ViewCam c;
    c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
    for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) c.r[a * 3 + b] = c2w[a * 4 + b]; c.t[a] = c2w[a * 4 + 3]; }
    return c;
}

int nrf_view_check(const nrf_view *v, const char *who)
{
    NRF_CHECK_ARG(v, "%s: null view", who);
    NRF_CHECK_ARG(v->h > 0 && v->w > 0 && v->row0 >= 0 && v->rows >= 0 && v->row0 + v->rows <= v->h, "%s: rows [%d,%d) outside image of height %d", who, v->row0,
                  v->row0 + v->rows, v->h);
    NRF_CHECK_ARG(v->chunk > 0, "%s: Chunk must be positive", who);
    return NRF_OK;
}

int nrf_view_rays(const nrf_view *v, float *d_rays, float *d_near_far, void *stream)
{
    NRF_TRY(nrf_view_check(v, "nrf_view_rays"));
    NRF_CHECK_ARG(d_rays || v->rows == 0, "nrf_view_rays: null ray buffer");
    const int64_t n = (int64_t)v->rows * v->w;
    hipStream_t st = as_stream(stream);
    if (n == 0) {                                  // an empty tile (h < world): Near / Far of no rays = (+inf, -inf), the identities of min / max
        if (d_near_far) {
            hipLaunchKernelGGL(k_nf_init, dim3(1), dim3(1), 0, st, reinterpret_cast<int *>(d_near_far)); NRF_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_nf_decode, dim3(1), dim3(1), 0, st, reinterpret_cast<const int *>(d_near_far), d_near_far); NRF_LAUNCH_CHECK();
        }
        return NRF_OK;
    }
    NdcConst nc{};
    if (v->ndc) {
        // RayUtils.h:63-71 with focal = k[0][0], near = 1.f (NeRFRenderer.h:567)
        const double focal = (double)v->K[0];
        nc.sx = (float)(-1. / ((double)v->w / (2. * focal)));
        nc.sy = (float)(-1. / ((double)v->h / (2. * focal)));
        nc.near_ = 1.0f; nc.two_near = 2.0f; nc.m_two_near = -2.0f;
    }
    int *nf_enc = d_near_far ? reinterpret_cast<int *>(d_near_far) : nullptr;     // reduced in place as order-encoded ints, decoded to floats at the end
    if (nf_enc) { hipLaunchKernelGGL(k_nf_init, dim3(1), dim3(1), 0, st, nf_enc); NRF_LAUNCH_CHECK(); }
    const ViewCam cam = make_cam(v->K, v->c2w);
    const ViewCam cam_o = v->has_staticcam ? make_cam(v->K, v->c2w_staticcam) : cam;
    hipLaunchKernelGGL(k_view_rays, dim3((unsigned)ceil_div(n, (int64_t)256 * VIEW_RPT)), dim3(256), 0, st, v->w, v->row0, n, cam, cam_o, v->has_staticcam ? 1 : 0,
                       v->use_viewdirs ? 1 : 0, v->ndc ? 1 : 0, nc, make_bbox(v->bbox), d_rays, nf_enc);
    NRF_LAUNCH_CHECK();
    if (nf_enc) { hipLaunchKernelGGL(k_nf_decode, dim3(1), dim3(1), 0, st, nf_enc, d_near_far); NRF_LAUNCH_CHECK(); }
    return NRF_OK;
}

int nrf_near_far_range(const float *d_rays, int64_t n, int ray_stride, float *near_min, float *far_max, void *stream)
{
    NRF_CHECK_ARG(d_rays && n > 0 && ray_stride >= 8 && near_min && far_max, "nrf_near_far_range: bad argument");
    int *d_out = nullptr;
    NRF_HIP(scratch_take(reinterpret_cast<void **>(&d_out), 2 * sizeof(int), as_stream(stream)));
    const int init[2] = {0x7f800000 /* +inf */, (int)(0xff800000u ^ 0x7fffffffu) /* enc(-inf) */};
    NRF_HIP(hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, as_stream(stream)));
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024);
    hipLaunchKernelGGL(k_near_far_range, dim3(grid), dim3(256), 0, as_stream(stream), n, ray_stride, d_rays, reinterpret_cast<float *>(d_out));
    NRF_LAUNCH_CHECK();
    int res[2];
    NRF_HIP(hipMemcpyAsync(res, d_out, sizeof(res), hipMemcpyDeviceToHost, as_stream(stream)));
    NRF_HIP(hipStreamSynchronize(as_stream(stream)));
    NRF_HIP(scratch_give(d_out, as_stream(stream)));
    auto dec = [](int v) { int b = v >= 0 ? v : (v ^ 0x7fffffff); float f; memcpy(&f, &b, 4); return f; };
    *near_min = dec(res[0]);
    *far_max = dec(res[1]);
    return NRF_OK;
}

int nrf_near_far_range_device(const float *d_rays, int64_t n, int ray_stride, float *d_near_far, void *stream)
{
    NRF_CHECK_ARG(d_rays && n > 0 && ray_stride >= 8 && d_near_far, "nrf_near_far_range_device: bad argument");
    hipStream_t st = as_stream(stream);
    int *enc = reinterpret_cast<int *>(d_near_far);          // the two order-encoded words live in the output itself until the decode (as nrf_view_rays)
    hipLaunchKernelGGL(k_nf_init, dim3(1), dim3(1), 0, st, enc); NRF_LAUNCH_CHECK();
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024);
    hipLaunchKernelGGL(k_near_far_range, dim3(grid), dim3(256), 0, st, n, ray_stride, d_rays, d_near_far);
    NRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nf_decode, dim3(1), dim3(1), 0, st, reinterpret_cast<const int *>(enc), d_near_far); NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_precrop_bounds(int h, int w, int iter, int precrop_iters, float precrop_frac, int *out)
{
    NRF_CHECK_ARG(out && h > 0 && w > 0, "nrf_precrop_bounds: bad argument");
    if (iter < precrop_iters) {
        const int dh = (int)(h / 2 * precrop_frac), dw = (int)(w / 2 * precrop_frac);
        out[0] = h / 2 - dh; out[1] = h / 2 + dh - 1; out[2] = w / 2 - dw; out[3] = w / 2 + dw - 1;
    } else { out[0] = 0; out[1] = h - 1; out[2] = 0; out[3] = w - 1; }
    return NRF_OK;
}

// the render-factor step of NeRFExecutor::RenderView (NeRFExecutor.h:618-627): h, w and K's fx, fy, cx, cy divided by the (float) factor
int nrf_render_view_dims(int h, int w, const float *K, float render_factor, int *h1, int *w1, float *K1)
{
    NRF_CHECK_ARG(h1 && w1 && h >= 0 && w >= 0 && render_factor >= 0.0f, "nrf_render_view_dims: bad argument");
    NRF_CHECK_ARG(K || !K1, "nrf_render_view_dims: K1 wanted but K is NULL");
    if (K1) for (int i = 0; i < 9; i++) K1[i] = K[i];
    *h1 = h; *w1 = w;
    if (render_factor != 0.0f) {
        *h1 = (int)((float)h / render_factor);              // int = int / float, as `h = h / rparams.RenderFactor`
        *w1 = (int)((float)w / render_factor);
        if (K1) { K1[0] = K[0] / render_factor; K1[4] = K[4] / render_factor; K1[2] = K[2] / render_factor; K1[5] = K[5] / render_factor; }
    }
    return NRF_OK;
}

int nrf_rand_pixels(uint64_t seed, int64_t iter, int h_start, int h_end, int w_start, int w_end, int64_t n, int64_t *d_rand_h, int64_t *d_rand_w, void *stream)
{
    NRF_CHECK_ARG(d_rand_h && d_rand_w && n >= 0 && iter >= 0 && h_end >= h_start && w_end >= w_start, "nrf_rand_pixels: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_rand_pixels, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, seed, (uint64_t)iter * (uint64_t)n, h_start,
                       (uint64_t)(h_end - h_start + 1), w_start, (uint64_t)(w_end - w_start + 1), d_rand_h, d_rand_w);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_ray_batch(const float *K, const float *c2w, const int64_t *d_rand_h, const int64_t *d_rand_w, int64_t n, float *d_o, float *d_d, float *cone_angle,
                  void *stream)
{
    NRF_CHECK_ARG(K && c2w && d_rand_h && d_rand_w && d_o && d_d && n >= 0, "nrf_ray_batch: bad argument");
    if (cone_angle) { const float px = 1.0f / K[0], py = 1.0f / K[4]; *cone_angle = (px + py) / 2.0f; }
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_ray_batch, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, d_rand_h, d_rand_w, K[0], K[2], K[4], K[5],
                       c2w[0], c2w[1], c2w[2], c2w[4], c2w[5], c2w[6], c2w[8], c2w[9], c2w[10], c2w[3], c2w[7], c2w[11], d_o, d_d);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_gather_pixels(const float *d_image, int h, int w, int c, const int64_t *d_rand_h, const int64_t *d_rand_w, int64_t n, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_image && d_rand_h && d_rand_w && d_out && h > 0 && w > 0 && c > 0 && n >= 0, "nrf_gather_pixels: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_gather_pixels, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, as_stream(stream), n * c, w, c, d_image, d_rand_h, d_rand_w, d_out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_normalize_depth(const float *d_depth, int64_t n, float near_, float far_, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_depth && d_out && n >= 0, "nrf_normalize_depth: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_normalize_depth, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, near_, far_ - near_, d_depth, d_out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_to_u8(const float *d_x, int64_t n, uint8_t *d_out, void *stream)
{
    NRF_CHECK_ARG(d_x && d_out && n >= 0, "nrf_to_u8: bad argument");
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_to_u8, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, as_stream(stream), n, d_x, d_out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_linspace(float start, float end, int steps, float *out)
{
    NRF_CHECK_ARG(out && steps >= 1, "nrf_linspace: bad argument");
    if (steps == 1) { out[0] = start; return NRF_OK; }
    // ATen linspace_kernel: fp32 step, halves mirrored, one FUSED rounding per element (see oracle/nerf_oracle.c orc_linspace)
    const float step = (end - start) / (float)(steps - 1);
    const int halfway = steps / 2;
    for (int i = 0; i < steps; i++)
        out[i] = (i < halfway) ? __builtin_fmaf(step, (float)i, start) : __builtin_fmaf(-step, (float)(steps - i - 1), end);
    return NRF_OK;
}

int nrf_z_vals(const float *d_rays, int ray_stride, int64_t n, const float *d_t, int s, int lindisp, float *d_z, void *stream)
{
    NRF_CHECK_ARG(d_rays && d_t && d_z && ray_stride >= 8 && n >= 0 && s > 0, "nrf_z_vals: bad argument");
    const int64_t total = n * s;
    if (total == 0) return NRF_OK;
    hipLaunchKernelGGL(k_z_vals, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), total, s, ray_stride, lindisp, d_rays, d_t, d_z);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_points(const float *d_rays, int ray_stride, const float *d_z, int64_t n, int s, float *d_pts, void *stream)
{
    NRF_CHECK_ARG(d_rays && d_z && d_pts && ray_stride >= 8 && n >= 0 && s > 0, "nrf_points: bad argument");
    const int64_t total = n * s;
    if (total == 0) return NRF_OK;
    hipLaunchKernelGGL(k_points, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), total, s, ray_stride, d_rays, d_z, d_pts);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // extern "C"
