// stoch.hip -- the stochastic branches of RenderRays (NeRFRenderer.h:366-459):
//   stratified jitter            :404-417   (Perturb > 0)
//   TangentScatter               :307-362   (cone rays, i.e. ThinRay = false)
//   stochastic preconditioning   :433-443 + ReflectBoundary :285-304
// Every kernel takes its random draws either as explicit arrays (the reference's torch::rand / randn tensors -- that is how the
// tests compare value for value) or, with a NULL array, generates them from (seed, stream, global element index) with
// include/nrf_rng.h: no draw buffer is written, and a render is independent of Chunk and of the ray sharding.
#include "common.h"
#include "stoch.h"

namespace nrf {

__device__ __forceinline__ float draw_u(const float *arr, int64_t local, const RngRef &g, uint32_t stream, uint64_t global)
{
    return arr ? arr[local] : nrf_rng_uniform(g.seed, stream, global);
}

__device__ __forceinline__ float draw_n(const float *arr, int64_t local, const RngRef &g, uint32_t stream, uint64_t global)
{
    return arr ? arr[local] : nrf_rng_normal(g.seed, stream, global);
}

// one thread per depth; neighbours re-read from global (L1/L2 hits)
__global__ void k_jitter_z(int64_t total, int s, const float *__restrict__ z, const float *__restrict__ t_rand, RngRef g, float *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int64_t ray = (q >> 31) == 0 ? (int64_t)((uint32_t)q / (uint32_t)s) : q / s;
    const int k = (int)(q - ray * s);
    const float *zi = z + ray * s;
    const float upper = (k < s - 1) ? 0.5f * (zi[k + 1] + zi[k]) : zi[s - 1];
    const float lower = (k > 0) ? 0.5f * (zi[k] + zi[k - 1]) : zi[0];
    const float interval = upper - lower;
    const float t = draw_u(t_rand, q, g, NRF_RNG_T_RAND, (uint64_t)(g.ray_base * s + q));
    out[q] = lower + ((interval > 1e-8f) ? interval * t : 0.0f);
}

__device__ __forceinline__ void normalize3(float (&v)[3])
{
    float nrm = __builtin_sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    nrm = fmaxf(nrm, 1e-8f);
    v[0] = v[0] / nrm; v[1] = v[1] / nrm; v[2] = v[2] / nrm;
}

__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&o)[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// pts (explicit, or o + d*z when NULL)  ->  [+ noise*alpha, reflected into the box]  ->  [+ in-cone tangent-plane offset, clamped]
__global__ void k_stoch_points(int64_t total, int s, const float *__restrict__ pts_in, const float *__restrict__ rays, int ray_stride,
                               const float *__restrict__ z, StochPoints sp, RngRef g, float *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int64_t ray = (q >> 31) == 0 ? (int64_t)((uint32_t)q / (uint32_t)s) : q / s;
    const float *rp = rays + ray * ray_stride;
    const float zz = z[q];
    const uint64_t gq = (uint64_t)(g.ray_base * s + q);
    float p[3];
    if (pts_in) { p[0] = pts_in[q * 3]; p[1] = pts_in[q * 3 + 1]; p[2] = pts_in[q * 3 + 2]; }
    else { p[0] = rp[0] + rp[3] * zz; p[1] = rp[1] + rp[4] * zz; p[2] = rp[2] + rp[5] * zz; }
    if (sp.precond) {                                                                   // :433-443, :285-304
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float ext = sp.box.mx[a] - sp.box.mn[a];
            const float v = p[a] + draw_n(sp.noise, q * 3 + a, g, sp.stream_noise, gq * 3 + a) * sp.alpha;
            float x = (v - sp.box.mn[a]) / ext;
            x = fmodf(x, 2.0f);
            if (x > 1.0f) x = 2.0f - x;
            p[a] = x * ext + sp.box.mn[a];
        }
    }
    if (sp.cone) {                                                                      // :307-362
        float dn[3] = {rp[3], rp[4], rp[5]};
        normalize3(dn);
        const float ax = fabsf(dn[0]), ay = fabsf(dn[1]), az = fabsf(dn[2]);
        const bool mx = (ax < ay) && (ax < az), my = (ay < ax) && (ay < az);
        const float up[3] = {mx ? 1.0f : 0.0f, (!mx && my) ? 1.0f : 0.0f, (!mx && !my) ? 1.0f : 0.0f};
        float tg[3], bt[3];
        cross3(dn, up, tg); normalize3(tg);
        cross3(dn, tg, bt); normalize3(bt);
        float u1 = draw_u(sp.u_r, q, g, sp.stream_r, gq);
        u1 = fminf(fmaxf(u1, 1e-8f), 1.0f - 1e-8f);
        const float r = __builtin_sqrtf(u1);
        const float theta = fmodf(draw_u(sp.u_theta, q, g, sp.stream_theta, gq) * 2.0f * 3.14159274f, 6.2831855f);
        float sn, cs;
        nrf_sincosf(theta, &sn, &cs);
        const float ox = r * cs, oy = r * sn;
        const float radius = sp.cone_angle * zz;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float v = p[a] + (tg[a] * ox + bt[a] * oy) * radius;
            if (sp.clamp) v = fminf(fmaxf(v, sp.box.mn[a]), sp.box.mx[a]);
            p[a] = v;
        }
    }
    out[q * 3] = p[0]; out[q * 3 + 1] = p[1]; out[q * 3 + 2] = p[2];
}

__global__ void k_rng_fill(int64_t count, uint64_t seed, uint32_t stream, uint64_t idx0, int normal, float *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = normal ? nrf_rng_normal(seed, stream, idx0 + (uint64_t)k) : nrf_rng_uniform(seed, stream, idx0 + (uint64_t)k);
}

int launch_jitter_z(const float *z, const float *t_rand, const RngRef &g, int64_t n, int s, float *out, hipStream_t st)
{
    const int64_t total = n * s;
    if (total == 0) return NRF_OK;
    hipLaunchKernelGGL(k_jitter_z, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, total, s, z, t_rand, g, out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int launch_stoch_points(const float *pts_in, const float *rays, int ray_stride, const float *z, int64_t n, int s, const StochPoints &sp,
                        const RngRef &g, float *out, hipStream_t st)
{
    const int64_t total = n * s;
    if (total == 0) return NRF_OK;
    hipLaunchKernelGGL(k_stoch_points, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, total, s, pts_in, rays, ray_stride, z, sp, g, out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace nrf

using namespace nrf;

extern "C" {

int nrf_rng_fill(uint64_t seed, uint32_t rng_stream, uint64_t index0, int64_t count, int normal, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_out && count >= 0, "nrf_rng_fill: bad argument");
    if (count == 0) return NRF_OK;
    hipLaunchKernelGGL(k_rng_fill, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, as_stream(stream), count, seed, rng_stream, index0, normal, d_out);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_jitter_z(const float *d_z, const float *d_t_rand, int64_t n, int s, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_z && d_t_rand && d_out && d_out != d_z && n >= 0 && s >= 2, "nrf_jitter_z: bad argument (out must not alias z)");
    return launch_jitter_z(d_z, d_t_rand, RngRef{0, 0}, n, s, d_out, as_stream(stream));
}

int nrf_tangent_scatter(const float *d_pts, const float *d_rays, int ray_stride, const float *d_z, int64_t n, int s, float cone_angle,
                        const float *d_u_r, const float *d_u_theta, const float *bbox, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_rays && d_z && d_u_r && d_u_theta && d_out && ray_stride >= 6 && n >= 0 && s >= 1, "nrf_tangent_scatter: bad argument");
    StochPoints sp{};
    sp.cone = 1; sp.cone_angle = cone_angle; sp.u_r = d_u_r; sp.u_theta = d_u_theta;
    if (bbox) { sp.clamp = 1; for (int a = 0; a < 3; a++) { sp.box.mn[a] = bbox[a]; sp.box.mx[a] = bbox[3 + a]; } }
    return launch_stoch_points(d_pts, d_rays, ray_stride, d_z, n, s, sp, RngRef{0, 0}, d_out, as_stream(stream));
}

int nrf_precondition(const float *d_pts, const float *d_noise, float alpha, const float *bbox, int64_t p, float *d_out, void *stream)
{
    NRF_CHECK_ARG(d_pts && d_noise && bbox && d_out && p >= 0, "nrf_precondition: bad argument");
    StochPoints sp{};
    sp.precond = 1; sp.alpha = alpha; sp.noise = d_noise;
    for (int a = 0; a < 3; a++) { sp.box.mn[a] = bbox[a]; sp.box.mx[a] = bbox[3 + a]; }
    // one "ray" per point: rays/z are not read beyond the (unused) o + d*z operands, so pass the points themselves
    return launch_stoch_points(d_pts, d_pts, 3, d_pts, p, 1, sp, RngRef{0, 0}, d_out, as_stream(stream));
}

}  // extern "C"
