// stoch.h -- internal interface of stoch.hip (stochastic branches of RenderRays).
#pragma once
#include "common.h"
#include "nrf_rng.h"

namespace nrf {

struct RngRef {
    uint64_t seed;
    int64_t ray_base;      // index of the chunk's first ray in the whole image: draws are addressed by GLOBAL element index
};

struct StochPoints {
    int precond = 0;                       // stochastic preconditioning + ReflectBoundary
    float alpha = 0.0f;
    const float *noise = nullptr;          // [p,3] normal draws, or NULL -> nrf_rng_normal(stream_noise)
    uint32_t stream_noise = NRF_RNG_PRECOND;
    int cone = 0;                          // TangentScatter
    float cone_angle = 0.0f;
    const float *u_r = nullptr, *u_theta = nullptr;    // [p] uniform draws, or NULL -> nrf_rng_uniform(stream_r / stream_theta)
    uint32_t stream_r = NRF_RNG_R_COARSE, stream_theta = NRF_RNG_THETA_COARSE;
    int clamp = 0;
    Bbox box = {{0, 0, 0}, {0, 0, 0}};
};

struct SigmaNoise {                        // RawNoiseStd > 0: sigma + normal*std before the relu (NeRFRenderer.h:251-252)
    int on = 0;
    const float *arr = nullptr;            // [n,s] normal draws, or NULL -> nrf_rng_normal(stream)
    float std = 0.0f;
    RngRef g = {0, 0};
    uint32_t stream = NRF_RNG_NOISE_COARSE;
};

int launch_raw2outputs(const float *raw, const float *z, const float *dirs, int d_stride, int64_t n, int s, int c, int sigma_ch, int white, float *rgb,
                       float *disp, float *acc, float *weights, float *depth, const SigmaNoise &nz, hipStream_t st, bool fast = false, const int32_t *src = nullptr,
                       const float *raw2 = nullptr, int64_t n_split = 0, uint32_t *flag = nullptr);
int launch_clip_embedding(const float *embeds, int embed_stride, int embed_dim, const float *weights, int64_t n, int s, float *out, hipStream_t st, uint32_t *flag = nullptr);
int launch_fine_depths(const float *z, const float *weights, int64_t n, int s, const float *u, int64_t u_stride, const RngRef &g, int ns, int sum_vec,
                       float *zf, hipStream_t st, int32_t *src = nullptr, float *z_new = nullptr);
int launch_jitter_z(const float *z, const float *t_rand, const RngRef &g, int64_t n, int s, float *out, hipStream_t st);
int launch_stoch_points(const float *pts_in, const float *rays, int ray_stride, const float *z, int64_t n, int s, const StochPoints &sp,
                        const RngRef &g, float *out, hipStream_t st);

}  // namespace nrf
