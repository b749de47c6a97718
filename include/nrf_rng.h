/* nrf_rng.h -- counter-based random numbers for the stochastic render branches (Perturb > 0, cone rays /
 * TangentScatter, RawNoiseStd, stochastic preconditioning: NeRFRenderer.h:342-343, :415, :252, :439; Sampler.h:23).
 *
 * The reference draws from torch's global Mersenne-Twister / Philox stream, which makes a render depend on call
 * order, chunk size and device.  Here every draw is a pure function of (seed, stream, element index), so a
 * render is reproducible, independent of Chunk and of how rays are sharded over GPUs, and the CPU oracle and the
 * HIP kernels produce the same numbers.  Shared by oracle/nerf_oracle.c and nerfpp_amd/csrc (like nrf_math.h).
 *
 *   u32     = high word of splitmix64(seed + GOLDEN*idx  ^  stream*K)
 *   uniform = (u32 >> 8) * 2^-24            in [0,1), the same 24-bit grid as torch's CPU float uniform
 *   normal  = Box-Muller on draws 2*idx, 2*idx+1 (cosine branch)
 */
#ifndef NRF_RNG_H
#define NRF_RNG_H
#include <stdint.h>
#include "nrf_math.h"

enum {
    NRF_RNG_T_RAND = 1,        /* stratified jitter            [n, S]          */
    NRF_RNG_R_COARSE = 2,      /* TangentScatter radius        [n, S]          */
    NRF_RNG_THETA_COARSE = 3,  /* TangentScatter angle         [n, S]          */
    NRF_RNG_NOISE_COARSE = 4,  /* raw_noise_std randn          [n, S]          */
    NRF_RNG_U_PDF = 5,         /* SamplePDF u (det = false)    [n, N_importance] */
    NRF_RNG_PRECOND = 6,       /* stochastic preconditioning   [n, S+Ni, 3]    */
    NRF_RNG_R_FINE = 7,
    NRF_RNG_THETA_FINE = 8,
    NRF_RNG_NOISE_FINE = 9
};

NRF_HD uint32_t nrf_rng_u32(uint64_t seed, uint32_t stream, uint64_t idx)
{
    uint64_t x = seed + idx * 0x9E3779B97F4A7C15ull;
    x ^= (uint64_t)stream * 0xD1B54A32D192ED03ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}

NRF_HD float nrf_rng_uniform(uint64_t seed, uint32_t stream, uint64_t idx)
{
    return (float)(nrf_rng_u32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

NRF_HD float nrf_rng_normal(uint64_t seed, uint32_t stream, uint64_t idx)
{
    const float u1 = (float)((nrf_rng_u32(seed, stream, 2 * idx) >> 8) + 1u) * (1.0f / 16777216.0f);   /* (0,1] */
    const float u2 = nrf_rng_uniform(seed, stream, 2 * idx + 1);
    const float r = __builtin_sqrtf(-2.0f * nrf_logf(u1));
    return r * nrf_cosf(6.2831855f * u2);
}

#endif /* NRF_RNG_H */
