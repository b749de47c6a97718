/*
 * nrf_synth.h -- deterministic synthetic-parameter generator shared by the bench,
 * the tests, the CPU oracle and the reference-driver that emits golden vectors.
 *
 * There is no dataset, checkpoint or network in the build environment, so every
 * weight / hash-table entry used for parity and benchmarking is DEFINED by this
 * closed form: value(seed, i) = amp * (2 * u01(seed, i) - 1).  The same formula is
 * restated in numpy in nerfpp_amd/synth.py; tests check both agree bit for bit.
 *
 * Plain C99, no dependencies.
 */
#ifndef NRF_SYNTH_H
#define NRF_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lowbias32 integer finaliser over a Weyl-sequence index. */
static inline uint32_t nrf_synth_u32(uint32_t seed, uint32_t i)
{
    uint32_t x = i * 0x9E3779B9u + seed;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

/* 24-bit uniform in [0, 1): exactly representable in fp32. */
static inline float nrf_synth_u01(uint32_t seed, uint32_t i)
{
    return (float)(nrf_synth_u32(seed, i) >> 8) * (1.0f / 16777216.0f);
}

/* Symmetric uniform in [-amp, amp): one fp32 rounding (the final multiply). */
static inline float nrf_synth_sym(uint32_t seed, uint32_t i, float amp)
{
    return amp * (2.0f * nrf_synth_u01(seed, i) - 1.0f);
}

#ifdef __cplusplus
}
#endif
#endif /* NRF_SYNTH_H */
