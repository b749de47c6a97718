/*
 * nrf_math.h -- portable fp32 exp / log / sin / cos built ONLY from IEEE-754 correctly rounded operations
 * (fused multiply-add, multiply, add, divide, int<->float conversion, bit casts).
 *
 * Why: libm (glibc), SLEEF (what ATen's CPU kernels call) and the GPU's OCML all differ in the last ulp.  The
 * hierarchical sampler turns ulp-level differences of the coarse weights into different sample INDICES wherever the
 * CDF has a plateau, so a renderer that wants bit-reproducible sample indices across CPU and GPU needs one definition
 * of these four functions.  This header is that definition: compiled by hipcc for gfx950 (device) and by gcc for the
 * CPU oracle, it returns the same bits on both for every input.  Accuracy: <= 1.5 ulp on the ranges the renderer uses
 * (exp: all x; log: x > 0 normal; sin/cos: |x| < 1e4), i.e. within the spread of the three libraries above.
 *
 * C99 / C++ / HIP.  fmaf() must be a real fused operation (gcc: build with -mfma; HIP: always).
 */
#ifndef NRF_MATH_H
#define NRF_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define NRF_HD __host__ __device__ static inline
#else
#define NRF_HD static inline
#endif

NRF_HD float nrf_bits_to_f32(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
NRF_HD uint32_t nrf_f32_to_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
NRF_HD float nrf_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* 2^k as fp32 for k in [-126, 127] */
NRF_HD float nrf_pow2i(int k) { return nrf_bits_to_f32((uint32_t)(k + 127) << 23); }

/* exp(x): k = round(x*log2e); r = x - k*ln2 (two-constant Cody-Waite); degree-7 Taylor on |r| <= ln2/2; scale by 2^k in two
 * exact steps so that results in the subnormal range round once. */
NRF_HD float nrf_expf(float x)
{
    if (x != x) return x;
    if (x > 88.7228394f) return nrf_bits_to_f32(0x7f800000u);
    if (x < -103.972084f) return 0.0f;
    const float kf = __builtin_floorf(nrf_fma(x, 1.44269504088896341f, 0.5f));
    float r = nrf_fma(-kf, 0.693145751953125f, x);          /* ln2 high part: 0x3f317200 (trailing zeros) */
    r = nrf_fma(-kf, 1.42860682030941723e-6f, r);            /* ln2 low part */
    float p = 1.0f / 5040.0f;
    p = nrf_fma(p, r, 1.0f / 720.0f);
    p = nrf_fma(p, r, 1.0f / 120.0f);
    p = nrf_fma(p, r, 1.0f / 24.0f);
    p = nrf_fma(p, r, 1.0f / 6.0f);
    p = nrf_fma(p, r, 0.5f);
    const float r2 = r * r;
    p = nrf_fma(p, r2, r);
    p = p + 1.0f;
    const int k = (int)kf;
    const int k1 = k / 2, k2 = k - k1;                       /* |k| <= 150 -> both halves are normal powers of two */
    return (p * nrf_pow2i(k1)) * nrf_pow2i(k2);
}

/* log(x) for x > 0 (x <= 0 and NaN follow IEEE conventions): x = m * 2^e, m in [sqrt(1/2), sqrt(2));
 * log m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^9; result = e*ln2_hi + (log m + e*ln2_lo). */
NRF_HD float nrf_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return nrf_bits_to_f32(0x7fc00000u);
    if (x == 0.0f) return nrf_bits_to_f32(0xff800000u);
    uint32_t ux = nrf_f32_to_bits(x);
    if (ux == 0x7f800000u) return x;
    int e = 0;
    if (ux < 0x00800000u) { x = x * 8388608.0f; ux = nrf_f32_to_bits(x); e = -23; }     /* subnormal */
    e += (int)(ux >> 23) - 127;
    uint32_t um = (ux & 0x007fffffu) | 0x3f800000u;          /* m in [1, 2) */
    if (um >= 0x3fb504f3u) { um -= 0x00800000u; e += 1; }    /* m >= sqrt(2): halve */
    const float m = nrf_bits_to_f32(um);
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 2.0f / 9.0f;
    p = nrf_fma(p, z, 2.0f / 7.0f);
    p = nrf_fma(p, z, 2.0f / 5.0f);
    p = nrf_fma(p, z, 2.0f / 3.0f);
    const float lm = nrf_fma(p * z, s, 2.0f * s);            /* 2s + s*z*p */
    const float ef = (float)e;
    return nrf_fma(ef, 0.693145751953125f, nrf_fma(ef, 1.42860682030941723e-6f, lm));
}

/* sin/cos: k = round(x * 2/pi); r = x - k*pi/2 with a three-constant Cody-Waite split (exact for |k| < 2^15);
 * cephes single-precision minimax polynomials on |r| <= pi/4. */
NRF_HD void nrf_sincosf(float x, float *sn, float *cs)
{
    const float kf = __builtin_floorf(nrf_fma(x, 0.636619772367581343f, 0.5f));
    float r = nrf_fma(-kf, 1.5703125f, x);
    r = nrf_fma(-kf, 4.837512969970703125e-4f, r);
    r = nrf_fma(-kf, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = nrf_fma(ps, z, 8.3321608736e-3f);
    ps = nrf_fma(ps, z, -1.6666654611e-1f);
    const float s = nrf_fma(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = nrf_fma(pc, z, -1.388731625493765e-3f);
    pc = nrf_fma(pc, z, 4.166664568298827e-2f);
    const float c = nrf_fma(pc * z, z, nrf_fma(-0.5f, z, 1.0f));
    const int q = (int)kf & 3;
    *sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

NRF_HD float nrf_sinf(float x) { float s, c; nrf_sincosf(x, &s, &c); return s; }
NRF_HD float nrf_cosf(float x) { float s, c; nrf_sincosf(x, &s, &c); return c; }

/* torch::sigmoid */
NRF_HD float nrf_sigmoidf(float x) { return 1.0f / (1.0f + nrf_expf(-x)); }

#endif /* NRF_MATH_H */
