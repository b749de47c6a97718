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
int launch_dirs_f16(const float *rays, int stride, int64_t n, int degree, int variant, __half *out, __half *out_lo, hipStream_t st);

constexpr int HASH_LM_DEFAULT_VARIANT = 0;

}  // namespace nrf
