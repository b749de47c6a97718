// train.hip -- N1: the pieces of one optimisation step of NeRFExecutor::Train (NeRFExecutor.h:862-995) that are not the MLP:
//   huber_loss / mse_loss                       :882-887          nrf_huber_loss
//   RawToOutputs backward (+ TruncExp::backward) NeRFRenderer.h:199-282, CustomOps.cpp:11-15   nrf_raw2outputs_backward
//   hash-grid backward                          HashEmbedder: nn::Embedding index_add of the trilinear weights (NeRF.cpp:279-298)
//                                               CuHashEmbedder: CuHashEmbedder.cu:105-216, host :277-325   nrf_hash_backward
//   Adam(lr, betas (0.9, 0.99), eps 1e-15)      :539               nrf_adam_step
// Gradients flow only through the fine pass (z_samples are detached, NeRFRenderer.h:429).
#include "encode.h"

namespace nrf {

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// out[0] += sum huber, out[1] += sum squared error (both divided by count on the host side of the call)
__global__ void k_huber(int64_t count, float norm, const float *__restrict__ pred, const float *__restrict__ target, double *__restrict__ out, float *__restrict__ grad)
{
    double h = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        const float z = fabsf(d);
        h += (z < 1.0f) ? 0.5 * (double)z * (double)z : (double)z - 0.5;
        q += (double)d * (double)d;
        if (grad) grad[i] = (d < -1.0f) ? -norm : (d > 1.0f ? norm : norm * d);
    }
    h = wsum(h); q = wsum(q);
    if ((threadIdx.x & 63) == 0) { unsafeAtomicAdd(out, h); unsafeAtomicAdd(out + 1, q); }
}

__global__ void k_finish_loss(int64_t count, const double *__restrict__ acc, float *__restrict__ loss_mse)
{
    loss_mse[0] = (float)(acc[0] / (double)count);
    loss_mse[1] = (float)(acc[1] / (double)count);
}

// One wave per ray.  Forward quantities are recomputed exactly as k_raw2outputs does (same double scan for the log-transmittance);
// the reverse pass needs suffix sums of g_L over later samples: a reverse wave scan per 64-sample block, blocks walked last to first.
constexpr int BW_RAYS = 4;
__device__ __forceinline__ double wave_incl_scan_d(double v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ void __launch_bounds__(64 * BW_RAYS)
k_raw2outputs_bwd(int64_t n, int s, int c, int white, const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dirs, int d_stride,
                  const float *__restrict__ g_rgb, float *__restrict__ g_raw, float *__restrict__ lt_scratch /* [n, s] exclusive log-transmittance */,
                  const float *__restrict__ noise, float noise_std)
{
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * BW_RAYS + (threadIdx.x >> 6);
    if (ray >= n) return;
    const float *dv = dirs + ray * d_stride;
    float nn = dv[0] * dv[0]; nn = nn + dv[1] * dv[1]; nn = nn + dv[2] * dv[2];
    const float nrm = sqrtf(nn);
    const float *zr = z + ray * s;
    float *lt = lt_scratch + ray * s;
    // pass 1 (forward order): exclusive prefix of log(clamp_min(1 - alpha, 1e-10)), rounded to fp32 like torch::cumsum's output
    double carry = 0.0;
    for (int base = 0; base < s; base += 64) {
        const int j = base + lane;
        float lg = 0.0f;
        if (j < s) {
            const float *r = raw + (ray * s + j) * c;
            float dist = (j + 1 < s) ? (zr[j + 1] - zr[j]) : 1e10f;
            dist = dist * nrm;
            float sr = r[3];
            if (noise) sr = sr + noise[ray * s + j] * noise_std;     // the forward's sigma + randn * raw_noise_std (NeRFRenderer.h:251-252)
            const float sig = sr > 0.0f ? sr : 0.0f;
            const float alpha = -nrf_expf(-sig * dist) + 1.0f;
            const float om = 1.0f - alpha;
            lg = nrf_logf(om > 1e-10f ? om : 1e-10f);
        }
        const double incl = wave_incl_scan_d((double)lg, lane);
        if (j < s) lt[j] = (float)(carry + (incl - (double)lg));
        carry += __shfl(incl, 63);
    }
    wave_sync();
    // pass 2 (reverse order)
    const float gr = g_rgb[ray * 3], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
    const float gsum = gr + gg + gb;
    double suffix = 0.0;                                   // sum of g_L over samples in later blocks
    const int nblk = (s + 63) / 64;
    for (int blk = nblk - 1; blk >= 0; blk--) {
        const int j = blk * 64 + lane;
        const bool live = j < s;
        float gL = 0.0f, gw = 0.0f, alpha = 0.0f, trans = 0.0f, x = 0.0f, dist = 0.0f, sraw = 0.0f;
        float col[3] = {0.0f, 0.0f, 0.0f};
        if (live) {
            const float *r = raw + (ray * s + j) * c;
            dist = (j + 1 < s) ? (zr[j + 1] - zr[j]) : 1e10f;
            dist = dist * nrm;
            sraw = r[3];
            if (noise) sraw = sraw + noise[ray * s + j] * noise_std;
            const float sig = sraw > 0.0f ? sraw : 0.0f;
            x = -sig * dist;
            alpha = -nrf_expf(x) + 1.0f;
            const float l = lt[j];
            trans = nrf_expf(l);
            col[0] = nrf_sigmoidf(r[0]); col[1] = nrf_sigmoidf(r[1]); col[2] = nrf_sigmoidf(r[2]);
            gw = gr * col[0] + gg * col[1] + gb * col[2];
            if (white) gw -= gsum;
            const float cl = fminf(fmaxf(l, -100.0f), 5.0f);
            gL = gw * alpha * nrf_expf(cl);                // g_T * dT/dL with TruncExp's clamped derivative
        }
        // suffix over LATER samples: total of this block minus the inclusive prefix, plus later blocks
        const double incl = wave_incl_scan_d((double)gL, lane);
        const double total = __shfl(incl, 63);
        const double later = suffix + (total - incl);
        suffix += total;
        if (live) {
            float *g = g_raw + (ray * s + j) * c;
            const float w = alpha * trans;
            g[0] = gr * w * (col[0] * (1.0f - col[0]));
            g[1] = gg * w * (col[1] * (1.0f - col[1]));
            g[2] = gb * w * (col[2] * (1.0f - col[2]));
            float g_alpha = gw * trans;
            const float om = 1.0f - alpha;
            if (om >= 1e-10f) g_alpha -= (float)later / om;
            const float cx = fminf(fmaxf(x, -100.0f), 5.0f);
            const float g_x = -g_alpha * nrf_expf(cx);
            g[3] = (sraw > 0.0f) ? -g_x * dist : 0.0f;
            for (int k = 4; k < c; k++) g[k] = 0.0f;
        }
    }
}

// d loss / d sigma = 0 where the embedder's keep mask is false (raw[~keep, -1] = 0 is an in-place overwrite in the forward)
__global__ void k_mask_grad(int64_t p, int c, const uint8_t *__restrict__ keep, float *__restrict__ g_raw, const int32_t *__restrict__ src)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p && !keep[src ? (int64_t)src[i] : i]) g_raw[i * c + (c - 1)] = 0.0f;          // src: the keep mask by feature COLUMN, point i's column = src[i]
}

// HashEmbedder (fp32 tables [L][2^T][F]): row gradient = g * wz * wy * wx in autograd's chain order
template <int F>
__global__ void k_hash_ngp_bwd(HashParams hp, const float *__restrict__ pts, int64_t p, const float *__restrict__ g_emb, int g_stride, float *__restrict__ g_table)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= p) return;
    const float x[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
    float w[3];
    int32_t idx[3];
    const float res = hp.level_scale[l];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(x[a], hp.bbox.mx[a]), hp.bbox.mn[a]);
        const float grid = (hp.bbox.mx[a] - hp.bbox.mn[a]) / res;
        const float fl = floorf((c - hp.bbox.mn[a]) / grid);
        idx[a] = (int32_t)fl;
        const float vmin = fl * grid + hp.bbox.mn[a];
        const float vmax = vmin + grid;
        w[a] = (x[a] - vmin) / (vmax - vmin);
    }
    float g[F];
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; f++) { g[f] = g_emb[i * g_stride + l * F + f]; any |= g[f] != 0.0f; }
    if (!any) return;
    float *tl = g_table + (int64_t)l * ((int64_t)1 << hp.log2_t) * F;
    const uint32_t hmask = (1u << hp.log2_t) - 1u;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint32_t cx = (uint32_t)idx[0] + ((c >> 2) & 1), cy = (uint32_t)idx[1] + ((c >> 1) & 1), cz = (uint32_t)idx[2] + (c & 1);
        const uint32_t h = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & hmask;
        const float wx = ((c >> 2) & 1) ? w[0] : 1.0f - w[0], wy = ((c >> 1) & 1) ? w[1] : 1.0f - w[1], wz = (c & 1) ? w[2] : 1.0f - w[2];
#pragma unroll
        for (int f = 0; f < F; f++) unsafeAtomicAdd(tl + (int64_t)h * F + f, ((g[f] * wz) * wy) * wx);
    }
}

// CuHashEmbedder (CuHashEmbedder.cu:105-216): grad_in = fp16(g * 128) (:297), each corner adds fp16(grad_in * w) (:196-197) into the
// level's slice of the pool at the forward's (quirky) element offset, result / 128 (:323).  The reference accumulates with fp16
// atomics (order-dependent, saturating); here the per-corner contributions are rounded exactly as there but ACCUMULATED IN FP32.
template <int F>
__global__ void k_hash_cu_bwd(HashParams hp, const float *__restrict__ pts, int64_t p, const float *__restrict__ g_emb, int g_stride, float *__restrict__ g_table)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= p) return;
    float fr[3];
    uint32_t pos[3];
    const float mul = hp.level_scale[l];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = fmaxf(fminf(pts[i * 3 + a], hp.bbox.mx[a]), hp.bbox.mn[a]);      // QueryPoints = the clamped points (CuHashEmbedder.cpp:92-96)
        float q = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]) * mul;
        q = q + hp.bias[l * 3 + a];
        const float fl = floorf(q);
        pos[a] = (uint32_t)fl;
        fr[a] = q - fl;
    }
    float g[F];
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; f++) { g[f] = __half2float(__float2half_rn(g_emb[i * g_stride + l * F + f] * 128.0f)); any |= g[f] != 0.0f; }
    if (!any) return;
    const uint32_t pa = hp.primes[l * 3 + 0], pb = hp.primes[l * 3 + 1], pc = hp.primes[l * 3 + 2];
    const uint32_t lsz = hp.local_size[l];
    float *tl = g_table + hp.local_idx[l];
    const float a = fr[0], b = fr[1], c = fr[2];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t hx = (pos[0] + ((k >> 2) & 1)) * pa, hy = (pos[1] + ((k >> 1) & 1)) * pb, hz = (pos[2] + (k & 1)) * pc;
        const uint32_t ps = ((lsz & (lsz - 1u)) == 0u) ? ((hx ^ hy ^ hz) & (lsz - 1u)) : ((hx ^ hy ^ hz) % lsz);       // local_size is a power of two for T >= 4
        const float w = ((k & 4) ? a : 1.0f - a) * ((k & 2) ? b : 1.0f - b) * ((k & 1) ? c : 1.0f - c);
#pragma unroll
        for (int f = 0; f < F; f++) unsafeAtomicAdd(tl + (size_t)ps * F + f, __half2float(__float2half_rn(g[f] * w)) * (1.0f / 128.0f));
    }
}

// Ray-coherent variants: float atomics run at the memory side and scattered ones (one row per lane) at ~1/17 of the shaped rate
// (0.08 vs 1.3 TB/s of added bytes), so the count of atomics IS the cost of the table gradient.  Consecutive samples of a ray
// (importance sampling packs them densely) mostly sit in the same voxel of a level: one thread walks SEG consecutive samples of one ray
// at one level, sums the eight corner contributions in registers while the voxel does not change and issues the atomics only when it
// does.  Same addends as the per-point kernels, summed in a different order (fp32).
// samples of a ray one thread walks; training step of 16 384 rays, same call (docs/history/profiles/round4/r5b_*): 8: 8.0 ms, 12: 8.65, 16: 7.85, 24: 8.6, 32: 7.8, 48: 8.55
#ifndef NRF_BWD_SEG
#define NRF_BWD_SEG 16
#endif
constexpr int BWD_SEG = NRF_BWD_SEG;
// training step of 16 384 rays x 192 samples, same call, alternating builds (docs/history/profiles/round4/r5a_*): 2^16 10.3 ms, 2^17 8.1-8.9, 2^18 7.84, 2^19 8.04, 2^20 8.6, 2^22 9.25
// ... and with the 8-byte records of the binned form (r5n_*): 2^17 6.4 ms, 2^18 5.93, 2^19 6.2, 2^20 6.75 -- a pass of 2^18 points emits ~130 MB of records, which the 256-MB
// Infinity Cache still holds between the emit pass and k_bin_accumulate
#ifndef NRF_PACKED_GROUP_LOG2
#define NRF_PACKED_GROUP_LOG2 18
#endif
constexpr int64_t PACKED_GROUP_PTS = (int64_t)1 << NRF_PACKED_GROUP_LOG2;      // points per fixed-point pass of nrf_hash_backward_rays_packed / _binned

// Q (F == 2 only): both features of a table entry leave in ONE 64-bit integer atomic.  The L2 atomic units retire ~21-24 G operations/s
// whatever the operand type (tools/scratch/atomic_bench.hip: fp32, f64, u32, u64 and packed-f16 adds all land there), so the table gradient
// is bound by the NUMBER of atomics and the packed form halves it.  Each feature is a 32-bit fixed-point field, value * qscale rounded to
// nearest, packed as hi * 2^32 + lo in two's complement: integer addition of such words adds the fields exactly as long as neither field's
// total leaves int32 (the low field's borrows are undone by the sign-extending decode in k_unpack_q).  qscale is a power of two chosen on the
// device from a rigorous bound on any entry's total (k_level_mass), so the fields cannot overflow.
// BINNED (MODE 1 = count, 2 = emit; Q only): the same walk, but a flushed contribution becomes a RECORD instead of an atomic.  The table's 64-bit words are cut
// into bins of 2^14 consecutive words (128 KB of packed accumulators: one workgroup's LDS); a count pass sizes the bins, an exclusive scan places them, an
// emit pass writes {word, q0, q1} records bin by bin, and k_bin_accumulate sums each bin in LDS (ds_add_u64 on the packed word) and adds the decoded fields to
// the fp32 gradient: the ~48 contributions an entry of a fine level receives per step merge in LDS instead of serialising in the L2 atomic units.  Integer
// sums do not depend on order: the result equals the atomic path's bit for bit.
constexpr int BIN_SHIFT = 14;
constexpr int BIN_WORDS = 1 << BIN_SHIFT;
constexpr int BIN_MAX_PER_LEVEL = 40;        // bins one level's words can touch: 2^19 / 2^14 = 32, + 1 for an unaligned base (+ margin)
// A record = 8 bytes: the word's 14 bits inside its bin (the bin is where the record lies) and the two fixed-point addends as 25-bit two's-complement fields.  The
// records of a pass are written once and read once through HBM (~16 M of them per 2^18 points): that traffic IS the time of the emit pass and of k_bin_accumulate
// -- training step, same call: 16-byte records {word, q0, q1, pad} 6.55 ms, 12-byte 6.20, 8-byte 5.9 (docs/history/profiles/round4/r5l_*, r5m_*).
// An addend outside the field (|q| >= 2^24) goes to a side list as {word, q0, q1} instead: the scale bounds the sum of |q| over a level's records of a pass by 2^30
// (k_qscale), so at most 64 records per level and field can be that large -- the list holds 128 per level and cannot overflow.
typedef unsigned long long BinRec;
struct BinOvf { uint32_t word; int32_t q0, q1; };
constexpr int BIN_OVF_PER_LEVEL = 128;
__device__ __forceinline__ bool bin_rec_fits(int32_t q0, int32_t q1) { return q0 >= -(1 << 24) && q0 < (1 << 24) && q1 >= -(1 << 24) && q1 < (1 << 24); }
__device__ __forceinline__ BinRec bin_rec_pack(uint32_t word, int32_t q0, int32_t q1)
{
    return (BinRec)(word & (uint32_t)((1 << 14) - 1)) | ((BinRec)((uint32_t)q0 & 0x1ffffffu) << 14) | ((BinRec)((uint32_t)q1 & 0x1ffffffu) << 39);
}
struct BinSink {
    uint32_t *wg_hist;       // [levels][workgroups][BIN_MAX_PER_LEVEL] records of a workgroup per bin (count pass writes, emit pass reads)
    uint32_t *gcount;        // [nbins + 1] records per bin (count pass); zeroed again by k_bin_scan
    uint32_t *cursor;        // [nbins] next free record slot of a bin (scan initialises to the bin's start)
    BinRec *rec;             // records
    uint32_t *ovf_count;     // entries of the side list (k_bin_scan zeroes it between the count pass and the emit pass)
    BinOvf *ovf;             // [BIN_OVF_PER_LEVEL * levels]
    uint32_t ovf_cap;
};

template <int F, bool CU, bool Q, int MODE>
__device__ __forceinline__ void hash_bwd_walk(const HashParams &hp, const float *__restrict__ pts, int64_t n, int s, const float *__restrict__ g_emb, int g_stride,
                                              float *__restrict__ g_table, const float *__restrict__ qscale_p, const BinSink &sink, uint32_t *bin_cnt, const uint32_t *bin_base);

template <int F, bool CU, bool Q = false, int MODE = 0>
__global__ void k_hash_bwd_ray(HashParams hp, const float *__restrict__ pts, int64_t n, int s, const float *__restrict__ g_emb, int g_stride,
                               float *__restrict__ g_table, const float *__restrict__ qscale_p = nullptr, BinSink sink = BinSink{})
{
    __shared__ uint32_t bin_cnt[MODE ? BIN_MAX_PER_LEVEL : 1];
    __shared__ uint32_t bin_base[MODE ? BIN_MAX_PER_LEVEL : 1];
    if constexpr (MODE != 0) {
        static_assert(MODE == 0 || Q, "binning carries the packed fixed-point fields");
        uint32_t *mine = sink.wg_hist + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BIN_MAX_PER_LEVEL;
        if (threadIdx.x < BIN_MAX_PER_LEVEL) {
            if (MODE == 1) bin_cnt[threadIdx.x] = 0;
            else {          // reserve this workgroup's slots in every bin it feeds; bin_cnt becomes the running rank inside the reservation
                const uint32_t c = mine[threadIdx.x];
                const uint32_t wb0 = (uint32_t)((CU ? (int64_t)hp.local_idx[blockIdx.y] : (int64_t)blockIdx.y * ((int64_t)1 << hp.log2_t) * F) / 2) >> BIN_SHIFT;
                bin_base[threadIdx.x] = c ? atomicAdd(sink.cursor + wb0 + threadIdx.x, c) : 0u;
                bin_cnt[threadIdx.x] = 0;
            }
        }
        __syncthreads();
    }
    hash_bwd_walk<F, CU, Q, MODE>(hp, pts, n, s, g_emb, g_stride, g_table, qscale_p, sink, bin_cnt, bin_base);
    if constexpr (MODE == 1) {
        __syncthreads();
        if (threadIdx.x < BIN_MAX_PER_LEVEL) {
            uint32_t *mine = sink.wg_hist + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BIN_MAX_PER_LEVEL;
            const uint32_t c = bin_cnt[threadIdx.x];
            mine[threadIdx.x] = c;
            const uint32_t wb0 = (uint32_t)((CU ? (int64_t)hp.local_idx[blockIdx.y] : (int64_t)blockIdx.y * ((int64_t)1 << hp.log2_t) * F) / 2) >> BIN_SHIFT;
            if (c) atomicAdd(sink.gcount + wb0 + threadIdx.x, c);
        }
    }
}

template <int F, bool CU, bool Q, int MODE>
__device__ __forceinline__ void hash_bwd_walk(const HashParams &hp, const float *__restrict__ pts, int64_t n, int s, const float *__restrict__ g_emb, int g_stride,
                                              float *__restrict__ g_table, const float *__restrict__ qscale_p, const BinSink &sink, uint32_t *bin_cnt, const uint32_t *bin_base)
{
    const int nseg = (s + BWD_SEG - 1) / BWD_SEG;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (t >= n * nseg) return;
    const int64_t ray = t / nseg;
    const int j0 = (int)(t - ray * nseg) * BWD_SEG;
    const int j1 = (j0 + BWD_SEG < s) ? j0 + BWD_SEG : s;
    float acc[8][F];
    uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    bool have = false;
    const float scale = hp.level_scale[l];
    float *tl;
    uint32_t pa = 0, pb = 0, pc = 0, lsz = 1, hmask = 0;
    if constexpr (CU) {
        tl = g_table + hp.local_idx[l];
        pa = hp.primes[l * 3 + 0]; pb = hp.primes[l * 3 + 1]; pc = hp.primes[l * 3 + 2]; lsz = hp.local_size[l];
    } else {
        tl = g_table + (int64_t)l * ((int64_t)1 << hp.log2_t) * F;
        hmask = (1u << hp.log2_t) - 1u;
    }
    const bool lsz_pow2 = (lsz & (lsz - 1u)) == 0u;
    float qscale = 1.0f;
    if constexpr (Q) qscale = qscale_p[0];
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t cx = cur[0] + ((k >> 2) & 1), cy = cur[1] + ((k >> 1) & 1), cz = cur[2] + (k & 1);
            // `% local_size` (CuHashEmbedder.cu:70-77): local_size = (2^T >> 4) << 4 is a power of two for every T >= 4 (wave-uniform test, as cu_blend in encode.h):
            // the general 32-bit modulo is ~30 vector instructions, eight times per flush
            const uint32_t hv = (cx * pa) ^ (cy * pb) ^ (cz * pc);
            const uint32_t row = CU ? (lsz_pow2 ? (hv & (lsz - 1u)) : (hv % lsz)) : ((cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & hmask);
            if constexpr (Q) {
                static_assert(!Q || F == 2, "packed atomics carry exactly two features");
                const int32_t q0 = __float2int_rn(acc[k][0] * qscale), q1 = __float2int_rn(acc[k][F - 1] * qscale);
                if constexpr (MODE == 0) {
                    if (q0 | q1) atomicAdd(reinterpret_cast<unsigned long long *>(tl) + row, (unsigned long long)(((int64_t)q1 << 32) + (int64_t)q0));
                } else if (q0 | q1) {
                    const uint32_t wbase = (uint32_t)((tl - g_table) / 2);                 // the level's first 64-bit word
                    const uint32_t word = wbase + row, b = (word >> BIN_SHIFT) - (wbase >> BIN_SHIFT);
                    if (!bin_rec_fits(q0, q1)) {                                             // (neither pass counts it)
                        if constexpr (MODE == 2) {
                            const uint32_t k2 = atomicAdd(sink.ovf_count, 1u);
                            if (k2 < sink.ovf_cap) sink.ovf[k2] = BinOvf{word, q0, q1};
                            else {
                                // beyond the side list (k_qscale's bound makes this unreachable today: at most BIN_OVF_PER_LEVEL such addends per level and pass): never
                                // dropped -- added to the fp32 gradient directly.  k_bin_accumulate's read-modify-write of this word runs after this kernel (stream order)
                                float *gw = g_table + (size_t)word * 2;
                                unsafeAtomicAdd(gw, (float)q0 * qscale_p[1]);
                                unsafeAtomicAdd(gw + 1, (float)q1 * qscale_p[1]);
                            }
                        }
                    } else if constexpr (MODE == 1) atomicAdd(bin_cnt + b, 1u);
                    else {
                        const uint32_t rank = atomicAdd(bin_cnt + b, 1u);
                        sink.rec[(size_t)bin_base[b] + rank] = bin_rec_pack(word, q0, q1);
                    }
                }
            } else {
#pragma unroll
                for (int f = 0; f < F; f++)
                    if (acc[k][f] != 0.0f) unsafeAtomicAdd(tl + (size_t)row * F + f, acc[k][f]);
            }
        }
    };
    // (requesting the segment's 16 gradients and points up front instead of where they are used -- 5 registers per sample, the loop unrolled -- was slower: training step
    // 6.33-6.45 against 6.04 ms, same call, docs/history/profiles/round4/r5q_*)
    for (int j = j0; j < j1; j++) {
        const int64_t i = ray * s + j;
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; f++) {
            g[f] = g_emb[i * g_stride + l * F + f];
            if constexpr (CU) g[f] = __half2float(__float2half_rn(g[f] * 128.0f));
            any |= g[f] != 0.0f;
        }
        if (!any) continue;
        uint32_t pos[3];
        float w[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float x = pts[i * 3 + a];
            const float c = fmaxf(fminf(x, hp.bbox.mx[a]), hp.bbox.mn[a]);
            if constexpr (CU) {
                float q = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]) * scale;
                q = q + hp.bias[l * 3 + a];
                const float fl = floorf(q);
                pos[a] = (uint32_t)fl; w[a] = q - fl;
            } else {
                const float grid = (hp.bbox.mx[a] - hp.bbox.mn[a]) / scale;
                const float fl = floorf((c - hp.bbox.mn[a]) / grid);
                pos[a] = (uint32_t)(int32_t)fl;
                const float vmin = fl * grid + hp.bbox.mn[a];
                const float vmax = vmin + grid;
                w[a] = (x - vmin) / (vmax - vmin);
            }
        }
        if (!have || pos[0] != cur[0] || pos[1] != cur[1] || pos[2] != cur[2]) {
            if (have) flush();
            have = true;
            cur[0] = pos[0]; cur[1] = pos[1]; cur[2] = pos[2];
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int f = 0; f < F; f++) acc[k][f] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float wx = ((k >> 2) & 1) ? w[0] : 1.0f - w[0], wy = ((k >> 1) & 1) ? w[1] : 1.0f - w[1], wz = (k & 1) ? w[2] : 1.0f - w[2];
#pragma unroll
            for (int f = 0; f < F; f++) {
                if constexpr (CU) acc[k][f] += __half2float(__float2half_rn(g[f] * (wx * wy * wz))) * (1.0f / 128.0f);
                else acc[k][f] += ((g[f] * wz) * wy) * wx;
            }
        }
    }
    if (have) flush();
}

// The same walk with one LANE PER FEATURE (F = 4, 8: LeRF's language grid): F neighbouring lanes walk one ray segment together, each carrying its own feature's eight corner sums.
// A flush is then eight atomic instructions whose lanes of a group add to F CONSECUTIVE floats of one table entry (a 16- / 32-byte segment of one line) instead of F x 8
// instructions with every lane in a row of its own -- float atomics are priced by the segments an instruction touches (MI355X_MICROARCH.md, Global float atomics).  The sums are
// the thread-per-segment walk's, term for term: the same samples are skipped (gradient zero in all F features: a group-wide test), cells change at the same samples.
template <int F, bool CU>
__global__ void __launch_bounds__(256) k_hash_bwd_ray_fl(HashParams hp, const float *__restrict__ pts, int64_t n, int s, const float *__restrict__ g_emb, int g_stride,
                                                         float *__restrict__ g_table)
{
    static_assert(F == 4 || F == 8, "a group of F lanes inside one wavefront");
    const int nseg = (s + BWD_SEG - 1) / BWD_SEG;
    const int64_t tt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = tt / F;
    const int f = (int)(tt - t * F);
    const int l = blockIdx.y;
    if (t >= n * nseg) return;                      // whole groups leave together (the thread count per level is a multiple of F)
    const int64_t ray = t / nseg;
    const int j0 = (int)(t - ray * nseg) * BWD_SEG;
    const int j1 = (j0 + BWD_SEG < s) ? j0 + BWD_SEG : s;
    const int gshift = (int)(threadIdx.x & 63) & ~(F - 1);       // the group's first lane inside its wavefront
    float acc[8];
    uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    bool have = false;
    const float scale = hp.level_scale[l];
    float *tl;
    uint32_t pa = 0, pb = 0, pc = 0, lsz = 1, hmask = 0;
    if constexpr (CU) {
        tl = g_table + hp.local_idx[l];
        pa = hp.primes[l * 3 + 0]; pb = hp.primes[l * 3 + 1]; pc = hp.primes[l * 3 + 2]; lsz = hp.local_size[l];
    } else {
        tl = g_table + (int64_t)l * ((int64_t)1 << hp.log2_t) * F;
        hmask = (1u << hp.log2_t) - 1u;
    }
    const bool lsz_pow2 = (lsz & (lsz - 1u)) == 0u;
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t cx = cur[0] + ((k >> 2) & 1), cy = cur[1] + ((k >> 1) & 1), cz = cur[2] + (k & 1);
            const uint32_t hv = (cx * pa) ^ (cy * pb) ^ (cz * pc);
            const uint32_t row = CU ? (lsz_pow2 ? (hv & (lsz - 1u)) : (hv % lsz)) : ((cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & hmask);
            if (acc[k] != 0.0f) unsafeAtomicAdd(tl + (size_t)row * F + f, acc[k]);
        }
    };
    for (int j = j0; j < j1; j++) {
        const int64_t i = ray * s + j;
        float g = g_emb[i * g_stride + l * F + f];
        if constexpr (CU) g = __half2float(__float2half_rn(g * 128.0f));
        const unsigned long long nz = __ballot(g != 0.0f);
        if (((nz >> gshift) & ((1ull << F) - 1ull)) == 0ull) continue;
        uint32_t pos[3];
        float w[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float x = pts[i * 3 + a];
            const float c = fmaxf(fminf(x, hp.bbox.mx[a]), hp.bbox.mn[a]);
            if constexpr (CU) {
                float q = (c - hp.bbox.mn[a]) / (hp.bbox.mx[a] - hp.bbox.mn[a]) * scale;
                q = q + hp.bias[l * 3 + a];
                const float fl = floorf(q);
                pos[a] = (uint32_t)fl; w[a] = q - fl;
            } else {
                const float grid = (hp.bbox.mx[a] - hp.bbox.mn[a]) / scale;
                const float fl = floorf((c - hp.bbox.mn[a]) / grid);
                pos[a] = (uint32_t)(int32_t)fl;
                const float vmin = fl * grid + hp.bbox.mn[a];
                const float vmax = vmin + grid;
                w[a] = (x - vmin) / (vmax - vmin);
            }
        }
        if (!have || pos[0] != cur[0] || pos[1] != cur[1] || pos[2] != cur[2]) {
            if (have) flush();
            have = true;
            cur[0] = pos[0]; cur[1] = pos[1]; cur[2] = pos[2];
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float wx = ((k >> 2) & 1) ? w[0] : 1.0f - w[0], wy = ((k >> 1) & 1) ? w[1] : 1.0f - w[1], wz = (k & 1) ? w[2] : 1.0f - w[2];
            if constexpr (CU) acc[k] += __half2float(__float2half_rn(g * (wx * wy * wz))) * (1.0f / 128.0f);
            else acc[k] += ((g * wz) * wy) * wx;
        }
    }
    if (have) flush();
}

// mass[l] = sum over the points of max_f |g[pt][l][f]|: no table entry of level l can receive more than that (corner weights are in [0,1]
// and sum to one per point), whatever the hash collisions and however the samples cluster.  Accumulated in double.
// HashEmbedder mode only: its interpolation weights are computed from the UNCLAMPED coordinate (NeRF.cpp:265-277), so a point outside the box
// has weights beyond [0,1]; `pts` != NULL multiplies the point's mass by prod_a (1 + excess_a / finest cell size), a bound for every level.
// blockIdx.y = pass (group of `group_pts` points; the launches of one pass at a time use gridDim.y = 1 and group_pts = p): its points, its row of `mass`.
__global__ void __launch_bounds__(256) k_level_mass(int64_t p, int L, int F, const float *__restrict__ g, double *__restrict__ mass, const float *__restrict__ pts,
                                                    Bbox bbox, float finest_res, int64_t group_pts)
{
    __shared__ double red[4][NRF_MAX_LEVELS];
    {
        const int64_t first = (int64_t)blockIdx.y * group_pts;
        g += first * (int64_t)(L * F);
        if (pts) pts += first * 3;
        mass += (int64_t)blockIdx.y * L;
        p = (p - first) < group_pts ? (p - first) : group_pts;
    }
    const int l = threadIdx.x % 16, sub = threadIdx.x / 16;          // 16 threads share a row: coalesced for the L = 16 default
    double m[(NRF_MAX_LEVELS + 15) / 16] = {};
    for (int64_t i = (int64_t)blockIdx.x * 16 + sub; i < p; i += (int64_t)gridDim.x * 16)
    {
        float wb = 1.0f;
        if (pts) {
            for (int a = 0; a < 3; a++) {
                const float x = pts[i * 3 + a], ex = fmaxf(fmaxf(bbox.mn[a] - x, x - bbox.mx[a]), 0.0f);
                wb *= 1.0f + ex * finest_res / (bbox.mx[a] - bbox.mn[a]) * 1.001f;
            }
        }
        for (int ll = l, q = 0; ll < L; ll += 16, q++) {
            float a = 0.0f;
            for (int f = 0; f < F; f++) a = fmaxf(a, fabsf(g[i * (int64_t)(L * F) + ll * F + f]));
            m[q] += (double)a * (double)wb;
        }
    }
    for (int ll = l, q = 0; ll < L; ll += 16, q++) {
        double v = m[q];
        // lanes l, l+16, l+32, l+48 of a wave hold the same level
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][ll] = v;
    }
    __syncthreads();
    if (threadIdx.x < L) unsafeAtomicAdd(mass + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// qs[0] = 2^k with bound * 2^k <= 2^30 (k clamped to +-96), qs[1] = 2^-k.  bound = the largest mass any memory word can collect: one level's
// in the HashEmbedder layout, two adjacent levels' in the CuHashEmbedder layout (whose level blocks overlap by half, CuHashEmbedder.cpp:62-68),
// times 1.01 for the fp16 rounding of the CuHashEmbedder addends.  The 2^30 leaves 2^30 units for the round-to-nearest of the individual addends.
__global__ void k_qscale(int L, int overlap, const double *__restrict__ mass, float *__restrict__ qs)
{
    mass += (int64_t)blockIdx.x * L; qs += (int64_t)blockIdx.x * 2;          // one block per pass
    double b = 0.0;
    for (int l = 0; l < L; l++) {
        const double v = mass[l] + ((overlap && l + 1 < L) ? mass[l + 1] : 0.0);
        b = v > b ? v : b;
    }
    int k = 0;
    if (b > 0.0) { int e; (void)frexp(b * 1.01, &e); k = 30 - e; }      // b * 1.01 < 2^e
    k = k < -96 ? -96 : (k > 96 ? 96 : k);
    qs[0] = ldexpf(1.0f, k); qs[1] = ldexpf(1.0f, -k);
}

// g_table[2e], [2e+1] += the two fixed-point fields of word e
__global__ void k_unpack_q(int64_t entries, unsigned long long *__restrict__ q, const float *__restrict__ qs, float *__restrict__ g_table)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= entries) return;
    const int64_t w = (int64_t)q[e];
    if (w == 0) return;
    q[e] = 0;                         // leaves the word table zero for the next group / call
    const int32_t lo = (int32_t)(uint32_t)(w & 0xffffffffll);
    const int32_t hi = (int32_t)((w - (int64_t)lo) >> 32);
    float2 *gp = reinterpret_cast<float2 *>(g_table) + e;
    float2 v = *gp;
    v.x += (float)lo * qs[1]; v.y += (float)hi * qs[1];
    *gp = v;
}

// start[b] = records of the bins before b (exclusive scan; start[nbins] = total), cursor[b] = start[b]
// ... and leaves the counts zeroed for the next pass (nothing reads them after this)
__global__ void k_bin_scan(int nbins, uint32_t *__restrict__ gcount, uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t *__restrict__ ovf_count)
{
    if (threadIdx.x == 0) *ovf_count = 0u;             // the previous pass's side list has been consumed (stream order); the emit pass behind this fills it again
    __shared__ uint32_t part[256];
    const int per = (nbins + 255) / 256, b0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (int i = b0; i < b0 + per && i < nbins; i++) sum += gcount[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < 256; i++) { const uint32_t v = part[i]; part[i] = run; run += v; } start[nbins] = run; }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int i = b0; i < b0 + per && i < nbins; i++) { start[i] = run; cursor[i] = run; run += gcount[i]; gcount[i] = 0u; }
}

// One workgroup per bin: its records summed in LDS (packed 64-bit words, integer addition: exact and order-free), then the two fixed-point fields of every touched
// word decoded and added to the fp32 gradient -- words of a bin belong to this workgroup alone, so the read-modify-write needs no atomic.
constexpr int BIN_THREADS = 1024;          // one workgroup per CU (128 KB of LDS): sixteen waves, four record loads in flight per lane, to cover the loads' latency
__global__ void __launch_bounds__(BIN_THREADS) k_bin_accumulate(const uint32_t *__restrict__ start, const BinRec *__restrict__ rec, const float *__restrict__ qs, int64_t entries,
                                                                float *__restrict__ g_table, const uint32_t *__restrict__ ovf_count, const BinOvf *__restrict__ ovf, uint32_t ovf_cap)
{
    extern __shared__ unsigned long long bin_acc[];
    const uint32_t r0 = start[blockIdx.x], r1 = start[blockIdx.x + 1];
    uint32_t novf = *ovf_count;
    novf = novf < ovf_cap ? novf : ovf_cap;
    if (r0 == r1 && novf == 0) return;
    for (int e = threadIdx.x; e < BIN_WORDS; e += BIN_THREADS) bin_acc[e] = 0ull;
    __syncthreads();
    auto add = [&](const BinRec r) {
        const int32_t q0 = (int32_t)((uint32_t)(r >> 14) << 7) >> 7;                      // bits 14..38, sign-extended
        const int64_t q1 = (int64_t)r >> 39;                                              // bits 39..63
        atomicAdd(bin_acc + ((uint32_t)r & (uint32_t)(BIN_WORDS - 1)), (unsigned long long)((q1 << 32) + (int64_t)q0));
    };
    uint32_t i = r0 + threadIdx.x;
    for (; i + 3u * BIN_THREADS < r1; i += 4u * BIN_THREADS) {
        const BinRec a = rec[i], b = rec[i + BIN_THREADS], c = rec[i + 2u * BIN_THREADS], d = rec[i + 3u * BIN_THREADS];
        add(a); add(b); add(c); add(d);
    }
    for (; i < r1; i += BIN_THREADS) add(rec[i]);
    for (uint32_t k = threadIdx.x; k < novf; k += BIN_THREADS) {                          // the few addends beyond the 25-bit fields
        const BinOvf o = ovf[k];
        if ((o.word >> BIN_SHIFT) == blockIdx.x) atomicAdd(bin_acc + (o.word & (uint32_t)(BIN_WORDS - 1)), (unsigned long long)(((int64_t)o.q1 << 32) + (int64_t)o.q0));
    }
    __syncthreads();
    const float inv = qs[1];
    for (int e = threadIdx.x; e < BIN_WORDS; e += BIN_THREADS) {
        const int64_t w = (int64_t)bin_acc[e];
        if (w == 0) continue;
        const int64_t word = (int64_t)blockIdx.x * BIN_WORDS + e;
        if (word >= entries) continue;
        const int32_t lo = (int32_t)(uint32_t)(w & 0xffffffffll);
        const int32_t hi = (int32_t)((w - (int64_t)lo) >> 32);
        float2 *gp = reinterpret_cast<float2 *>(g_table) + word;
        float2 v = *gp;
        v.x += (float)lo * inv; v.y += (float)hi * inv;
        *gp = v;
    }
}

// TotalVariationLoss (NeRF.h:255-300): one thread per cube vertex owns the three forward differences starting at it
template <int F>
__global__ void k_tv_loss(int n, int cube, int log2_t, int vx, int vy, int vz, float weight, const float *__restrict__ tl, float *__restrict__ loss, float *__restrict__ g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double tv = 0.0;
    if (t < n * n * n) {
        const int z = t % n, y = (t / n) % n, x = t / (n * n);
        const uint32_t hmask = (1u << log2_t) - 1u;
        auto row = [&](int xx, int yy, int zz) { return (size_t)((((uint32_t)(vx + xx)) ^ ((uint32_t)(vy + yy) * 2654435761u) ^ ((uint32_t)(vz + zz) * 805459861u)) & hmask); };
        const size_t v = row(x, y, z);
        const int c[3] = {x, y, z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (c[a] + 1 >= n) continue;
            const size_t u = row(x + (a == 0), y + (a == 1), z + (a == 2));
#pragma unroll
            for (int f = 0; f < F; f++) {
                const float d = tl[u * F + f] - tl[v * F + f];
                tv += (double)(d * d);
                if (g) {
                    const float gr = weight * (2.0f * d) / (float)cube;
                    unsafeAtomicAdd(g + u * F + f, gr);
                    unsafeAtomicAdd(g + v * F + f, -gr);
                }
            }
        }
    }
    tv = wsum(tv);
    if ((threadIdx.x & 63) == 0 && tv != 0.0) unsafeAtomicAdd(loss, (float)(tv * (double)weight / (double)cube));
}

__global__ void k_adam(int64_t n, float lr_over_bc1, float bc2_sqrt, float b1, float b2, float eps, float *__restrict__ p, const float *__restrict__ g,
                       float *__restrict__ m, float *__restrict__ v, const uint32_t *__restrict__ flags, int n_flags)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // guarded step (nrf_adam_step_guarded): any non-zero word -- the fp16 backward's overflow report -- and nothing is updated, moments included
    for (int f = 0; f < n_flags; f++) if (flags[f] != 0u) return;
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - lr_over_bc1 * (mi / denom);
}

}  // namespace nrf

using namespace nrf;

// stream-ordered scratch that is given back on every path out of its scope, the error returns included
struct AsyncBuf {
    void *p = nullptr;
    hipStream_t st;
    explicit AsyncBuf(hipStream_t s) : st(s) {}
    ~AsyncBuf() { if (p) (void)scratch_give(p, st); }
    AsyncBuf(const AsyncBuf &) = delete;
    AsyncBuf &operator=(const AsyncBuf &) = delete;
};

extern "C" {

int nrf_huber_loss(const float *d_pred, const float *d_target, int64_t count, float *d_loss_mse, float *d_grad, void *stream)
{
    NRF_CHECK_ARG(d_pred && d_target && d_loss_mse && count > 0, "nrf_huber_loss: bad argument");
    hipStream_t st = as_stream(stream);
    AsyncBuf buf(st);
    NRF_HIP(scratch_take(&buf.p, 2 * sizeof(double), st));
    double *acc = static_cast<double *>(buf.p);
    NRF_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(double), st));
    const unsigned grid = (unsigned)(ceil_div(count, 256) < 512 ? ceil_div(count, 256) : 512);
    hipLaunchKernelGGL(k_huber, dim3(grid), dim3(256), 0, st, count, 1.0f / (float)count, d_pred, d_target, acc, d_grad);
    NRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_loss, dim3(1), dim3(1), 0, st, count, acc, d_loss_mse);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_raw2outputs_backward(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c, int white_bkgr,
                             const float *d_g_rgb, float *d_g_raw, void *stream)
{
    return nrf_raw2outputs_backward_noise(d_raw, d_z, d_dirs, d_stride, n, s, c, white_bkgr, nullptr, 0.0f, d_g_rgb, d_g_raw, stream);
}

int nrf_raw2outputs_backward_noise(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c, int white_bkgr,
                                   const float *d_noise, float noise_std, const float *d_g_rgb, float *d_g_raw, void *stream)
{
    NRF_CHECK_ARG(d_raw && d_z && d_dirs && d_g_rgb && d_g_raw && n >= 0 && s >= 1 && c >= 4 && d_stride >= 3, "nrf_raw2outputs_backward: bad argument");
    if (n == 0) return NRF_OK;
    hipStream_t st = as_stream(stream);
    AsyncBuf buf(st);
    NRF_HIP(scratch_take(&buf.p, (size_t)n * s * sizeof(float), st));
    float *lt = static_cast<float *>(buf.p);
    hipLaunchKernelGGL(k_raw2outputs_bwd, dim3((unsigned)ceil_div(n, BW_RAYS)), dim3(64 * BW_RAYS), 0, st, n, s, c, white_bkgr, d_raw, d_z, d_dirs, d_stride, d_g_rgb,
                       d_g_raw, lt, d_noise, noise_std);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_mask_sigma_grad(const uint8_t *d_keep, int64_t p, int c, float *d_g_raw, void *stream)
{
    NRF_CHECK_ARG(d_keep && d_g_raw && p >= 0 && c >= 1, "nrf_mask_sigma_grad: bad argument");
    if (p == 0) return NRF_OK;
    hipLaunchKernelGGL(k_mask_grad, dim3((unsigned)ceil_div(p, 256)), dim3(256), 0, as_stream(stream), p, c, d_keep, d_g_raw, (const int32_t *)nullptr);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_mask_sigma_grad_src(const uint8_t *d_keep_cols, const int32_t *d_src, int64_t p, int c, float *d_g_raw, void *stream)
{
    NRF_CHECK_ARG(d_keep_cols && d_src && d_g_raw && p >= 0 && c >= 1, "nrf_mask_sigma_grad_src: bad argument");
    if (p == 0) return NRF_OK;
    hipLaunchKernelGGL(k_mask_grad, dim3((unsigned)ceil_div(p, 256)), dim3(256), 0, as_stream(stream), p, c, d_keep_cols, d_g_raw, d_src);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_hash_backward(const nrf_hash *h, const float *d_x, int64_t p, const float *d_g_emb, float *d_g_table, void *stream)
{
    NRF_CHECK_ARG(h && d_x && d_g_emb && d_g_table && p >= 0, "nrf_hash_backward: bad argument");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_NGP || h->primes_set, "nrf_hash_backward: CuHashEmbedder-mode grid without primes");
    if (p == 0) return NRF_OK;
    const int F = h->desc.n_features, L = h->desc.n_levels;
    dim3 grid((unsigned)ceil_div(p, 256), (unsigned)L);
    hipStream_t st = as_stream(stream);
#define NRF_BWD(FF)                                                                                                                        \
    do {                                                                                                                                   \
        if (h->desc.mode == NRF_HASH_NGP) hipLaunchKernelGGL(k_hash_ngp_bwd<FF>, grid, dim3(256), 0, st, h->params, d_x, p, d_g_emb, L * F, d_g_table); \
        else hipLaunchKernelGGL(k_hash_cu_bwd<FF>, grid, dim3(256), 0, st, h->params, d_x, p, d_g_emb, L * F, d_g_table);                 \
    } while (0)
    switch (F) {
        case 1: NRF_BWD(1); break;
        case 2: NRF_BWD(2); break;
        case 4: NRF_BWD(4); break;
        case 8: NRF_BWD(8); break;
        default: set_error("nrf_hash_backward: n_features %d not built (1, 2, 4, 8)", F); return NRF_ERR_UNSUPPORTED;
    }
#undef NRF_BWD
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_hash_backward_rays(const nrf_hash *h, const float *d_pts, int64_t n, int s, const float *d_g_emb, float *d_g_table, void *stream)
{
    NRF_CHECK_ARG(h && d_pts && d_g_emb && d_g_table && n >= 0 && s >= 1, "nrf_hash_backward_rays: bad argument");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_NGP || h->primes_set, "nrf_hash_backward_rays: CuHashEmbedder-mode grid without primes");
    if (n == 0) return NRF_OK;
    const int F = h->desc.n_features, L = h->desc.n_levels;
    const int64_t threads = n * ((s + BWD_SEG - 1) / BWD_SEG);
    dim3 grid((unsigned)ceil_div(threads, 256), (unsigned)L);
    hipStream_t st = as_stream(stream);
#define NRF_BWD(FF)                                                                                                                                     \
    do {                                                                                                                                                \
        if (h->desc.mode == NRF_HASH_NGP) hipLaunchKernelGGL((k_hash_bwd_ray<FF, false>), grid, dim3(256), 0, st, h->params, d_pts, n, s, d_g_emb, L * F, d_g_table); \
        else hipLaunchKernelGGL((k_hash_bwd_ray<FF, true>), grid, dim3(256), 0, st, h->params, d_pts, n, s, d_g_emb, L * F, d_g_table);               \
    } while (0)
    // F = 4, 8: a lane per feature (k_hash_bwd_ray_fl); NRF_HASH_BWD_FEATURE_LANES=0 keeps the thread-per-segment walk (A/B, tests)
    static const bool feature_lanes = [] { const char *e = getenv("NRF_HASH_BWD_FEATURE_LANES"); return !(e && e[0] == '0'); }();
#define NRF_BWD_FL(FF)                                                                                                                                  \
    do {                                                                                                                                                \
        dim3 gfl((unsigned)ceil_div(threads * FF, 256), (unsigned)L);                                                                                   \
        if (h->desc.mode == NRF_HASH_NGP) hipLaunchKernelGGL((k_hash_bwd_ray_fl<FF, false>), gfl, dim3(256), 0, st, h->params, d_pts, n, s, d_g_emb, L * F, d_g_table); \
        else hipLaunchKernelGGL((k_hash_bwd_ray_fl<FF, true>), gfl, dim3(256), 0, st, h->params, d_pts, n, s, d_g_emb, L * F, d_g_table);               \
    } while (0)
    switch (F) {
        case 1: NRF_BWD(1); break;
        case 2: NRF_BWD(2); break;
        case 4: if (feature_lanes) NRF_BWD_FL(4); else NRF_BWD(4); break;
        case 8: if (feature_lanes) NRF_BWD_FL(8); else NRF_BWD(8); break;
        default: set_error("nrf_hash_backward_rays: n_features %d not built (1, 2, 4, 8)", F); return NRF_ERR_UNSUPPORTED;
    }
#undef NRF_BWD
#undef NRF_BWD_FL
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

size_t nrf_hash_backward_packed_workspace_bytes(const nrf_hash *h) { return h ? 1024 + (size_t)nrf_hash_table_elems(h) / 2 * 8 : 0; }

int nrf_hash_backward_rays_packed(const nrf_hash *h, const float *d_pts, int64_t n, int s, const float *d_g_emb, float *d_g_table, void *d_workspace,
                                  size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(h && d_pts && d_g_emb && d_g_table && d_workspace && n >= 0 && s >= 1, "nrf_hash_backward_rays_packed: bad argument");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_NGP || h->primes_set, "nrf_hash_backward_rays_packed: CuHashEmbedder-mode grid without primes");
    if (h->desc.n_features != 2) { set_error("nrf_hash_backward_rays_packed: built for 2 features per level (a table entry = one 64-bit word); use nrf_hash_backward_rays"); return NRF_ERR_UNSUPPORTED; }
    if (workspace_bytes < nrf_hash_backward_packed_workspace_bytes(h)) { set_error("nrf_hash_backward_rays_packed: workspace %zu < %zu bytes", workspace_bytes, nrf_hash_backward_packed_workspace_bytes(h)); return NRF_ERR_WORKSPACE; }
    if ((reinterpret_cast<uintptr_t>(d_workspace) & 255) || (reinterpret_cast<uintptr_t>(d_g_table) & 7)) { set_error("nrf_hash_backward_rays_packed: workspace must be 256-byte, g_table 8-byte aligned"); return NRF_ERR_INVALID_ARG; }
    if (n == 0) return NRF_OK;
    const int L = h->desc.n_levels;
    hipStream_t st = as_stream(stream);
    double *mass = reinterpret_cast<double *>(d_workspace);                                         // [L] (<= 64 levels = 512 bytes)
    float *qs = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(d_workspace) + 768);
    unsigned long long *q = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(d_workspace) + 1024);
    const int64_t entries = nrf_hash_table_elems(h) / 2;
    NRF_HIP(hipMemsetAsync(d_workspace, 0, 1024 + (size_t)entries * 8, st));
    float *qt = reinterpret_cast<float *>(q);                        // the kernel indexes its table in floats: entry e = floats 2e, 2e+1 = word e
    const bool ngp = h->desc.mode == NRF_HASH_NGP;
    // The bound behind the scale is a sum over ALL points of a pass while a typical entry sees a handful, so the resolution relative to one addend
    // is ~2 * points * 2^-30.  Groups of <= 2^18 points keep that at 5e-4 (the float atomics' own summation order moves entries by ~1e-6); each
    // group costs one sweep over the word table (k_unpack_q, ~25 us at 2^23 words) on top of its atomics.
    const int64_t rays_per_group = (PACKED_GROUP_PTS / s) > 0 ? (PACKED_GROUP_PTS / s) : 1;
    for (int64_t r0 = 0; r0 < n; r0 += rays_per_group) {
        const int64_t nr = (n - r0) < rays_per_group ? (n - r0) : rays_per_group;
        const int64_t p = nr * s;
        const float *gp = d_g_emb + r0 * s * (int64_t)(L * 2), *pp = d_pts + r0 * s * 3;
        if (r0) NRF_HIP(hipMemsetAsync(mass, 0, 512, st));
        hipLaunchKernelGGL(k_level_mass, dim3((unsigned)(ceil_div(p, 16) < 2048 ? ceil_div(p, 16) : 2048)), dim3(256), 0, st, p, L, 2, gp, mass, ngp ? pp : (const float *)nullptr,
                           h->params.bbox, (float)h->desc.finest_resolution, p);
        hipLaunchKernelGGL(k_qscale, dim3(1), dim3(1), 0, st, L, ngp ? 0 : 1, mass, qs);
        const int64_t threads = nr * ((s + BWD_SEG - 1) / BWD_SEG);
        dim3 grid((unsigned)ceil_div(threads, 256), (unsigned)L);
        if (ngp) hipLaunchKernelGGL((k_hash_bwd_ray<2, false, true>), grid, dim3(256), 0, st, h->params, pp, nr, s, gp, L * 2, qt, (const float *)qs);
        else hipLaunchKernelGGL((k_hash_bwd_ray<2, true, true>), grid, dim3(256), 0, st, h->params, pp, nr, s, gp, L * 2, qt, (const float *)qs);
        hipLaunchKernelGGL(k_unpack_q, dim3((unsigned)ceil_div(entries, 256)), dim3(256), 0, st, entries, q, (const float *)qs, d_g_table);
    }
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// workspace of the binned form: header (64 KB: mass [passes][L], scale [passes][2]) | gcount [nbins + 1] | start [nbins + 1] | cursor [nbins] | wg_hist | records
constexpr size_t BIN_HDR = 65536;
constexpr size_t BIN_OVF_BYTES = 256 + ((size_t)BIN_OVF_PER_LEVEL * NRF_MAX_LEVELS * sizeof(BinOvf) + 255) / 256 * 256;      // counter | side list, behind the header
// n_rays < 0: the worst case (a whole group of 2^18 points per pass); otherwise the pass never holds more than n_rays rays and the record buffer is sized for that
static void binned_layout(const nrf_hash *h, int s, int64_t n_rays, int64_t *nbins, int64_t *rays_per_group, int64_t *nwg, size_t *off_hist, size_t *off_rec, size_t *total)
{
    const int L = h->desc.n_levels;
    const int64_t entries = nrf_hash_table_elems(h) / 2;
    *nbins = ceil_div(entries, (int64_t)BIN_WORDS);
    *rays_per_group = (PACKED_GROUP_PTS / s) > 0 ? (PACKED_GROUP_PTS / s) : 1;
    const int64_t held = (n_rays >= 0 && n_rays < *rays_per_group) ? (n_rays > 0 ? n_rays : 1) : *rays_per_group;      // rays a pass can hold at most
    const int64_t threads = held * ((s + BWD_SEG - 1) / BWD_SEG);
    *nwg = ceil_div(threads, (int64_t)256);
    size_t o = BIN_HDR + BIN_OVF_BYTES + align_up((size_t)(*nbins + 1) * 4, 256) * 2 + align_up((size_t)*nbins * 4, 256);
    *off_hist = o;
    o += align_up((size_t)L * *nwg * BIN_MAX_PER_LEVEL * 4, 256);
    *off_rec = o;
    o += (size_t)held * s * 8 * L * sizeof(BinRec);                    // every sample may flush eight records per level
    *total = o;
}

size_t nrf_hash_backward_binned_workspace_bytes(const nrf_hash *h, int s)
{
    if (!h || s < 1) return 0;
    int64_t nbins, rpg, nwg; size_t oh, orr, total;
    binned_layout(h, s, -1, &nbins, &rpg, &nwg, &oh, &orr, &total);
    return total;
}

size_t nrf_hash_backward_binned_workspace_bytes_for(const nrf_hash *h, int64_t n, int s)
{
    if (!h || s < 1 || n < 0) return 0;
    int64_t nbins, rpg, nwg; size_t oh, orr, total;
    binned_layout(h, s, n, &nbins, &rpg, &nwg, &oh, &orr, &total);
    return total;
}

int nrf_hash_backward_rays_binned(const nrf_hash *h, const float *d_pts, int64_t n, int s, const float *d_g_emb, float *d_g_table, void *d_workspace,
                                  size_t workspace_bytes, void *stream)
{
    NRF_CHECK_ARG(h && d_pts && d_g_emb && d_g_table && d_workspace && n >= 0 && s >= 1, "nrf_hash_backward_rays_binned: bad argument");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_NGP || h->primes_set, "nrf_hash_backward_rays_binned: CuHashEmbedder-mode grid without primes");
    if (h->desc.n_features != 2) { set_error("nrf_hash_backward_rays_binned: built for 2 features per level (a table entry = one 64-bit word); use nrf_hash_backward_rays"); return NRF_ERR_UNSUPPORTED; }
    if (h->desc.log2_hashmap_size > 19) { set_error("nrf_hash_backward_rays_binned: a level of 2^%d words touches more than %d bins; use nrf_hash_backward_rays_packed", h->desc.log2_hashmap_size, BIN_MAX_PER_LEVEL); return NRF_ERR_UNSUPPORTED; }
    int64_t nbins, rays_per_group, nwg; size_t off_hist, off_rec, total;
    binned_layout(h, s, n, &nbins, &rays_per_group, &nwg, &off_hist, &off_rec, &total);       // sized for THIS call's rays: any workspace of nrf_hash_backward_binned_workspace_bytes[_for] fits
    if (workspace_bytes < total) { set_error("nrf_hash_backward_rays_binned: workspace %zu < %zu bytes", workspace_bytes, total); return NRF_ERR_WORKSPACE; }
    if ((reinterpret_cast<uintptr_t>(d_workspace) & 255) || (reinterpret_cast<uintptr_t>(d_g_table) & 7)) { set_error("nrf_hash_backward_rays_binned: workspace must be 256-byte, g_table 8-byte aligned"); return NRF_ERR_INVALID_ARG; }
    if (n == 0) return NRF_OK;
    const int L = h->desc.n_levels;
    hipStream_t st = as_stream(stream);
    unsigned char *ws = reinterpret_cast<unsigned char *>(d_workspace);
    // The level masses and scales of ALL passes come first, in two launches (a launch of 2^18 points is latency-bound: 30 us for 33 MB, thirteen of them per
    // training step, plus the scale kernel and a memset per pass: docs/history/profiles/round4/r5c_*); same per-pass sums, same scales, same table gradient bits.
    const int64_t pass_cap = (int64_t)(BIN_HDR / ((size_t)L * 8 + 8));                          // passes whose masses and scales the header holds
    double *mass = reinterpret_cast<double *>(ws);
    float *qs = reinterpret_cast<float *>(ws + (size_t)pass_cap * L * 8);
    const size_t cnt_bytes = align_up((size_t)(nbins + 1) * 4, 256);
    unsigned char *cb = ws + BIN_HDR + BIN_OVF_BYTES;
    uint32_t *gcount = reinterpret_cast<uint32_t *>(cb), *start = reinterpret_cast<uint32_t *>(cb + cnt_bytes), *cursor = reinterpret_cast<uint32_t *>(cb + 2 * cnt_bytes);
    BinSink sink{reinterpret_cast<uint32_t *>(ws + off_hist), gcount, cursor, reinterpret_cast<BinRec *>(ws + off_rec),
                 reinterpret_cast<uint32_t *>(ws + BIN_HDR), reinterpret_cast<BinOvf *>(ws + BIN_HDR + 256), (uint32_t)(BIN_OVF_PER_LEVEL * L)};
    const int64_t entries = nrf_hash_table_elems(h) / 2;
    const bool ngp = h->desc.mode == NRF_HASH_NGP;
    static PerDeviceOnce attr_set;          // idempotent one-time setup per device (common.h)
    const size_t lds = (size_t)BIN_WORDS * 8;
    if (attr_set.needed()) { NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_accumulate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set.done(); }
    // same groups and the same scale as the packed form: the integer fields are identical, so is the result
    const int64_t group_pts = rays_per_group * s;
    int64_t pass = 0;
    for (int64_t r0 = 0; r0 < n; r0 += rays_per_group, pass++) {
        const int64_t nr = (n - r0) < rays_per_group ? (n - r0) : rays_per_group;
        const float *gp = d_g_emb + r0 * s * (int64_t)(L * 2), *pp = d_pts + r0 * s * 3;
        if (pass % pass_cap == 0) {
            // masses and scales of the next pass_cap passes (all of them unless the call holds more than ~450 x 2^18 points); the bin counts are zeroed once, k_bin_scan
            // leaves them zeroed
            const int64_t rays_left = n - r0, passes = ceil_div(rays_left, rays_per_group) < pass_cap ? ceil_div(rays_left, rays_per_group) : pass_cap;
            const int64_t pts_here = (rays_left < passes * rays_per_group ? rays_left : passes * rays_per_group) * s;
            NRF_HIP(hipMemsetAsync(ws, 0, pass == 0 ? BIN_HDR + BIN_OVF_BYTES + cnt_bytes : BIN_HDR, st));
            const int64_t blocks = ceil_div(group_pts < pts_here ? group_pts : pts_here, 16);
            hipLaunchKernelGGL(k_level_mass, dim3((unsigned)(blocks < 2048 ? blocks : 2048), (unsigned)passes), dim3(256), 0, st, pts_here, L, 2, gp, mass, ngp ? pp : (const float *)nullptr,
                               h->params.bbox, (float)h->desc.finest_resolution, group_pts);
            hipLaunchKernelGGL(k_qscale, dim3((unsigned)passes), dim3(1), 0, st, L, ngp ? 0 : 1, mass, qs);
        }
        const float *qsp = qs + (pass % pass_cap) * 2;
        const int64_t threads = nr * ((s + BWD_SEG - 1) / BWD_SEG);
        dim3 grid((unsigned)ceil_div(threads, 256), (unsigned)L);
        if (ngp) hipLaunchKernelGGL((k_hash_bwd_ray<2, false, true, 1>), grid, dim3(256), 0, st, h->params, pp, nr, s, gp, L * 2, d_g_table, qsp, sink);
        else hipLaunchKernelGGL((k_hash_bwd_ray<2, true, true, 1>), grid, dim3(256), 0, st, h->params, pp, nr, s, gp, L * 2, d_g_table, qsp, sink);
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(256), 0, st, (int)nbins, gcount, start, cursor, sink.ovf_count);
        if (ngp) hipLaunchKernelGGL((k_hash_bwd_ray<2, false, true, 2>), grid, dim3(256), 0, st, h->params, pp, nr, s, gp, L * 2, d_g_table, qsp, sink);
        else hipLaunchKernelGGL((k_hash_bwd_ray<2, true, true, 2>), grid, dim3(256), 0, st, h->params, pp, nr, s, gp, L * 2, d_g_table, qsp, sink);
        hipLaunchKernelGGL(k_bin_accumulate, dim3((unsigned)nbins), dim3(BIN_THREADS), lds, st, (const uint32_t *)start, (const BinRec *)sink.rec, qsp, entries, d_g_table, (const uint32_t *)sink.ovf_count,
                           (const BinOvf *)sink.ovf, sink.ovf_cap);
    }
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_hash_tv_loss(const nrf_hash *h, const float *d_table, int level, const int *min_vertex, int cube_size, float weight, float *d_loss, float *d_g_table,
                     void *stream)
{
    NRF_CHECK_ARG(h && d_table && min_vertex && d_loss && cube_size >= 1 && cube_size <= 511, "nrf_hash_tv_loss: bad argument");
    NRF_CHECK_ARG(h->desc.mode == NRF_HASH_NGP, "nrf_hash_tv_loss: the reference defines this regulariser for the LibTorch HashEmbedder only (NeRFExecutor.h:896-913)");
    NRF_CHECK_ARG(level >= 0 && level < h->desc.n_levels, "nrf_hash_tv_loss: level %d outside [0,%d)", level, h->desc.n_levels);
    const int n = cube_size + 1, F = h->desc.n_features, T = h->desc.log2_hashmap_size;
    const size_t off = (size_t)level * ((size_t)1 << T) * F;
    const unsigned grid = (unsigned)ceil_div((int64_t)n * n * n, 256);
    hipStream_t st = as_stream(stream);
    float *g = d_g_table ? d_g_table + off : nullptr;
#define NRF_TV(FF) hipLaunchKernelGGL(k_tv_loss<FF>, dim3(grid), dim3(256), 0, st, n, cube_size, T, min_vertex[0], min_vertex[1], min_vertex[2], weight, d_table + off, d_loss, g)
    switch (F) {
        case 1: NRF_TV(1); break;
        case 2: NRF_TV(2); break;
        case 4: NRF_TV(4); break;
        case 8: NRF_TV(8); break;
        default: set_error("nrf_hash_tv_loss: n_features %d not built", F); return NRF_ERR_UNSUPPORTED;
    }
#undef NRF_TV
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_adam_step_guarded(float *d_p, const float *d_g, float *d_m, float *d_v, int64_t n, float lr, float beta1, float beta2, float eps, int t, const uint32_t *d_flags,
                          int n_flags, void *stream)
{
    NRF_CHECK_ARG(d_p && d_g && d_m && d_v && n >= 0 && t >= 1 && n_flags >= 0 && n_flags <= 16 && (n_flags == 0 || d_flags), "nrf_adam_step: bad argument");
    if (n == 0) return NRF_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, (float)((double)lr / bc1), (float)sqrt(bc2), beta1, beta2, eps,
                       d_p, d_g, d_m, d_v, d_flags, n_flags);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int nrf_adam_step(float *d_p, const float *d_g, float *d_m, float *d_v, int64_t n, float lr, float beta1, float beta2, float eps, int t, void *stream)
{
    return nrf_adam_step_guarded(d_p, d_g, d_m, d_v, n, lr, beta1, beta2, eps, t, nullptr, 0, stream);
}

}  // extern "C"
