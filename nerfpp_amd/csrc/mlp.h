// mlp.h -- MLP handle shared by mlp.hip (generic fp32 path) and the matrix-core translation units (mlp_small_*, mlp_nerf_*, mlp_lerf_*).
#pragma once
#include "common.h"

#include <functional>
#include <vector>

namespace nrf {

enum { MLP_SMALL = 0, MLP_NERF = 1, MLP_LERF = 2 };

// A derived weight image as a GATHER from the parameter blob: element e = convert(kind[e], params[src[e]]), 0 where src < 0.  Built once per handle by decoding
// probe runs of the host packers and verified against the host-packed image (mlp.hip, build_weight_maps); nrf_mlp_set_params then refreshes every image on the
// device, asynchronously -- an optimisation step no longer takes a host round trip.
enum : uint8_t { WM_ZERO = 0, WM_F16_HI = 1, WM_F16_LO = 2, WM_F32 = 3 };
struct WeightMap {
    int32_t *d_src = nullptr;
    uint8_t *d_kind = nullptr;
    int64_t n = 0;
    int elem = 2;              // bytes per element of the image: 2 (fp16, kinds HI / LO) or 4 (fp32)
    void *d_out = nullptr;     // the image (owned by the handle's d_packed_* / layer fields)
    bool scaled = false;       // gathers from the handle's SCALED blob (d_params_scaled): the split-precision operand images
};

// NRF_PREC_F16_SPLIT range safety of the bias-free NeRFSmall: ReLU is positively homogeneous, so layer l may be computed with 2^e_l W_l and the product of the scales
// taken out again at the end -- exact in fp32 (powers of two), while the fp16 (hi, lo) pairs the matrix cores read (weights split at pack time, activations split between
// layers) sit where both halves are fp16 NORMALS whatever the magnitude of the checkpoint's weights.  Scale groups: sigma-net layer l -> group l; colour layer 0 has two
// (its view-direction columns, and its geo columns, which also undo the sigma net's cumulative scale: the geo features arrive scaled); colour layer l >= 1 -> NL + 1 + l.
enum { SMALL_MAX_GROUPS = 12, SMALL_SCALE_INV_SIGMA = 0, SMALL_SCALE_INV_RGB = 1, SMALL_SCALE_HIDDEN = 2, SMALL_SCALE_COUNT = 8 };

struct LinearLayer {
    int in = 0, out = 0;
    float *d_wt = nullptr;      // W^T [in][out] fp32 (lane = output neuron reads coalesced)
    float *d_bias = nullptr;    // [out] or nullptr
    size_t w_off = 0;           // offset of W [out][in] in the parameter blob (floats)
};

// a column range of row-major fp32 rows: the operands of the generic fp32 layer kernels (mlp.hip)
struct Seg {
    const float *p;   // rows
    int stride;       // floats between rows
    int off;          // first column
    int n;            // columns taken
};

}  // namespace nrf

struct nrf_mlp {
    int family = 0;
    nrf_mlp_small_desc small{};
    nrf_mlp_nerf_desc nerf{};
    int in_dims = 0, out_dims = 0;
    int max_width = 0;                       // widest activation row (for workspace sizing)
    // classic network, host re-pack: views_linears_0 o feature_linear as computed by mlp_nerf_pack_f16, handed to mlp_nerf_pack_sigma_f32 of the same upload (then cleared)
    std::vector<float> host_merged, host_merged_b;
    std::vector<nrf::LinearLayer> layers;    // in forward order (see mlp.hip for the per-family order)
    float *d_params = nullptr;               // the fp32 blob, checkpoint order, on device
    int64_t n_params = 0;
    // packed operands of the matrix-core paths (built at create time)
    void *d_packed_f16 = nullptr;
    size_t packed_f16_bytes = 0;
    void *d_packed_split = nullptr;          // NRF_PREC_F16_SPLIT: (hi, lo) fragment pairs
    size_t packed_split_bytes = 0;
    void *d_packed_sigma_f32 = nullptr;      // fp32 MFMA fragments of the sigma net (sigma_small_f32.hip)
    size_t packed_sigma_f32_bytes = 0;
    int lerf_precision = NRF_PREC_F16_MFMA;  // arithmetic of the fused LeRF passes (nrf_lerf_set_precision)
    float *d_lerf_gram = nullptr;            // LeRF device pack: the Gram matrix [256][256] in its packed form (scaled, block triangle), W^T W itself, the bits of its largest entry
    bool lerf_gram_current = false;          // d_lerf_gram's W^T W belongs to the parameters now in d_params (set by the device packer, cleared by a host re-pack)
    bool lerf_device_pack = false;           // the device packers reproduce the host packers' images byte for byte (checked at creation): nrf_mlp_set_params stays on the device
    float lerf_gram_scale = 1.0f;            // LeRF: the Gram matrix of the embedding layer is stored divided by this power of two (fp16 range), see mlp_lerf_mfma.hip
    void *d_packed_bwd = nullptr;            // W^T fragments of the matrix-core backward (mlp_small_bwd_mfma.hip)
    size_t packed_bwd_bytes = 0;
    std::vector<nrf::WeightMap> maps;        // NeRFSmall: every derived image as a gather from d_params (empty: nrf_mlp_set_params repacks on the host)
    // range-safe split precision (see SMALL_MAX_GROUPS above): chosen ON THE DEVICE from the blob at every nrf_mlp_set_params (k_small_scales), in stream order
    float *d_params_scaled = nullptr;        // blob[i] * gscale[group[i]]: what the split images (d_packed_split, the GEO tail of d_packed_sigma_f32) gather from
    uint8_t *d_group = nullptr;              // scale group of blob element i
    float *d_gscale = nullptr;               // [SMALL_MAX_GROUPS] powers of two
    float *d_scales = nullptr;               // [SMALL_SCALE_COUNT] what the kernels read: 2^-S_sigma, 2^-S_colour, 2^S_hidden (all 1 while no scaling is in force)
    float in_rms_hint = 0.25f;               // expected RMS of the position features; nrf_mlp_set_input_rms_hint
    const float *d_in_rms_src = nullptr;     // ... or where to read it on the device: the RMS of the hash table a renderer pairs this network with (render.hip, ensure_scales)
    uint64_t in_rms_seen = 0;                // table upload counter of that hash grid at the last rescale
    bool split_scaling = true;               // false: the split images from the blob as it is (nrf_mlp_set_split_scaling; NRF_SPLIT_UNSCALED=1 makes that the default)
};

namespace nrf {

// bytes of scratch nrf_mlp forward needs for `p` points in precision `prec`
size_t mlp_workspace_bytes(const nrf_mlp *m, int64_t p, int prec);

// x: [p, x_stride] rows holding [input_ch | input_ch_views]; out: [p, out_stride]
int mlp_forward(const nrf_mlp *m, const float *d_x, int x_stride, int64_t p, int prec, float *d_out, int out_stride,
                void *d_ws, size_t ws_bytes, hipStream_t st);

// the generic fp32 building blocks of mlp.hip (forward: an FMA chain per output in ascending k == the oracle; backward: dW by a TN product over the points with one
// atomic add per element and workgroup, g_in by the same forward kernel on the blob's own [out][in] matrix), shared with lerf_train.hip / the classic backward:
//   y[pt][y_off + o] = act(sum_k W[o][k] concat(a, b)[pt][k] + bias[o])
int run_linear(int64_t npts, Seg a, Seg b, const LinearLayer &L, int relu, float *y, int y_stride, int y_off, hipStream_t st);
//   dw[o][i] += sum_pt g[pt][o] concat(a, b)[pt][i]            (dw: the layer's [out][in] block of a parameter-gradient blob)
int run_grad_w(int64_t npts, Seg g, Seg a, Seg b, int out, int in, float *dw, hipStream_t st);
int run_grad_b(int64_t npts, Seg g, int out, float *db, hipStream_t st);          // db[o] += sum_pt g[pt][o]
//   y[pt][k] = sum_o g[pt][o] W[o][k]
int run_backprop(int64_t npts, Seg g, const nrf_mlp *m, const LinearLayer &L, float *y, int y_stride, hipStream_t st);
//   g[pt][k] = 0 where act[pt][k] <= 0
int run_relu_mask(int64_t npts, int n, float *g, int g_stride, const float *act, int act_stride, hipStream_t st);

// the same three products for the TRAINING paths (no bit-exactness requirement): rocBLAS sgemm on the fp32 matrix cores when the library is present, else the kernels above
// (gemm_f32.hip); the weights are read from the handle's blob m->d_params
int run_linear_fast(int64_t npts, Seg a, Seg b, const nrf_mlp *m, const LinearLayer &L, int relu, float *y, int y_stride, int y_off, hipStream_t st,
                    uint64_t *relu_bits = nullptr, int relu_bits_ld = 0);          // relu_bits: where run_backprop_uses_mask_bits says so, the output's ReLU mask as bits too
int run_grad_w_fast(int64_t npts, Seg g, Seg a, Seg b, int out, int in, float *dw, hipStream_t st, int arith = 0);          // arith != 0 (train_gemm_for): gemm_tn_bf16x3
int gemm_tn_bf16x3(int64_t P, Seg g, Seg x, int out, int in, int col0, float *dw, hipStream_t st, const float *xscale = nullptr, int xscale_ld = 0);
int gemm_tn_thin(int64_t P, Seg g, Seg x, int out, int in, int col0, float *dw, hipStream_t st);          // fewer than 32 rows: fp32 FMAs, four rows per pass over x
int gemm_tn_bf16x3_2(int64_t P, Seg g, Seg x, int col0, Seg x1, int col1, int skip1, int keep1, int out, int in, float *dw, hipStream_t st, float *db = nullptr, const float *xscale = nullptr, int xscale_ld = 0);          // two X segments in one pass over G
// mask_act (optional, bf16x3 mode only -- callers test run_backprop_fuses_mask()): y = mask_act > 0 ? y : 0, the ReLU mask of the stage that consumes y
int run_backprop_fast(int64_t npts, Seg g, const nrf_mlp *m, const LinearLayer &L, float *y, int y_stride, hipStream_t st, const float *mask_act = nullptr, int mask_stride = 0,
                      const float *add = nullptr, int add_stride = 0, const uint64_t *mask_bits = nullptr, int mask_bits_ld = 0);          // add (split-precision modes only, see run_backprop_fuses_mask): y = G W (. mask) + add
bool run_backprop_fuses_mask(const nrf_mlp *m, int64_t npts);
int fp32_gemm_available();
// gemm_bf16x3.hip: the same two products (forward, back-propagation) as split-precision bf16 matrix-core GEMMs with the bias / ReLU / ReLU-mask epilogues fused
int host_pack_threads();            // threads of the host-side weight packers (a training loop re-packs every step): min(cores, 8), NRF_PACK_THREADS overrides
int train_gemm_mode();               // -1: by family (default); 0: fp32 products; 1: bf16x3; 2: f16x3 with power-of-two scaled operands (gemm_bf16x3.hip; NRF_TRAIN_GEMM, nrf_set_train_gemm)
int train_gemm_for(const nrf_mlp *m);        // the arithmetic of this network's products: the explicit mode, or by family (NeRFSmall: fp32 products; classic, LeRF: f16x3)
bool run_backprop_uses_mask_bits(const nrf_mlp *m, int64_t npts, int width);          // a layer of `width` outputs whose forward product can leave its ReLU mask as bits (and whose consumers read them)
bool gemm_nt_bits_ok(int64_t M, int N);          // ReLU masks as bits between a forward product and the back-propagation product that needs them (gemm_bf16x3.hip)
int gemm_nt_split(int arithmetic, int64_t M, int N, Seg a, Seg b, const float *B, int ldb, float *c, int ldc, const float *bias, int relu, const float *mask, int mask_ld,
                  hipStream_t st, const float *add = nullptr, int add_ld = 0, const float *r1s = nullptr, int r1s_ld = 0, const float *r1w = nullptr,
                  uint64_t *bits_out = nullptr, int bits_out_ld = 0, const uint64_t *bits_in = nullptr, int bits_in_ld = 0);
int gemm_rm(hipStream_t st, bool transA, bool transB, int64_t M, int64_t N, int64_t K, float alpha, const float *A, int lda, const float *B, int ldb, float beta, float *C, int ldc);

// matrix-core paths (separate translation units)
int mlp_small_mfma_available(const nrf_mlp *m);
int mlp_small_rescale(nrf_mlp *m, hipStream_t st);                    // mlp.hip: scales from the current blob -> d_gscale / d_scales / d_params_scaled (no-op without weight maps)
int mlp_small_forward_mfma_lm(const nrf_mlp *m, const __half2 *feats, const __half2 *feats_lo, int64_t pstride, const __half *dirs, const __half *dirs_lo, int s,
                              const uint8_t *keep, int64_t p, float *out, hipStream_t st, const int32_t *src = nullptr);
int mlp_small_color_from_geo_lm(const nrf_mlp *m, const void *geo, int64_t geo_stride, const float *sigma, const __half *dirs, const __half *dirs_lo, int s,
                                const uint8_t *keep, int64_t p, float *out, hipStream_t st);
int mlp_small_pack_sigma_f32(nrf_mlp *m, const std::vector<float> &host_params);
int mlp_small_sigma_f32_available(const nrf_mlp *m);
int mlp_small_sigma_f32_lm(const nrf_mlp *m, const void *feats, int f32_in, int64_t pstride, const uint8_t *keep, int64_t p, float *sigma, hipStream_t st, void *geo = nullptr,
                           int64_t geo_stride = 0);
int mlp_nerf_mfma_available(const nrf_mlp *m);
int mlp_nerf_forward_mfma_fused(const nrf_mlp *m, const float *pts, const float *rays, int ray_stride, const float *z, int s, const __half *dirs, int64_t p, float *out, hipStream_t st);
int launch_dirs_pe_f16(const float *rays, int stride, int64_t n, __half *out, hipStream_t st);
// NRF_PREC_F16_SPLIT for the classic NeRF (mlp_nerf_split_mfma.hip)
int mlp_nerf_split_available(const nrf_mlp *m);
int mlp_nerf_forward_split_rows(const nrf_mlp *m, const float *x, int xs, int64_t p, float *out, int os, hipStream_t st);
int mlp_nerf_forward_split_fused(const nrf_mlp *m, const float *pts, const float *rays, int ray_stride, const float *z, int s, const __half *dirs, const __half *dirs_lo,
                                 int64_t p, float *out, hipStream_t st);
int launch_dirs_pe_split(const float *rays, int stride, int64_t n, __half *out_hi, __half *out_lo, hipStream_t st);
int mlp_small_pack_f16(nrf_mlp *m, const std::vector<float> &host_params);
int mlp_small_pack_bwd(nrf_mlp *m, const std::vector<float> &host_params);
// the same images on the host (bytes), for the weight maps: false when the shape is outside the built family
bool mlp_small_images_host(const nrf_mlp_small_desc &d, const std::vector<float> &hp, std::vector<uint8_t> &f16_img, std::vector<uint8_t> &split_img);
bool mlp_small_bwd_image_host(const nrf_mlp *m, const std::vector<float> &hp, std::vector<uint8_t> &img);
bool mlp_small_sigma_image_host(const nrf_mlp_small_desc &d, const std::vector<float> &hp, std::vector<uint8_t> &head_f32, std::vector<uint8_t> &tail_f16);
size_t mlp_small_backward_mfma_workspace_bytes(const nrf_mlp *m, int64_t p);
int mlp_small_backward_mfma_lm(const nrf_mlp *m, const __half2 *feats_lm, const __half *dirs, int s_per_ray, const float *g_out, int gos, int64_t p, float *g_params,
                               float *g_x, int gxs, void *ws, size_t ws_bytes, hipStream_t st, int64_t lm_pstride = 0, const int32_t *lm_src = nullptr);
int mlp_small_backward_mfma(const nrf_mlp *m, const float *x, int xs, const float *g_out, int gos, int64_t p, float *g_params, float *g_x, int gxs, void *ws,
                            size_t ws_bytes, hipStream_t st);
int mlp_nerf_pack_f16(nrf_mlp *m, const std::vector<float> &host_params);
// the classic network's images as host vectors from (blob, merged views layer): what the packers upload and what mlp.hip's weight maps are decoded from
bool nerf_f16_images_host(const nrf_mlp_nerf_desc &d, const float *hp, const float *merged, const float *merged_b, std::vector<_Float16> &img, std::vector<_Float16> &img2,
                          std::vector<float> &bias);
bool nerf_sigma_image_host(const nrf_mlp_nerf_desc &d, const float *hp, const float *merged, const float *merged_b, std::vector<float> &img, size_t &f16_at, size_t &f16_floats);
size_t nerf_blob_offset_views();
// views_linears_0 o feature_linear of the classic network (no activation between them, NeRF.cpp:112-115), in double, rows over up to 8 host threads:
// merged[r][k] = sum_f Wv[r][f] Wf[f][k], merged_b[r] = sum_f Wv[r][f] bf[f] + bv[r]   (Wv rows are wv_stride floats apart)
void nerf_merged_views_host(const float *wv, int wv_stride, const float *wf, const float *bf, const float *bv, int rows, int w, std::vector<float> &merged, std::vector<float> &merged_b);
// f(i) for i in [0, n) on up to 8 host threads (the per-step re-pack of a training loop: nrf_mlp_set_params)
void host_parallel_for(int n, const std::function<void(int, int)> &range_fn);
// sigma_nerf_f32.hip: the classic network's density branch in exact fp32 on the matrix cores (coarse pass)
int mlp_nerf_pack_sigma_f32(nrf_mlp *m, const std::vector<float> &host_params);
int mlp_nerf_sigma_f32_available(const nrf_mlp *m);
int mlp_nerf_exact_coarse(const nrf_mlp *m, const float *pts, const float *rays, int ray_stride, const float *z, int s, const __half *dirs, const __half *dirs_lo, int64_t p,
                          float *raw, hipStream_t st);
int mlp_lerf_pack_f16(nrf_mlp *m, const std::vector<float> &host_params);
int mlp_lerf_pack_f16_device(nrf_mlp *m, hipStream_t st);                              // the same images from m->d_params, on the device
int mlp_lerf_pack_sigma_f32_device(nrf_mlp *m, hipStream_t st);                        // sigma_lerf_f32.hip
int mlp_lerf_pack_sigma_f32(nrf_mlp *m, const std::vector<float> &host_params);      // sigma_lerf_f32.hip

}  // namespace nrf
