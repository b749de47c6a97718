/*
 * nerfpp_hip.h -- C ABI of the MI355X-native (gfx950) NeRF / HashNeRF volume-rendering path.
 *
 * The reference (DeliriumV01D/NeRFpp) has no C ABI: its hot path sits behind C++ templates and
 * virtuals (BaseEmbedderImpl, BaseNeRFImpl, NeRFRenderer<...>).  This header is the boundary a
 * host binds instead: `extern "C"`, plain pointers and sizes, no torch types.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference's src/).  The
 * LibTorch adapter classes (include/nerfpp_torch.h) and the Python mirror (nerfpp_amd/) are thin
 * callers of exactly these functions.
 *
 * Conventions
 *   - every `d_` pointer is DEVICE memory (hipMalloc / a torch tensor's data_ptr on the same GPU),
 *     fp32 row-major unless stated; every other pointer is HOST memory read during the call;
 *   - every launch goes to the caller's stream (`void *stream` is a hipStream_t; NULL = the
 *     per-process default stream).  No call synchronises the device, allocates device memory
 *     (handles own theirs; scratch comes from the caller through *_workspace_bytes) or throws;
 *   - return value: NRF_OK or an NRF_ERR_* code; nrf_last_error() gives the text (thread-local);
 *   - the caller owns all buffers; handles are created/destroyed explicitly and are immutable
 *     during a call (re-entrant: no per-call state is kept on a handle, unlike
 *     CuHashEmbedderImpl::QueryPoints, CuHashEmbedder.h:26-27).
 *   - there is NO CPU fallback: without a gfx950 device every compute entry returns NRF_ERR_HIP.
 */
#ifndef NERFPP_HIP_H
#define NERFPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRF_API __attribute__((visibility("default")))

enum {
    NRF_OK = 0,
    NRF_ERR_INVALID_ARG = 1,
    NRF_ERR_HIP = 2,          /* a HIP runtime call / kernel launch failed (no device, OOM, ...) */
    NRF_ERR_UNSUPPORTED = 3,  /* valid in the reference, not built here (message says what) */
    NRF_ERR_WORKSPACE = 4,    /* caller's workspace too small */
    NRF_ERR_NONFINITE = 5     /* a matrix-core render produced inf / NaN network outputs (an fp16 operand left its range): see nrf_render_params.overflow_policy */
};

NRF_API int nrf_version(void);
NRF_API const char *nrf_last_error(void);
NRF_API const char *nrf_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * Rays                                                         RayUtils.h
 * ------------------------------------------------------------------------------------------- */

/* GetDirections + GetRays (RayUtils.h:5-46) for image rows [row0, row0+rows) of an h x w image.
 * K: host [9] row-major 3x3, c2w: host [12] row-major 3x4.  d_o, d_d: [rows*w, 3].
 * Ray r = (y-row0)*w + x (row-major, y outer) -- the reference's pixel<->ray mapping.
 * cone_angle (host, optional) receives ((1/fx + 1/fy)/2)*1.1 (RayUtils.h:35-43). */
NRF_API int nrf_get_rays(int h, int w, const float *K, const float *c2w, int row0, int rows,
                         float *d_o, float *d_d, float *cone_angle, void *stream);

/* NDCRays (RayUtils.h:49-83); in place allowed. */
NRF_API int nrf_ndc_rays(int h, int w, float focal, float near_plane, const float *d_o, const float *d_d, int64_t n,
                         float *d_o_out, float *d_d_out, void *stream);

/* ---- ray-batch producer of the training loop: NeRFDataset::get_batch / GetRayBatch (NeRFDataset.cpp:109-208), SURVEY 8f row N2 ----
 * nrf_precrop_bounds : CalculateBounds (:44-65), host only; out = {h_start, h_end, w_start, w_end}, inclusive.
 * nrf_rand_pixels    : the batch's pixel coordinates (:154-155 draws torch::randint): counter-based, element k of iteration `iter` is
 *                      lo + floor(u32(seed, stream, iter*n + k) * range / 2^32), streams 16 (rows) / 17 (columns) of include/nrf_rng.h.
 * nrf_ray_batch      : GetRayBatch (:109-145): rays through pixels (rand_h[k], rand_w[k]); cone_angle = (1/fx + 1/fy)/2 to host.
 * nrf_gather_pixels  : target_s = CurrentImage.index({rand_h, rand_w}) (:156); image [h, w, c] fp32 on the device. */
NRF_API int nrf_precrop_bounds(int h, int w, int iter, int precrop_iters, float precrop_frac, int *out);
NRF_API int nrf_rand_pixels(uint64_t seed, int64_t iter, int h_start, int h_end, int w_start, int w_end, int64_t n, int64_t *d_rand_h,
                            int64_t *d_rand_w, void *stream);
NRF_API int nrf_ray_batch(const float *K, const float *c2w, const int64_t *d_rand_h, const int64_t *d_rand_w, int64_t n, float *d_o, float *d_d,
                          float *cone_angle, void *stream);
NRF_API int nrf_gather_pixels(const float *d_image, int h, int w, int c, const int64_t *d_rand_h, const int64_t *d_rand_w, int64_t n, float *d_out,
                              void *stream);

/* IntersectWithAABB (RayUtils.h:87-126). bbox: host [6] = min xyz, max xyz. */
NRF_API int nrf_aabb(const float *d_o, const float *d_d, const float *bbox, int64_t n, float near_plane,
                     float *d_near, float *d_far, void *stream);

/* The ray-batch assembly of NeRFRenderer::Render (NeRFRenderer.h:549-583):
 * viewdirs = d/||d||, near/far from the AABB, rays = cat[o, d, near, far, (viewdirs)].
 * d_rays: [n, 11] when use_viewdirs else [n, 8]. */
NRF_API int nrf_pack_rays(const float *d_o, const float *d_d, const float *bbox, int64_t n, int use_viewdirs,
                          float *d_rays, void *stream);

/* ... with the view directions taken from d_view_src [n,3] instead of d_d: Render() normalises rays_d into viewdirs BEFORE the NDC warp replaces
 * rays_o / rays_d (NeRFRenderer.h:549-568), so an Ndc + UseViewdirs batch packs the warped o, d with the un-warped directions.  d_rays: [n, 11]. */
NRF_API int nrf_pack_rays_viewsrc(const float *d_o, const float *d_d, const float *d_view_src, const float *bbox, int64_t n, float *d_rays, void *stream);

/* The pose branch of NeRFRenderer::Render (NeRFRenderer.h:541-583) for the row tile [row0, row0 + rows) of an h x w frame, as ONE kernel:
 * GetRays(h, w, K, c2w) -> view directions d/||d|| (from c2w's rays, before c2w_staticcam replaces the camera, :549-561, and before the NDC warp) ->
 * NDCRays(h, w, K[0], 1.f) when `ndc` (:563-568) -> IntersectWithAABB -> rays_ = cat[o, d, near, far, (viewdirs)].  Zero-initialise the struct.
 * `chunk` is BatchifyRays' Chunk (used by nrf_render_rows; results do not depend on it). */
typedef struct nrf_view {
    int h, w;                 /* frame size (after any RenderFactor, nrf_render_view_dims) */
    float K[9];               /* row-major 3x3 */
    float c2w[12];            /* row-major 3x4 */
    int has_staticcam;        /* c2w_staticcam given (honoured only with use_viewdirs, as in the reference) */
    float c2w_staticcam[12];
    int row0, rows;           /* the tile; ray r of the tile is pixel (row0 + r / w, r % w) */
    int use_viewdirs;         /* UseViewdirs: ray stride 11, else 8 */
    int ndc;                  /* Ndc */
    int chunk;                /* Chunk */
    float bbox[6];            /* BoundingBox: min xyz, max xyz */
} nrf_view;
/* d_rays: [rows*w, 8|11].  d_near_far (optional): DEVICE [2] floats receiving min(near), max(far) of the tile (NeRFRenderer.h:602-603) --
 * written on `stream`, no host synchronisation (the reference's two .item() calls stall the host once per frame). */
NRF_API int nrf_view_rays(const nrf_view *v, float *d_rays, float *d_near_far, void *stream);

/* min(near), max(far) over a packed ray batch (NeRFRenderer.h:602-603); results to host, synchronises `stream`. */
NRF_API int nrf_near_far_range(const float *d_rays, int64_t n, int ray_stride, float *near_min, float *far_max, void *stream);
/* ... the same into DEVICE memory (d_near_far [2] floats), in stream order, nothing waits: what a training loop wants (the reference's two .item() calls of
 * NeRFRenderer.h:602-603 stall its host once per rendered batch; the values are only read when a frame is post-processed). */
NRF_API int nrf_near_far_range_device(const float *d_rays, int64_t n, int ray_stride, float *d_near_far, void *stream);

/* torch::linspace(start, end, steps) as ATen's CPU kernel rounds it (one fused rounding per
 * element) -- for hosts without torch; the LibTorch / PyTorch callers pass torch::linspace itself. */
NRF_API int nrf_linspace(float start, float end, int steps, float *out_host);

/* z_vals (NeRFRenderer.h:393-402) and sample points pts = o + d*z (:419). d_t: [s] = linspace(0,1,s). */
NRF_API int nrf_z_vals(const float *d_rays, int ray_stride, int64_t n, const float *d_t, int s, int lindisp,
                       float *d_z, void *stream);
NRF_API int nrf_points(const float *d_rays, int ray_stride, const float *d_z, int64_t n, int s, float *d_pts, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Encoders                                                     BaseEmbedder.h:6-15 (plugin iface)
 * ------------------------------------------------------------------------------------------- */

/* EmbedderImpl::forward, sinusoidal PE (NeRF.cpp:4-39): [p,3] -> [p, 3+6*nfreq]. */
NRF_API int nrf_pe_encode(const float *d_x, int64_t p, int nfreq, float *d_out, void *stream);

/* Spherical harmonics: variant 0 = SHEncoderImpl (NeRF.cpp:131-201, degree 1..5),
 *                      variant 1 = CuSHEncoderImpl / CuSHKernel (CuSHEncoder.cu:4-118, degree 1..8).
 * [p,3] -> [p, degree^2]. */
enum { NRF_SH_LIBTORCH = 0, NRF_SH_CUDA = 1 };
NRF_API int nrf_sh_encode(const float *d_dirs, int64_t p, int degree, int variant, float *d_out, void *stream);

/* Multiresolution hash grid.
 *   NRF_HASH_NGP : HashEmbedderImpl  (NeRF.cpp:208-318)  fp32 table [L][2^T][F], int64 hash with the
 *                  Instant-NGP primes, floor()ed per-level resolution.
 *   NRF_HASH_CU  : CuHashEmbedderImpl (CuHashEmbedder.cpp:85-103 + CuHashEmbedder.cu:8-102) fp16 table
 *                  [L*2^T, F] (cast ONCE at upload, not per forward as .cu:257 does), uint32 hash with
 *                  per-level primes, un-floored scale, the level-offset overlap quirk (.cu:54),
 *                  blend in fp32 rounded once to fp16 (.cu:95). */
enum { NRF_HASH_NGP = 0, NRF_HASH_CU = 1 };

typedef struct nrf_hash_desc {
    int mode;                 /* NRF_HASH_NGP | NRF_HASH_CU */
    int n_levels;             /* L  (ctor arg n_levels, NeRFExecutor.h:430-432) */
    int n_features;           /* F  */
    int log2_hashmap_size;    /* T  */
    int base_resolution;
    int finest_resolution;
    float bbox[6];            /* min xyz, max xyz */
} nrf_hash_desc;

typedef struct nrf_hash nrf_hash;

NRF_API int nrf_hash_create(const nrf_hash_desc *desc, nrf_hash **out);
NRF_API void nrf_hash_destroy(nrf_hash *h);
NRF_API int nrf_hash_output_dims(const nrf_hash *h);              /* GetOutputDims() */
NRF_API int64_t nrf_hash_table_elems(const nrf_hash *h);          /* L * 2^T * F */

/* Upload the embedding table from an fp32 array in the reference's parameter layout
 * (NGP: embeddings_0..L-1 concatenated, each [2^T, F], NeRF.cpp:255-258;  CU: `embedder_embeddings`
 * [L*2^T, F], CuHashEmbedder.cpp:24).  `src` may be host or device memory (src_on_device).
 * STREAM ORDER: the upload, the fp32 -> fp16 cast of the CU mode and the re-bake of the dense image (see nrf_hash_set_dense_budget) are enqueued on `stream`
 * and NOT waited for (a training loop calls this every step) -- except when the dense image is (re)allocated, which completes before the call returns.  Encode /
 * render calls must therefore be issued on the SAME stream, or on one ordered behind it (an event recorded on `stream` after this call, or a synchronisation);
 * a render on an unrelated stream races with the cast and the bake.  (nrf_batchify_rays' lanes fork from the caller's stream and are ordered behind it.) */
NRF_API int nrf_hash_set_table(nrf_hash *h, const float *src, int src_on_device, void *stream);

/* NRF_HASH_CU only: per-level primes [L*3] (buffer `embedder_primes`, CuHashEmbedder.cpp:51-52) and
 * biases [L*3] (`embedder_biases`, :54-59; NULL = zeros).  Host pointers.  A zero or even multiplier is rejected (NRF_ERR_INVALID_ARG): the reference
 * only ever produces odd primes in [2^28, 2^30), and zeros -- an uninitialised buffer -- would hash every corner to row 0 without any error. */
NRF_API int nrf_hash_set_primes(nrf_hash *h, const int32_t *primes, const float *biases);

/* The renderer's fast path reads a DENSE image of the grid's coarse levels (every lattice vertex's table entries copied next to each other, baked from the
 * table at upload; outputs identical to the hashed lookup): `budget_bytes` of it are built, coarse to fine, default 24 GiB of the 288 GB (all 16 levels at
 * finest 512 take 4.4 GiB).  A training loop re-uploads the table every step and sets 0 (no bake); a renderer leaves the default.  Re-bakes immediately when a
 * table is present.  STREAM ORDER as nrf_hash_set_table: when the image is (re)allocated or released the call completes the bake / drains `stream` before it
 * returns; a re-bake into the existing image stays asynchronous on `stream`, and its readers must be ordered behind it there. */
NRF_API int nrf_hash_set_dense_budget(nrf_hash *h, int64_t budget_bytes, void *stream);
NRF_API int64_t nrf_hash_get_dense_budget(const nrf_hash *h);
/* Device memory the handle holds: the table in the mode's own type (fp16 CuHashEmbedder / fp32 HashEmbedder) and the baked image of the render fast path (dense coarse
 * levels + the level-major copy of the hashed ones; 0 before the first upload).  Replicated on every rank of a sharded render. */
NRF_API int nrf_hash_memory_bytes(const nrf_hash *h, int64_t *table_bytes, int64_t *baked_bytes);

/* CuHashEmbedder mode: the per-level position scales mul_l (host arrays of n_levels floats).  The reference computes them ON THE DEVICE, per thread, as
 * exp2f((log2f(finest) - log2f(base)) * l / (L - 1) + log2f(base)) (CuHashEmbedder.cu:40); nrf_hash_create evaluates the same expression with the host's
 * libm.  CUDA's exp2f / log2f are not libm's, and the features are fp16-ROUNDED blends: one ulp of mul_l flips the fp16 rounding of ~10 % of the features
 * of the Lego-sized grid and moves a rendered pixel by a few 1e-4 (tests/test_oracle_golden.py, sensitivity study).  A host that needs parity with a
 * particular CUDA build at the 1e-4 level reads the 16 values once from that build and sets them here; everything downstream then sees its voxels. */
NRF_API int nrf_hash_get_level_scales(const nrf_hash *h, float *scales_out);
NRF_API int nrf_hash_set_level_scales(nrf_hash *h, const float *scales, void *stream);

/* BaseEmbedderImpl::forward for the hash grid: x [p,3] -> (embedding [p, L*F] fp32, keep_mask [p] u8).
 * d_keep_mask may be NULL. */
NRF_API int nrf_hash_encode(const nrf_hash *h, const float *d_x, int64_t p, float *d_out, uint8_t *d_keep_mask, void *stream);
/* CuHashEmbedder mode: the same features level-major in fp16, d_feats [n_levels][p][n_features] halfs (16-byte aligned) -- the values the row-major call
 * returns (they are fp16-rounded there too, CuHashEmbedder.cu:95) in the layout a matrix-core consumer loads as operand fragments. */
NRF_API int nrf_hash_encode_lm_f16(const nrf_hash *h, const float *d_x, int64_t p, void *d_feats, uint8_t *d_keep_mask, void *stream);
/* ... into p columns of a wider table: level l of point i at d_feats + (l * pstride + i) * n_features halfs (d_feats = the first column to write, aligned to one
 * point's n_features * 2 bytes).  A renderer that keeps the coarse pass's columns next to the fine pass's new ones (see nrf_fine_depths_merge) encodes each once. */
NRF_API int nrf_hash_encode_lm_f16_strided(const nrf_hash *h, const float *d_x, int64_t p, void *d_feats, int64_t pstride, uint8_t *d_keep_mask, void *stream);

/* ---------------------------------------------------------------------------------------------
 * MLPs                                                         BaseNeRFImpl::forward (NeRF.h:33-42)
 * Parameters: ONE fp32 blob in the reference's named_parameters() == checkpoint order
 * (NeRFExecutor.h:1055-1070), Linear weights [out, in] row-major.
 * ------------------------------------------------------------------------------------------- */
enum {
    NRF_PREC_F32 = 0,      /* fp32 FMA chains in ascending k: the parity mode (== oracle bit for bit) */
    NRF_PREC_F16_MFMA = 1, /* fp16 operands on the matrix cores, fp32 accumulate: the fast mode */
    NRF_PREC_F16_SPLIT = 2 /* matrix cores with every operand carried as hi + lo fp16 pairs (22 significant bits, three MFMAs per
                              product): fp32-grade results at matrix-core speed.  Built for NeRFSmall, the classic 8 x 256 NeRF and
                              the fused LeRF passes (nrf_lerf_set_precision). */
};

typedef struct nrf_mlp_small_desc {      /* NeRFSmallImpl ctor (NeRF.cpp:322-360) */
    int input_ch, input_ch_views;
    int num_layers, hidden_dim, geo_feat_dim;
    int num_layers_color, hidden_dim_color;
    /* the predicted-normals head (NeRF.cpp:343-347, :393-407; the executor builds it only when n_importance == 0 && use_pred_normal, NeRFExecutor.h:487): a third
     * bias-free net on cat[sigma, geo_feat, input_pts] -> 3, appended to the output: [rgb, sigma, normal xyz] = 7 columns.  NRF_PREC_F32 only (the matrix-core
     * precisions answer NRF_ERR_UNSUPPORTED for such a handle); RawToOutputs ignores the extra columns (NeRFRenderer.h:279).  Zero-initialised fields = no head. */
    int use_pred_normal, num_layers_normals, hidden_dim_normals;
} nrf_mlp_small_desc;

typedef struct nrf_mlp_nerf_desc {       /* NeRFImpl ctor (NeRF.cpp:41-90) */
    int depth, width, input_ch, input_ch_views, output_ch, skip, use_viewdirs;
} nrf_mlp_nerf_desc;

typedef struct nrf_mlp nrf_mlp;

NRF_API int64_t nrf_mlp_small_param_count(const nrf_mlp_small_desc *d);
NRF_API int64_t nrf_mlp_nerf_param_count(const nrf_mlp_nerf_desc *d);
/* LeRFImpl (LeRF.cpp:3-26) described with nrf_mlp_small_desc: input_ch = input_ch_le, num_layers = num_layers_le, hidden_dim =
 * hidden_dim_le, geo_feat_dim = geo_feat_dim_le, hidden_dim_color = lang_embed_dim (other fields ignored). Output [p, lang_embed_dim + 1]. */
NRF_API int64_t nrf_mlp_lerf_param_count(const nrf_mlp_small_desc *d);
NRF_API int nrf_mlp_lerf_create(const nrf_mlp_small_desc *d, const float *params, int params_on_device, void *stream, nrf_mlp **out);
NRF_API int nrf_mlp_small_create(const nrf_mlp_small_desc *d, const float *params, int params_on_device, void *stream, nrf_mlp **out);
NRF_API int nrf_mlp_nerf_create(const nrf_mlp_nerf_desc *d, const float *params, int params_on_device, void *stream, nrf_mlp **out);
NRF_API void nrf_mlp_destroy(nrf_mlp *m);
NRF_API int nrf_mlp_output_dims(const nrf_mlp *m);
/* forward: x [p, input_ch + input_ch_views] -> out [p, output_dims] */
NRF_API int nrf_mlp_forward(const nrf_mlp *m, const float *d_x, int64_t p, int precision, float *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Compositing and hierarchical sampling
 * ------------------------------------------------------------------------------------------- */

/* NeRFRenderer::RawToOutputs (NeRFRenderer.h:199-282) + TruncExp::forward (CustomOps.cpp:5-9).
 * raw [n,s,c] (rgb at 0..2, sigma at 3), z [n,s], d [n,3] (stride d_stride floats).
 * Any output pointer may be NULL. */
NRF_API int nrf_raw2outputs(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c,
                            int white_bkgr, float *d_rgb, float *d_disp, float *d_acc, float *d_weights, float *d_depth,
                            void *stream);

/* LeRF: the weights / depth part of LeRFRenderer::RawToLEOutputs (LeRFRenderer.cpp:27-76): the same sigma -> alpha -> weights
 * arithmetic as RawToOutputs with sigma at channel `sigma_ch` of a c-wide raw tensor and no colour. */
NRF_API int nrf_raw2weights(const float *d_raw, int c, int sigma_ch, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s,
                            float *d_weights, float *d_depth, float *d_disp, float *d_acc, void *stream);

/* The same with sample (ray, j)'s raw row read at row d_src[ray * s + j] (int32) of d_raw: the feature-reusing LeRF pass keeps sigma_le in feature-COLUMN order
 * (coarse columns, then the new samples') and composes through the merge map of nrf_fine_depths_merge instead of gathering first. */
NRF_API int nrf_raw2weights_gather(const float *d_raw, int c, int sigma_ch, const int32_t *d_src, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s,
                                   float *d_weights, float *d_depth, float *d_disp, float *d_acc, void *stream);

/* The LeRF head fused with its render pass on the matrix cores (fp16 operands, fp32 accumulate), for the reference's LeRF shape
 * (main.cpp:203-213: in 128, hidden 256, 2 + 2 layers, geo 32, embedding 768) -- the [N, S, 769] raw tensor is never formed.
 *   nrf_lerf_sigma            : sigma_le = LeRFImpl::forward(x)[..., -1] (LeRF.cpp:86-95), zeroed where keep is false (LeRFRenderer.cpp:22-23)
 *   nrf_lerf_render_embedding : out[n, 768] = sum_s weights[n,s] * normalize(le(x[n,s]))  (LeRF.cpp:96-108 + the sum of LeRFRenderer.h:45-54);
 *                               s must be a multiple of 32.  L2-normalise `out` afterwards (nrf_render_clip_embedding's last step).
 *                               Evaluated as W . sum_s (weights / ||W a||) a with ||W a||^2 = a^T (W^T W) a (the output layer is bias-free, hence linear): the
 *                               256 -> 768 layer runs once per ray.  Takes n * 256 floats of stream-ordered scratch (hipMallocAsync / hipFreeAsync). */
NRF_API int nrf_lerf_mfma_available(const nrf_mlp *m);
/* Arithmetic of the four fused entries below, per handle: NRF_PREC_F16_MFMA (default: fp16 operands, fp32 accumulate) or NRF_PREC_F16_SPLIT (hi + lo fp16
 * operand pairs, three products: fp32-grade, as LeRFImpl::forward computes -- mlp_lerf_split_mfma.hip).  Call between, not during, passes. */
NRF_API int nrf_lerf_set_precision(nrf_mlp *m, int precision);
NRF_API int nrf_lerf_sigma(const nrf_mlp *m, const float *d_x, const uint8_t *d_keep, int64_t p, float *d_sigma, void *stream);
NRF_API int nrf_lerf_render_embedding(const nrf_mlp *m, const float *d_x, const float *d_weights, int64_t n, int s, float *d_out, void *stream);
/* ... reading the level-major fp16 features of nrf_hash_encode_lm_f16 (16 levels x 8 features: [16][p][8] halfs) instead of fp32 rows. */
NRF_API int nrf_lerf_sigma_lm(const nrf_mlp *m, const void *d_feats_lm, const uint8_t *d_keep, int64_t p, float *d_sigma, void *stream);
NRF_API int nrf_lerf_render_embedding_lm(const nrf_mlp *m, const void *d_feats_lm, const float *d_weights, int64_t n, int s, float *d_out, void *stream);
/* ... on columns of a wider level-major table ([16][pstride][8] halfs): _strided evaluates p consecutive columns starting at d_feats_lm; _gather reads column
 * d_src[i] for sample i of the n * s sorted depths (d_src = the merge map of nrf_fine_depths_merge; NULL = column i).  Same arithmetic, same results. */
NRF_API int nrf_lerf_sigma_lm_strided(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const uint8_t *d_keep, int64_t p, float *d_sigma, void *stream);
/* Split precision only: kernel A also leaves the sigma net's second output (sigma, geo32: LE0's chained operand, 192 bytes per column as fragment planes,
 * nrf_lerf_geo_bytes(columns)) in d_geo, and the embedding pass starts at LE0 from it instead of re-evaluating the sigma net (224 of its 832 matrix
 * instructions per 32 points).  d_geo / d_feats_lm of a _strided call point at the call's first column; geo_stride = columns of the whole table.
 * Same values in the same order as the recomputation: results unchanged.  NRF_ERR_UNSUPPORTED in NRF_PREC_F16_MFMA. */
NRF_API size_t nrf_lerf_geo_bytes(int64_t columns);
NRF_API int nrf_lerf_sigma_geo_lm_strided(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const uint8_t *d_keep, int64_t p, float *d_sigma,
                                          void *d_geo, int64_t geo_stride, void *stream);
/* The LeRF density net in EXACT fp32 on the matrix cores (sigma_lerf_f32.hip): d_sigma equals nrf_mlp_forward(..., NRF_PREC_F32)[..., -1] -- and the CPU oracle --
 * bit for bit (v_mfma_f32_32x32x2_f32 == the ascending-k fmaf chain), at ~0.8 of the 157 TFLOP/s fp32 matrix peak.  The coarse pass of a hierarchical LeRF render
 * consumes nothing but sigma_le (LeRFRenderer.cpp:139-170) and its weights choose the fine samples through a discontinuous function, so the split-precision render
 * runs its coarse pass through this entry: the fine sample set is then the fp32 path's own.  d_geo (optional): also leaves (sigma, geo32) as the (hi, lo) operand
 * planes of nrf_lerf_render_embedding_lm_geo, split from the exact values.  Level-major CuHashEmbedder features, same column conventions as the _strided entries. */
NRF_API int nrf_lerf_sigma_exact_available(const nrf_mlp *m);
NRF_API int nrf_lerf_sigma_exact_lm_strided(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const uint8_t *d_keep, int64_t p, float *d_sigma,
                                            void *d_geo, int64_t geo_stride, void *stream);
NRF_API int nrf_lerf_render_embedding_lm_geo(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const int32_t *d_src, const void *d_geo,
                                             int64_t geo_stride, const float *d_weights, int64_t n, int s, float *d_out, void *stream);
NRF_API int nrf_lerf_render_embedding_lm_gather(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const int32_t *d_src, const float *d_weights, int64_t n,
                                                int s, float *d_out, void *stream);

/* RenderCLIPEmbedding (LeRFRenderer.h:45-54): out[n, embed_dim] = normalize(sum_s weights[n,s] * embeds[n,s,:embed_dim], eps 1e-8).
 * embeds rows are embed_stride floats apart (the raw LeRF output is [n,s,embed_dim+1]). */
NRF_API int nrf_render_clip_embedding(const float *d_embeds, int embed_stride, int embed_dim, const float *d_weights, int64_t n, int s,
                                      float *d_out, void *stream);

/* Relevancy(embeds, positives, negatives) (call sites LeRFRenderer.cpp:79, NeRFExecutor.h:824): [n, embed_dim] L2-normalised embeddings against
 * [n_pos, embed_dim] positive and [n_neg, embed_dim] negative ("canonical") phrase embeddings -> d_out [n, 2] = (p_positive, p_negative) of the pairwise softmax
 * (temperature 10) against the negative phrase the positive does worst against; column 0 is what the hosts consume (LeRFRenderer.h:18, NeRFExecutor.h:714).
 * PARITY UNPINNED: the function's source is external (DeliriumV01D/RuCLIP, RuCLIPProcessor.h, no pinned version, absent from the reference tree) -- this is the
 * published LERF relevancy score (Kerr et al. 2023, section 3.3; nerfstudio `get_relevancy`) that it mirrors, restated in oracle/nerf_oracle.c (orc_relevancy) and
 * checked against that and against known answers.  positive_id selects the row of d_positives (the reference passes one positive phrase: 0). */
NRF_API int nrf_lerf_relevancy(const float *d_embeds, int64_t n, int embed_dim, const float *d_positives, int n_pos, const float *d_negatives, int n_neg,
                               int positive_id, float *d_out, void *stream);

/* The relevancy image of RenderPath (NeRFExecutor.h:713-719): rel[..., 0].mul(255).to(kU8) -> cv::applyColorMap(COLORMAP_JET) -> d_bgr [n, 3] bytes in OpenCV's
 * B, G, R order.  d_relevancy rows are rel_stride floats apart (2 for nrf_lerf_relevancy's output).  nrf_colormap_jet_u8 maps bytes that already are the image
 * (the training-time preview, NeRFExecutor.h:825-831); nrf_colormap_jet_lut writes the 256 x 3 table to HOST memory.
 * PARITY UNPINNED: OpenCV is not in this image; the table is restated from OpenCV's published colormap (oracle/nerf_oracle.c, orc_colormap_jet_lut, says how). */
NRF_API int nrf_relevancy_image(const float *d_relevancy, int64_t n, int rel_stride, uint8_t *d_bgr, void *stream);
NRF_API int nrf_colormap_jet_u8(const uint8_t *d_gray, int64_t n, uint8_t *d_bgr, void *stream);
NRF_API int nrf_colormap_jet_lut(uint8_t *lut_host /*[256*3]*/);

/* SamplePDF, deterministic branch (Sampler.h:6-43).  bins [n,nb], weights [n,nb-1], u [ns] device
 * (= linspace(0,1,ns)).  sum_vec: fp32 lanes of the host whose torch::sum order is reproduced for the
 * pdf normaliser (8 = any AVX2+/AVX-512 x86 build of ATen; 0 = order-free double accumulation).
 * d_inds (optional): the searchsorted indices, int64 like the reference's. */
NRF_API int nrf_sample_pdf(const float *d_bins, const float *d_weights, int64_t n, int nb, const float *d_u, int ns, int sum_vec,
                           float *d_samples, int64_t *d_inds, void *stream);

/* The fine-pass depth set of RenderRays (NeRFRenderer.h:427-431): z_mid, SamplePDF on weights[1:-1],
 * sort(cat(z, samples)).  z [n,s], weights [n,s] -> z_fine [n, s+ns]. */
NRF_API int nrf_fine_depths(const float *d_z, const float *d_weights, int64_t n, int s, const float *d_u, int ns, int sum_vec,
                            float *d_z_fine, void *stream);
/* ... and where each sorted depth came from.  s of the s + ns depths of a ray ARE its coarse depths (same z, hence the same sample point and the same
 * encoder features, bit for bit), so a fine pass need only encode the ns new ones:
 *   d_src [n, s+ns] int32 : column of sorted depth (ray, i) in a table holding the coarse pass's n*s points first (ray-major) and the n*ns new points after
 *                           them: ray*s + j for coarse depth j, n*s + ray*ns + j for new sample j
 *   d_z_new [n, ns]       : the new samples' depths in SamplePDF order (ascending)
 * nrf_render_rays does this internally; the entry serves hosts that drive the passes themselves (the LeRF render pass).  n (s + ns) < 2^31. */
NRF_API int nrf_fine_depths_merge(const float *d_z, const float *d_weights, int64_t n, int s, const float *d_u, int ns, int sum_vec,
                                  float *d_z_fine, int32_t *d_src, float *d_z_new, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Stochastic branches of RenderRays (Perturb > 0, cone rays, training-time noise).  The stage functions take the random
 * draws as explicit device arrays -- hand them the reference's own torch::rand / randn tensors and the results can be
 * compared value for value.  nrf_render_rays itself generates draws in-kernel from (seed, stream, GLOBAL element index)
 * with include/nrf_rng.h (streams NRF_RNG_*), so a render does not depend on Chunk or on the ray sharding;
 * nrf_rng_fill materialises the same draws: element k = draw(seed, rng_stream, index0 + k), uniform [0,1) or normal.
 * ------------------------------------------------------------------------------------------- */
NRF_API int nrf_rng_fill(uint64_t seed, uint32_t rng_stream, uint64_t index0, int64_t count, int normal, float *d_out, void *stream);

/* Stratified jitter (NeRFRenderer.h:404-417): z [n,s], t_rand [n,s] uniform -> out [n,s] (must not alias z). */
NRF_API int nrf_jitter_z(const float *d_z, const float *d_t_rand, int64_t n, int s, float *d_out, void *stream);

/* TangentScatter (NeRFRenderer.h:307-362).  d_pts [n,s,3] or NULL (= o + d*z from the packed rays); rays_d is read from the
 * packed rays (columns 3..5); u_r / u_theta [n,s] uniform draws; bbox: host [6] or NULL (no clamp). */
NRF_API int nrf_tangent_scatter(const float *d_pts, const float *d_rays, int ray_stride, const float *d_z, int64_t n, int s, float cone_angle,
                                const float *d_u_r, const float *d_u_theta, const float *bbox, float *d_out, void *stream);

/* Stochastic preconditioning + ReflectBoundary (NeRFRenderer.h:433-443, :285-304): pts [p,3] + noise [p,3]*alpha, reflected. */
NRF_API int nrf_precondition(const float *d_pts, const float *d_noise, float alpha, const float *bbox, int64_t p, float *d_out, void *stream);

/* RawToOutputs with raw_noise_std > 0 (NeRFRenderer.h:251-252): noise [n,s] normal draws. */
NRF_API int nrf_raw2outputs_noise(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c, int white_bkgr,
                                  const float *d_noise, float noise_std, float *d_rgb, float *d_disp, float *d_acc, float *d_weights,
                                  float *d_depth, void *stream);

/* SamplePDF with det = false (Sampler.h:22-24) and the fine depth set built on it: u is [n, ns], one unsorted row per ray. */
NRF_API int nrf_sample_pdf_rand(const float *d_bins, const float *d_weights, int64_t n, int nb, const float *d_u, int ns, int sum_vec,
                                float *d_samples, int64_t *d_inds, void *stream);
NRF_API int nrf_fine_depths_rand(const float *d_z, const float *d_weights, int64_t n, int s, const float *d_u, int ns, int sum_vec,
                                 float *d_z_fine, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Renderer                                    NeRFRenderer<TEmbedder,TEmbedDirs,TNeRF> (NeRFRenderer.h:88-159)
 * ------------------------------------------------------------------------------------------- */
enum { NRF_DIRS_NONE = 0, NRF_DIRS_PE = 1, NRF_DIRS_SH_LIBTORCH = 2, NRF_DIRS_SH_CUDA = 3 };

typedef struct nrf_renderer_desc {
    const nrf_hash *hash;     /* position encoder: hash grid, or NULL for sinusoidal PE */
    int pe_freqs;             /* used when hash == NULL (Embedder multires, NeRFExecutor.h:427) */
    int dirs_encoder;         /* NRF_DIRS_* */
    int dirs_param;           /* PE: multires_views;  SH: degree */
    const nrf_mlp *mlp;
} nrf_renderer_desc;

/* NeRFRenderParams (NeRFRenderer.h:28-44) as RenderRays consumes them.  Zero-initialise the struct: all stochastic
 * fields 0 is the deterministic render path (Perturb = 0, RawNoiseStd = 0, ThinRay = true -- what FillRenderParams sets
 * at test time, NeRFExecutor.h:379-415, plus ThinRay). */
/* The coarse pass of a hierarchical render contributes only its compositing weights (NeRFRenderer.h:422-428), i.e. sigma, and those
 * weights choose the fine samples through searchsorted -- a discontinuous function.  NRF_COARSE_AUTO: with NRF_PREC_F16_SPLIT on the
 * HashNeRF fast path the coarse pass evaluates the sigma net ONLY, in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32 ==
 * the ascending-k fma chain of NRF_PREC_F32, bit for bit), so the fine sample set equals the parity mode's; that kernel also hands the
 * sigma net's output (sigma, geo_feat) of the coarse points to the fine pass, which evaluates the colour net alone at its n_samples coarse
 * depths (their sigma is then NRF_PREC_F32's bit for bit) and the whole network at the n_importance new ones; every other case runs
 * the whole network in `precision`.  NRF_COARSE_FULL forces the latter, NRF_COARSE_SIGMA_F32 asks for the former in
 * NRF_PREC_F16_MFMA too.  A caller that wants d_raw_coarse always gets the whole network. */
enum { NRF_COARSE_AUTO = 0, NRF_COARSE_FULL = 1, NRF_COARSE_SIGMA_F32 = 2 };

typedef struct nrf_render_params {
    int n_samples;            /* NSamples */
    int n_importance;         /* NImportance */
    int lindisp;              /* LinDisp */
    int white_bkgr;           /* WhiteBkgr */
    int precision;            /* NRF_PREC_* for the MLP */
    int sum_vec;              /* see nrf_sample_pdf */
    /* stochastic branches; draws come from include/nrf_rng.h keyed by (seed, stream, (ray_base + ray)*S + sample) */
    float perturb;            /* Perturb > 0: stratified jitter + SamplePDF(det = false) */
    int has_cone;             /* !ThinRay: TangentScatter on both passes with `cone_angle` (GetRays, RayUtils.h:43-44) */
    float cone_angle;
    float raw_noise_std;      /* RawNoiseStd */
    float precond_alpha;      /* StochasticPreconditioningAlpha (fine pass only, as in the reference) */
    int has_bbox;             /* BoundingBox given: TangentScatter clamps to it; required by precond_alpha > 0 */
    float bbox[6];
    uint64_t seed;
    int64_t ray_base;         /* index of d_rays[0] within the whole image / ray batch */
    int coarse_mode;          /* NRF_COARSE_*: how the coarse pass is evaluated when n_importance > 0 (it only supplies SamplePDF's weights) */
    int overflow_policy;      /* NRF_OVERFLOW_*: what happens when a matrix-core precision's network outputs are not finite (below) */
} nrf_render_params;

/* The matrix-core precisions carry operands in fp16 (NRF_PREC_F16_MFMA) or as fp16 (hi, lo) pairs (NRF_PREC_F16_SPLIT): an activation beyond 65 504 becomes an inf there.
 * NeRFSmall's split image is range-scaled against that (nrf_mlp_set_input_rms_hint); whatever still overflows -- or any other family's network -- shows as inf / NaN network
 * outputs, which the final compositing kernel of every chunk records in a per-chunk word of the renderer (4 FMAs per sample).  What the render entries do with it:
 *   NRF_OVERFLOW_AUTO / _RERENDER  after the Chunk loop the words are read back (ONE host synchronisation at the end of nrf_render_rays / nrf_batchify_rays /
 *                                  nrf_render_rows) and every flagged chunk is rendered again in NRF_PREC_F32 into the same outputs: the call's results are finite-input
 *                                  correct whatever the weights.  The default.
 *   NRF_OVERFLOW_ERROR             same read-back; a flagged chunk makes the call return NRF_ERR_NONFINITE (outputs of that chunk are not to be used).
 *   NRF_OVERFLOW_DEFERRED          no synchronisation: the words are copied to pinned host memory behind the call's work and looked at by the first LATER render call on
 *                                  this renderer that finds the copy complete, which then returns NRF_ERR_NONFINITE before doing anything (or by nrf_renderer_nonfinite).
 *                                  Up to 32 calls' words may be in flight; the 33rd call waits for the oldest copy.  For pipelines that keep several frames in flight.
 *   NRF_OVERFLOW_IGNORE            no detection at all (the compositing kernel skips the test). */
enum { NRF_OVERFLOW_AUTO = 0, NRF_OVERFLOW_RERENDER = 1, NRF_OVERFLOW_ERROR = 2, NRF_OVERFLOW_DEFERRED = 3, NRF_OVERFLOW_IGNORE = 4 };

typedef struct nrf_render_outputs {   /* NeRFRendererOutputs / NeRFRenderResult (NeRFRenderer.h:12-26); NULL = not wanted */
    float *d_rgb;             /* [n,3] */
    float *d_disp;            /* [n]   */
    float *d_acc;             /* [n]   */
    float *d_depth;           /* [n]   */
    float *d_weights;         /* [n, S_out]  S_out = n_samples + n_importance (or n_samples if n_importance == 0) */
    float *d_raw;             /* [n, S_out, 4] in depth order.  The fast paths keep the network outputs where they were computed (coarse depths | new samples) and the
                               * compositing kernel reads through the merge map; asking for d_raw adds one gather pass (16 B read + written per sample) */
    /* intermediates for stage-chained parity tests (optional) */
    float *d_z_coarse;        /* [n, n_samples] */
    float *d_raw_coarse;      /* [n, n_samples, 4] */
    float *d_weights_coarse;  /* [n, n_samples] */
    float *d_z_fine;          /* [n, n_samples + n_importance] */
} nrf_render_outputs;

typedef struct nrf_renderer nrf_renderer;

NRF_API int nrf_renderer_create(const nrf_renderer_desc *desc, nrf_renderer **out);
NRF_API void nrf_renderer_destroy(nrf_renderer *r);
/* Totals since the renderer was created: chunks whose matrix-core render produced non-finite network outputs, and how many of them were rendered again in NRF_PREC_F32
 * (NRF_OVERFLOW_RERENDER).  Completes a pending NRF_OVERFLOW_DEFERRED check first (waits for that call's work).  Either pointer may be NULL. */
/* Where the most recent render call on this renderer left the hash features of its fine depths -- a SINGLE-chunk call (nrf_render_rays, or nrf_batchify_rays with
 * Chunk >= n) of the feature-reusing fast path on a CuHashEmbedder grid: the level-major fp16 table [16][cols] (half2; coarse columns first, the new samples' behind
 * them), the keep mask by column, and the merge map [n, sf] (sorted depth i of the batch -> its column).  They live in the caller's workspace: valid until the next
 * render call on this renderer or any other use of that workspace.  NRF_ERR_UNSUPPORTED when the last call left none (serial is set either way).  For nrf_mlp_backward_f16_lm_src /
 * nrf_mask_sigma_grad_src: a training step's backward reads the features its own forward render encoded instead of encoding the fine points again. */
NRF_API int nrf_renderer_last_features(const nrf_renderer *r, const void **d_feats_lm, int64_t *cols, const uint8_t **d_keep_cols, const int32_t **d_src, int64_t *n, int *sf,
                                       uint64_t *serial /* optional: a count of the chunks this renderer has rendered -- unchanged between two queries = no render in between */);
NRF_API int nrf_renderer_nonfinite(const nrf_renderer *r, int64_t *flagged_chunks, int64_t *rerendered_chunks);

/* RunNetwork (NeRFRenderer.h:164-194): pts [n,s,3], viewdirs [n,3] (or NULL) -> raw [n,s,4]
 * with sigma forced to 0 where the embedder's keep_mask is false (:187-188). */
NRF_API size_t nrf_run_network_workspace_bytes(const nrf_renderer *r, int64_t n, int s);
NRF_API int nrf_run_network(const nrf_renderer *r, const float *d_pts, const float *d_viewdirs, int64_t n, int s, int precision,
                            float *d_raw, void *d_workspace, size_t workspace_bytes, void *stream);

/* RenderRays (NeRFRenderer.h:366-459) over one chunk of n packed rays [n, 8 | 11].
 * d_t: [n_samples] linspace(0,1,n_samples); d_u: [n_importance] linspace(0,1,n_importance). */
NRF_API size_t nrf_render_rays_workspace_bytes(const nrf_renderer *r, int64_t n, const nrf_render_params *p);
NRF_API int nrf_render_rays(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p,
                            const float *d_t, const float *d_u, const nrf_render_outputs *out,
                            void *d_workspace, size_t workspace_bytes, void *stream);

/* BatchifyRays (NeRFRenderer.h:465-525): the host loop over Chunk-sized slices of a packed ray batch, inside the library -- every output of `out`
 * is the whole batch's buffer ([n, ...]), slice i of the loop writes its rows in place (what the reference's torch::cat assembles afterwards), and
 * p->ray_base advances with the slice so the counter-based draws of the stochastic branches do not depend on Chunk.  Workspace: one chunk's per lane
 * (nrf_set_render_lanes).  A renderer is bound to ONE device and ONE caller at a time: its lane streams and fork / join events are created on first use, follow the
 * renderer to another device when a later call comes from there, and are shared by every call -- concurrent calls on one renderer from several host threads are not supported. */
NRF_API size_t nrf_batchify_rays_workspace_bytes(const nrf_renderer *r, int64_t n, int chunk, const nrf_render_params *p);
NRF_API int nrf_batchify_rays(const nrf_renderer *r, const float *d_rays, int ray_stride, int64_t n, int chunk, const nrf_render_params *p,
                              const float *d_t, const float *d_u, const nrf_render_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream);

/* NeRFRenderer::Render for a pose (NeRFRenderer.h:530-605), restricted to the row tile of `v`: nrf_view_rays + nrf_batchify_rays + the tile's
 * Near / Far in ONE call -- what a rank of the row-tile sharding issues per frame; ~25 asynchronous launches, no synchronisation, no allocation.
 * out: buffers for the TILE's rows*w rays.  d_rays_out (optional): receives the packed rays [rows*w, 8|11] (else they live in the workspace).
 * d_near_far (optional): see nrf_view_rays.  p->ray_base is taken as the index of the FRAME's first ray; the tile adds row0 * w itself.
 * ndc together with cone rays (p->has_cone): NDCRays' factor on cone_angle is the NDC direction's norm over itself (RayUtils.h:73-81), exactly 1: p->cone_angle is used as is. */
NRF_API size_t nrf_render_rows_workspace_bytes(const nrf_renderer *r, const nrf_view *v, const nrf_render_params *p);
NRF_API int nrf_render_rows(const nrf_renderer *r, const nrf_view *v, const nrf_render_params *p, const float *d_t, const float *d_u,
                            const nrf_render_outputs *out, float *d_rays_out, float *d_near_far, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training step: NeRFExecutor::Train, NeRFExecutor.h:862-995.
 *   render (nrf_render_rays) -> huber loss -> backward of the FINE pass only (z_samples are detached, NeRFRenderer.h:429):
 *   RawToOutputs -> sigma mask -> MLP -> hash grid -> Adam(lr, betas (0.9, 0.99), eps 1e-15) (:539).
 * All gradients fp32; accumulations use hardware fp32 atomics (order-free definition: compare with a tolerance).
 * ------------------------------------------------------------------------------------------- */
/* torch::nn::functional::huber_loss (delta 1, mean) and torch::mse_loss (:882-887).  d_loss_mse: device [2] = (huber, mse);
 * d_grad (optional): d huber / d pred, same shape as pred. */
NRF_API int nrf_huber_loss(const float *d_pred, const float *d_target, int64_t count, float *d_loss_mse, float *d_grad, void *stream);

/* Backward of RawToOutputs (NeRFRenderer.h:199-282) w.r.t. raw given d loss / d RGBMap [n,3]; TruncExp::backward clamps its
 * argument to [-100, 5] (CustomOps.cpp:11-15). */
NRF_API int nrf_raw2outputs_backward(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c, int white_bkgr,
                                     const float *d_g_rgb, float *d_g_raw, void *stream);
/* ... of a forward that ran with raw_noise_std > 0: d_noise [n,s] are the same normal draws (nrf_rng_fill(seed, NRF_RNG_NOISE_FINE, ...)). */
NRF_API int nrf_raw2outputs_backward_noise(const float *d_raw, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, int c,
                                           int white_bkgr, const float *d_noise, float noise_std, const float *d_g_rgb, float *d_g_raw, void *stream);
/* d/d sigma = 0 where keep is false: the backward of `outputs_flat[~keep_mask, -1] = 0` (NeRFRenderer.h:187-188). */
NRF_API int nrf_mask_sigma_grad(const uint8_t *d_keep, int64_t p, int c, float *d_g_raw, void *stream);

/* Backward of the MLP, fp32: NeRFSmallImpl::forward (NeRF.cpp:322-412) or -- round 5 -- the classic NeRFImpl::forward (NeRF.cpp:92-126: biases, the skip concat
 * cat[input_pts, h], the view-direction head or output_linear).  x [p, in_dims] as given to nrf_mlp_forward, g_out [p, output dims] (4 with view directions).
 * d_g_params (blob layout) is ACCUMULATED into; d_g_x (optional) receives d loss / d x[:, :input_ch] as [p, input_ch].  Pinned by LibTorch autograd through the compiled
 * NeRF.cpp (goldens train_hash, mlp_nerf_bwd*, train_classic). */
NRF_API size_t nrf_mlp_backward_workspace_bytes(const nrf_mlp *m, int64_t p);
NRF_API int nrf_mlp_backward(const nrf_mlp *m, const float *d_x, const float *d_g_out, int64_t p, float *d_g_params, float *d_g_x,
                             void *d_workspace, size_t workspace_bytes, void *stream);
/* The same gradients on the matrix cores (fp16 operands, fp32 accumulation, power-of-two loss scaling chosen on the device from
 * max|g_out|): one fused kernel per 2^22 points, forward + gradient chain + weight gradients.  Built for in 32 / views 16 / 64-wide /
 * 2-3 sigma + 3-4 colour layers; NRF_ERR_UNSUPPORTED otherwise.  Same arguments as nrf_mlp_backward.  The workspace is a fixed ~25 MB (one slot of operand
 * fragments per resident wave), whatever p. */
NRF_API size_t nrf_mlp_backward_f16_workspace_bytes(const nrf_mlp *m, int64_t p);
NRF_API int nrf_mlp_backward_f16(const nrf_mlp *m, const float *d_x, const float *d_g_out, int64_t p, float *d_g_params, float *d_g_x,
                                 void *d_workspace, size_t workspace_bytes, void *stream);
/* ... with the network input given the way the renderer's fast path has it instead of as [p, 48] fp32 rows: d_feats_lm = the level-major fp16 hash features of
 * nrf_hash_encode_lm_f16 ([16][p][2] halfs), d_dirs_f16 = [p / s][16] fp16 direction features, one row per RAY (point i belongs to ray i / s). */
NRF_API int nrf_mlp_backward_f16_lm(const nrf_mlp *m, const void *d_feats_lm, const void *d_dirs_f16, int s, const float *d_g_out, int64_t p, float *d_g_params,
                                    float *d_g_x, void *d_workspace, size_t workspace_bytes, void *stream);
/* Overflow report of the last nrf_mlp_backward_f16(_lm) that used `d_workspace` (the chain runs on fp16 operands behind a loss scale taken from max |g_out|):
 * flags_out[0] != 0: the incoming gradient held an inf / NaN; flags_out[1] != 0: an accumulated parameter gradient is not finite.  Host array of 2; synchronises
 * `stream`.  A caller skips (or rescales) the optimizer step when either is set. */
/* nrf_mlp_backward_f16_lm with the features read through a column map: point q reads column d_src[q] of the level-major table [16][pstride] (what
 * nrf_renderer_last_features hands over); nrf_mask_sigma_grad_src: nrf_mask_sigma_grad with the keep mask given by column the same way. */
NRF_API int nrf_mlp_backward_f16_lm_src(const nrf_mlp *m, const void *d_feats_lm, int64_t pstride, const int32_t *d_src, const void *d_dirs_f16, int s, const float *d_g_out, int64_t p,
                                        float *d_g_params, float *d_g_x, void *d_workspace, size_t workspace_bytes, void *stream);
NRF_API int nrf_mask_sigma_grad_src(const uint8_t *d_keep_cols, const int32_t *d_src, int64_t p, int c, float *d_g_raw, void *stream);
NRF_API int nrf_mlp_backward_f16_flags(const void *d_workspace, uint32_t *flags_out, void *stream);
/* The same two words without a host wait in the training step: _async copies them to h_flags2 (two uint32; pinned memory keeps the copy asynchronous) in `stream`'s order
 * -- read them after the stream, or an event recorded behind the call, has passed; _device returns where they live on the device, for nrf_adam_step_guarded: the
 * optimizer step is then skipped ON THE DEVICE when the chain overflowed, and the host learns of it one step later (nerfpp_amd/train.py). */
NRF_API int nrf_mlp_backward_f16_flags_async(const void *d_workspace, uint32_t *h_flags2, void *stream);
NRF_API const uint32_t *nrf_mlp_backward_f16_flags_device(const void *d_workspace);
/* Replace the parameter blob (same layout) and refresh the derived operands (transposed layers, matrix-core images).
 * NeRFSmall handles do it ON THE DEVICE in stream order (a gather + hi / lo split per image, no host round trip, nothing waits): work already issued on `stream`
 * reads the old images, work issued after reads the new ones; other streams of the caller's are the caller's to order.  The LeRF head of the built dimensions does it
 * on the device too when `params` is a device pointer (its Gram matrix W^T W in double with the host packer's summation order, the images by the host packer's own layout
 * function compiled for the device; checked byte for byte against the host packer when the handle is created) with ONE 4-byte read-back (the Gram matrix's
 * power-of-two scale is a launch argument).  The classic 8 x 256 network does it on the device as NeRFSmall does (its three images are gathers of [blob | merged
 * views layer]: the gather maps are decoded from the host packers run on probe blobs and verified against the host-packed images when the handle is created; the merged
 * layer is re-derived by a device kernel with the host's double sums).  Anything else (and NRF_MLP_HOST_REPACK=1) copies the blob to the host, repacks there and
 * synchronises `stream`. */
NRF_API int nrf_mlp_set_params(nrf_mlp *m, const float *params, int params_on_device, void *stream);
/* Number of derived images nrf_mlp_set_params refreshes on the device for this handle (0: the host repack). */
NRF_API int nrf_mlp_device_repack_images(const nrf_mlp *m);
/* NRF_PREC_F16_SPLIT, NeRFSmall (bias-free: NeRF.cpp:322-412): range safety.  ReLU is positively homogeneous, so the split-precision operand image holds 2^e_l W_l per
 * layer and the kernels take the product of the scales out of (rgb, sigma) again -- exact in fp32 -- while every fp16 (hi, lo) pair the matrix cores read (weights, and the
 * activations between layers) has two NORMAL halves whatever the magnitude of the checkpoint's weights: a model trained with xavier gain 0.1 (|W| ~ 0.01), a sigma head
 * scaled by 100, weights x 2^-8 or x 2^6 all render as accurately as weights near 1.  The exponents are chosen on the device from the blob itself at create time and at
 * every nrf_mlp_set_params (per-layer RMS row norms -> an RMS model of the activations, target RMS 8; mlp.hip, k_small_scales), in stream order.
 * nrf_mlp_set_input_rms_hint: the expected RMS of the position features (default 0.25; a renderer built on a hash grid sets it from the table).
 * nrf_mlp_get_split_scales: the exponents in force as powers of two -- group_scales_out[12] (sigma-net layers, colour layer 0's direction columns / geo columns, colour
 * layers 1..), kernel_scales_out[8] ([0] 2^-S_sigma, [1] 2^-S_colour, [2] 2^S_hidden); synchronises `stream`.  All 1 for other families and with NRF_SPLIT_UNSCALED=1. */
NRF_API int nrf_mlp_set_input_rms_hint(nrf_mlp *m, float rms, void *stream);
/* on = 0: the split images from the blob as it is (all exponents 0) -- the representation of rounds 4-5, kept as an A/B switch and for tests of the non-finite word */
NRF_API int nrf_mlp_set_split_scaling(nrf_mlp *m, int on, void *stream);
NRF_API int nrf_mlp_get_split_scales(const nrf_mlp *m, float *group_scales_out, float *kernel_scales_out, void *stream);

/* Backward of the hash grid w.r.t. its table: d_g_emb [p, L*F] -> d_g_table, fp32 in the table's own layout, ACCUMULATED into.
 * NRF_HASH_NGP: nn::Embedding's index_add of the trilinear weights (NeRF.cpp:279-298).  NRF_HASH_CU: CuHashEmbedderBackwardKernel
 * (CuHashEmbedder.cu:105-216) -- contributions rounded to fp16 after the x128 gradient scaling exactly as there, but accumulated
 * in fp32 instead of with fp16 atomics. */
NRF_API int nrf_hash_backward(const nrf_hash *h, const float *d_x, int64_t p, const float *d_g_emb, float *d_g_table, void *stream);

/* The same gradient for points that are the samples of rays, pts [n, s, 3] ray-major (what a training step has): consecutive samples
 * of a ray that share a voxel are summed in registers before the atomic add -- the count of scattered float atomics is the cost. */
NRF_API int nrf_hash_backward_rays(const nrf_hash *h, const float *d_pts, int64_t n, int s, const float *d_g_emb, float *d_g_table, void *stream);
/* ... with both features of an entry (n_features == 2) in ONE 64-bit integer atomic: two 32-bit fixed-point fields, scaled by a power of
 * two that a device-side pass derives from a rigorous bound on any entry's total (sum over the points of max_f |g|, per level), so the fields
 * cannot overflow; decoded and ACCUMULATED into d_g_table (fp32) at the end.  Half the atomics of nrf_hash_backward_rays -- the L2 atomic
 * rate, not bytes, bounds this pass.  Resolution: (bound / 2^30) per addend, i.e. ~(active entries of a level) * 2^-30 relative to a typical
 * entry.  d_workspace: nrf_hash_backward_packed_workspace_bytes(h), 256-byte aligned; no host synchronisation. */
NRF_API size_t nrf_hash_backward_packed_workspace_bytes(const nrf_hash *h);
NRF_API int nrf_hash_backward_rays_packed(const nrf_hash *h, const float *d_pts, int64_t n, int s, const float *d_g_emb, float *d_g_table,
                                          void *d_workspace, size_t workspace_bytes, void *stream);
/* The same gradient with the contributions MERGED before they reach memory: 8-byte records {word in bin, two 25-bit fixed-point fields} are binned by ranges of 2^14 table words
 * (count pass, scan, emit pass), each bin is summed in one workgroup's LDS and added to d_g_table without atomics.  Same groups, same scale, integer sums:
 * the result equals nrf_hash_backward_rays_packed bit for bit.  log2_hashmap_size <= 19.  Workspace: nrf_hash_backward_binned_workspace_bytes(h, s)
 * (~0.27 GB at s = 192: 2^18 points x 8 corners x levels x 8 bytes), 256-byte aligned. */
NRF_API size_t nrf_hash_backward_binned_workspace_bytes(const nrf_hash *h, int s);
/* ... for batches of at most n rays: the record buffer (8 B x 8 corners x levels per sample) is sized for min(n, one 2^18-point pass) instead of the whole pass */
NRF_API size_t nrf_hash_backward_binned_workspace_bytes_for(const nrf_hash *h, int64_t n, int s);
NRF_API int nrf_hash_backward_rays_binned(const nrf_hash *h, const float *d_pts, int64_t n, int s, const float *d_g_emb, float *d_g_table,
                                          void *d_workspace, size_t workspace_bytes, void *stream);

/* TotalVariationLoss of the LibTorch HashEmbedder (NeRF.h:255-300, NeRFExecutor.h:896-913; NRF_HASH_NGP grids): the cube of
 * (cube_size + 1)^3 lattice vertices at min_vertex (host [3]; the reference draws it with torch::randint) of `level`.
 * d_loss (device, 1 float) += weight * loss;  d_g_table (optional, table layout) += weight * d loss / d table. */
NRF_API int nrf_hash_tv_loss(const nrf_hash *h, const float *d_table, int level, const int *min_vertex, int cube_size, float weight, float *d_loss,
                             float *d_g_table, void *stream);

/* torch::optim::Adam::step without weight decay / amsgrad; t = 1, 2, ... */
NRF_API int nrf_adam_step(float *d_p, const float *d_g, float *d_m, float *d_v, int64_t n, float lr, float beta1, float beta2, float eps, int t,
                          void *stream);
/* nrf_adam_step that updates NOTHING (parameters and moments alike) when any of the n_flags (<= 16) words at d_flags is non-zero at the time the kernel runs */
NRF_API int nrf_adam_step_guarded(float *d_p, const float *d_g, float *d_m, float *d_v, int64_t n, float lr, float beta1, float beta2, float eps, int t,
                                  const uint32_t *d_flags, int n_flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Image-space tail of NeRFExecutor::RenderPath (NeRFExecutor.h:690, :698-700; TorchTensorToCVMat, NeRFRenderer.h:58-68)
 * ------------------------------------------------------------------------------------------- */
/* depth' = (depth - near) / (far - near) with the frame's scalar Near / Far (nrf_near_far_range).  In place allowed. */
NRF_API int nrf_normalize_depth(const float *d_depth, int64_t n, float near_, float far_, float *d_out, void *stream);
/* u8 = (uint8)clamp(x * 255, 0, 255): what cv::imwrite receives for RGB / disparity / normalised depth. */
NRF_API int nrf_to_u8(const float *d_x, int64_t n, uint8_t *d_out, void *stream);

/* The render-factor step of NeRFExecutor::RenderView (NeRFExecutor.h:618-627), host only: with render_factor != 0 the frame is rendered
 * downsampled, h1 = (int)(h / render_factor), w1 = (int)(w / render_factor) (int / float -> float -> int, as the reference's
 * `h = h / rparams.RenderFactor` does with its float field) and fx, fy, cx, cy of K (host [9], row-major) are divided by it in fp32;
 * render_factor == 0 copies.  NeRFExecutor::RenderPath (:657-662) scales h, w and a local focal but hands Render() the ORIGINAL k --
 * callers that mirror RenderPath pass k unchanged and only use h1 / w1 (K1 may be NULL). */
NRF_API int nrf_render_view_dims(int h, int w, const float *K, float render_factor, int *h1, int *w1, float *K1);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: whole-image render batches partitioned by ray into contiguous row tiles, one process per GPU, the model replicated
 * read-only; per-tile pixels return to every rank with ONE RCCL collective per step over xGMI (north_star).  The reference
 * is single-GPU (no counterpart).  RCCL is resolved with dlopen at first use -- the copy already mapped into the process (LibTorch's)
 * if there is one -- so this library has no link-time RCCL dependency; without RCCL the nrf_comm_* calls return NRF_ERR_UNSUPPORTED.
 * NRF_RCCL_LIBRARY=<path> in the environment names the copy to use instead (a host that maps several; the tests' threads-as-ranks stand-in).
 * ------------------------------------------------------------------------------------------- */
/* Rank `rank` of `world` renders image rows [row0, row0 + rows): contiguous, the first h % world ranks one row taller.  Host only. */
NRF_API int nrf_tile_partition(int h, int world, int rank, int *row0, int *rows);

#define NRF_COMM_ID_BYTES 128
typedef struct nrf_comm nrf_comm;
/* Rendezvous as in NCCL: rank 0 calls nrf_comm_unique_id (id_out: host [NRF_COMM_ID_BYTES]) and hands the bytes to every rank by any
 * out-of-band means (a file, a socket, torch.distributed, MPI); every rank then calls nrf_comm_create on ITS device
 * (hipSetDevice first; blocks until all `world` ranks arrive).  nrf_comm_wrap adopts a live ncclComm_t the host already owns
 * (it is not destroyed by nrf_comm_destroy). */
NRF_API int nrf_comm_unique_id(void *id_out);
NRF_API int nrf_comm_create(const void *id, int world, int rank, nrf_comm **out);
/* ... with a bounded rendezvous: the (ordinary, blocking) ncclCommInitRank runs on a helper thread bound to the caller's device and is waited for; after timeout_s
 * seconds without all `world` ranks NRF_ERR_HIP is returned (a peer that never started, or an id of another launch) and the caller should exit.  timeout_s <= 0 waits for
 * ever on the calling thread.  nrf_comm_create itself uses NRF_COMM_TIMEOUT_S from the environment (default 300; only a well-formed number is taken, 0 = wait for ever).
 * A non-blocking communicator handed over with nrf_comm_wrap is supported too: nrf_allgather_tiles waits (bounded) until its group has been enqueued before it returns. */
NRF_API int nrf_comm_create_timeout(const void *id, int world, int rank, double timeout_s, nrf_comm **out);
NRF_API int nrf_comm_wrap(void *nccl_comm, nrf_comm **out);
NRF_API void nrf_comm_destroy(nrf_comm *c);
NRF_API int nrf_comm_world(const nrf_comm *c);
NRF_API int nrf_comm_rank(const nrf_comm *c);
/* d_tiles: this rank's [frames, rows_rank, w, c] fp32 tiles of `frames` images (rows_rank from nrf_tile_partition; may be NULL on a rank that owns no rows, h < world);
 * d_frames: [frames, h, w, c] on every rank.  One fused launch on `stream` (ncclAllGather per frame when h % world == 0, grouped
 * ncclBroadcast per tile otherwise); asynchronous like every other call.  d_tiles and d_frames must not overlap. */
NRF_API int nrf_allgather_tiles(const nrf_comm *c, const float *d_tiles, int frames, int h, int w, int c_channels, float *d_frames, void *stream);
/* The data-parallel TRAINING step's exchange (no reference counterpart: the reference trains on one device, NeRFExecutor.h:862-995): the n_grads fp32
 * gradient buffers (hash table, network blob; counts[i] elements each) become their MEAN over the ranks, in place -- ncclAllReduce(sum) in slices of bucket_bytes
 * (0: 32 MiB) as one group launch, then a multiply by 1 / world, all on `stream`.  overflow >= 0 first makes the fp16 backward's per-rank overflow report collective
 * (all-reduce max of one word + one host read-back): *skip_out = 1 on every rank iff any rank passed a non-zero overflow, and no gradient is exchanged then (a peer's inf
 * is never summed in; every replica skips the same optimizer step).  overflow < 0: no agreement, nothing synchronises (skip_out may be NULL).  World of one: identity. */
NRF_API int nrf_allreduce_grads(const nrf_comm *c, float *const *d_grads, const int64_t *counts, int n_grads, int64_t bucket_bytes, int overflow, int *skip_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * LeRF render pass as library calls           LeRFRenderer (LeRFRenderer.h:56-132, LeRFRenderer.cpp:85-330)
 * ------------------------------------------------------------------------------------------- */
typedef struct nrf_lerf_renderer_desc {
    const nrf_hash *lang_embed;   /* LangEmbedFn: CuHashEmbedder L16 F8 (main.cpp:203-213) */
    const nrf_mlp *lerf;          /* Lerf: nrf_mlp_lerf_create; arithmetic of the fused passes: nrf_lerf_set_precision */
} nrf_lerf_renderer_desc;

typedef struct nrf_lerf_outputs {     /* LeRFRendererOutputs (LeRFRenderer.h:9-18); NULL = not wanted */
    float *d_embedding;        /* RenderedLangEmbedding [n, E] */
    float *d_disp;             /* DispMapLE [n] */
    float *d_acc;              /* AccMapLE [n] */
    float *d_depth;            /* DepthMapLE [n] */
    float *d_weights;          /* WeightsLE [n, n_samples + n_importance] */
    float *d_relevancy;        /* Relevancy [n, 2] (needs nrf_lerf_set_prompts) */
    /* intermediates for parity tests (optional) */
    float *d_z_coarse;         /* [n, n_samples] */
    float *d_weights_coarse;   /* [n, n_samples] */
    float *d_z_fine;           /* [n, n_samples + n_importance] */
} nrf_lerf_outputs;

typedef struct nrf_lerf_renderer nrf_lerf_renderer;
NRF_API int nrf_lerf_renderer_create(const nrf_lerf_renderer_desc *desc, nrf_lerf_renderer **out);
NRF_API void nrf_lerf_renderer_destroy(nrf_lerf_renderer *r);
/* The LeRF pass and nrf_render_params.overflow_policy: the compositing kernel of the fine pass looks at every sigma_le and the final normalise at every rendered embedding;
 * the pass has no fp32 single-call twin to render a flagged chunk again with, so AUTO / RERENDER / ERROR all end a call with one read-back and answer a non-finite value with
 * NRF_ERR_NONFINITE; DEFERRED reports at the next call (or here); IGNORE does not look.  flagged_calls: total since the renderer was created. */
NRF_API int nrf_lerf_renderer_nonfinite(const nrf_lerf_renderer *r, int64_t *flagged_calls);
/* Lanes of this renderer's Chunk loop, 1-4.  Default ONE: the LeRF kernels gain nothing from sharing the CUs (measured: 1 lane 140-143 ms per 800x800 frame, 2 lanes 146-147). */
NRF_API int nrf_lerf_renderer_set_lanes(nrf_lerf_renderer *r, int lanes);
/* LeRFRenderer::SetLeRFPrompts (LeRFRenderer.h:86): [n_pos, E] / [n_neg, E] fp32 phrase embeddings (host or device), copied; 0 / 0 clears them.  Synchronises `stream`. */
NRF_API int nrf_lerf_set_prompts(nrf_lerf_renderer *r, const float *positives, int n_pos, const float *negatives, int n_neg, int on_device, void *stream);

/* LeRFRenderer::RenderRays (LeRFRenderer.cpp:85-187) over one chunk of packed rays [n, 8 | 11], hierarchical (n_importance > 0), deterministic (ThinRay, Perturb = 0:
 * NRF_ERR_UNSUPPORTED otherwise), on the fused matrix-core path: every sample point is hash-encoded once and its density net evaluated once, raw_le [n, S, 769] is never
 * formed.  Of `p`: n_samples, n_importance (multiples of 32 in sum), lindisp, sum_vec, coarse_mode (NRF_COARSE_AUTO: sigma_le of the coarse pass in exact fp32 when the head
 * runs in NRF_PREC_F16_SPLIT, so that the fine sample set is the fp32 stage path's bit for bit; NRF_COARSE_FULL: the timed arithmetic).  d_t / d_u as nrf_render_rays. */
NRF_API size_t nrf_lerf_render_rays_workspace_bytes(const nrf_lerf_renderer *r, int64_t n, const nrf_render_params *p);
NRF_API int nrf_lerf_render_rays(const nrf_lerf_renderer *r, const float *d_rays, int ray_stride, int64_t n, const nrf_render_params *p, const float *d_t, const float *d_u,
                                 const nrf_lerf_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream);
/* LeRFRenderer::BatchifyRays (LeRFRenderer.cpp:189-263): the Chunk loop inside the library, slices written in place, on the lanes of nrf_set_render_lanes /
 * nrf_lerf_renderer_set_lanes (one device and one caller at a time per renderer, as nrf_batchify_rays). */
NRF_API size_t nrf_lerf_batchify_rays_workspace_bytes(const nrf_lerf_renderer *r, int64_t n, int chunk, const nrf_render_params *p);
NRF_API int nrf_lerf_batchify_rays(const nrf_lerf_renderer *r, const float *d_rays, int ray_stride, int64_t n, int chunk, const nrf_render_params *p, const float *d_t,
                                   const float *d_u, const nrf_lerf_outputs *out, void *d_workspace, size_t workspace_bytes, void *stream);
/* LeRFRenderer::Render for a pose (LeRFRenderer.cpp:265-330), restricted to the row tile of `v`: rays + the Chunk loop + (with prompts) the relevancy in ONE call,
 * no synchronisation, no torch ops; out: buffers for the tile's rows*w rays; d_rays_out / d_near_far as nrf_render_rows. */
NRF_API size_t nrf_lerf_render_rows_workspace_bytes(const nrf_lerf_renderer *r, const nrf_view *v, const nrf_render_params *p);
NRF_API int nrf_lerf_render_rows(const nrf_lerf_renderer *r, const nrf_view *v, const nrf_render_params *p, const float *d_t, const float *d_u, const nrf_lerf_outputs *out,
                                 float *d_rays_out, float *d_near_far, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * N1, LeRF branch of the optimisation step (NeRFExecutor.h:955-982): lang_loss and its backward into LeRFImpl and the language grid
 * ------------------------------------------------------------------------------------------- */
/* lang_loss = huber_loss(pred, target, reduction none, delta).sum(-1).nanmean() (NeRFExecutor.h:970-974; delta 1.25 there): d_loss [1]; d_grad [n, e] = d loss / d pred
 * (NULL: not wanted).  A row holding a NaN leaves the mean (count = the other rows); its gradient row is 0 except NaN at the NaN elements -- what LibTorch's backward
 * leaves there (golden train_lerf_nan). */
NRF_API int nrf_huber_rows_nanmean(const float *d_pred, const float *d_target, int64_t n, int e, float delta, float *d_loss, float *d_grad, void *stream);
/* Backward of the FINE pass of LeRFRenderer::RenderRays (LeRFRenderer.cpp:141-172; the coarse pass carries none: z_samples are detached, :150) downstream of the language
 * grid: d_emb [n*s, in] = lang_embed_fn->forward(pts) -> LeRFImpl::forward (LeRF.cpp:86-108) -> sigma_le[~keep] = 0 (LeRFRenderer.cpp:37-38) -> RawToLEOutputs' weights
 * (:38-66; d_noise [n, s] / noise_std: the RawNoiseStd draws, NULL / 0 = none) -> RenderCLIPEmbedding (LeRFRenderer.h:45-54), given d_g_rendered [n, E] = d loss /
 * d RenderedLangEmbedding.  fp32; the forward is recomputed here (every layer input kept, chunks of whole rays).  d_g_params: the head's blob layout, ACCUMULATED into;
 * d_g_emb [n*s, in] written (NULL: not wanted); d_rendered [n, E] / d_weights [n, s]: the recomputed forward (NULL: not wanted).  d_dirs: rays_d, row stride d_stride. */
NRF_API size_t nrf_lerf_head_backward_workspace_bytes(const nrf_mlp *lerf, int64_t n, int s);
NRF_API int nrf_lerf_head_backward(const nrf_mlp *lerf, const float *d_emb, const uint8_t *d_keep, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s,
                                   const float *d_noise, float noise_std, const float *d_g_rendered, float *d_g_params, float *d_g_emb, float *d_rendered, float *d_weights,
                                   void *d_workspace, size_t workspace_bytes, void *stream);
/* ... with the language grid in front -- the backward of one LeRFRenderer::Render call on a ray batch in ONE library call: d_pts [n, s, 3] the fine pass's sample points
 * (o + d z, after TangentScatter / preconditioning where those are on) -> nrf_hash_encode -> the head's backward -> nrf_hash_backward_rays (CuHashEmbedderBackwardKernel's
 * gradient, CuHashEmbedder.cu:105-216).  d_g_lerf_params (head blob) and d_g_table (the grid's fp32 table layout) are ACCUMULATED into. */
NRF_API size_t nrf_lerf_backward_points_workspace_bytes(const nrf_lerf_renderer *r, int64_t n, int s);
NRF_API int nrf_lerf_backward_points(const nrf_lerf_renderer *r, const float *d_pts, const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, const float *d_noise,
                                     float noise_std, const float *d_g_rendered, float *d_g_lerf_params, float *d_g_table, void *d_workspace, size_t workspace_bytes, void *stream);
/* Where the most recent render call on this LeRF renderer left the language features of its fine depths (as nrf_renderer_last_features; a call that rendered exactly ONE
 * chunk): the level-major fp16 table [16][cols][8] (coarse columns first, the new samples' behind them), the keep mask by column, the merge map [n, sf].  They live in the
 * caller's workspace: valid until the next render call on this renderer or any other use of that workspace.  NRF_ERR_UNSUPPORTED when the last call left none (serial is set
 * either way).  nrf_lerf_backward_points_src: nrf_lerf_backward_points whose head backward reads those rows through the map instead of encoding d_pts again (the points must be
 * the ones that render encoded: thin rays, no preconditioning); d_pts still feeds the grid's gradient scatter. */
NRF_API int nrf_lerf_renderer_last_features(const nrf_lerf_renderer *r, const void **d_feats_lm, int64_t *cols, const uint8_t **d_keep_cols, const int32_t **d_src, int64_t *n, int *sf,
                                            uint64_t *serial /* optional */);
NRF_API int nrf_lerf_backward_points_src(const nrf_lerf_renderer *r, const void *d_feats_lm, int64_t cols, const uint8_t *d_keep_cols, const int32_t *d_src, const float *d_pts,
                                         const float *d_z, const float *d_dirs, int d_stride, int64_t n, int s, const float *d_noise, float noise_std, const float *d_g_rendered,
                                         float *d_g_lerf_params, float *d_g_table, void *d_workspace, size_t workspace_bytes, void *stream);

/* The library's short-lived device buffers (a layer product's split operand image, slice sums, reduction cells) come from blocks it keeps per (device, stream) and reuses
 * from call to call -- not from hipMallocAsync, whose pool proved unsafe inside a LibTorch host (scratch.hip).  nrf_scratch_trim gives the idle blocks back to the driver
 * (after a hipDeviceSynchronize of their devices) and returns the bytes freed; optional. */
NRF_API size_t nrf_scratch_trim(void);

/* ---------------------------------------------------------------------------------------------
 * Instrumentation (bench / tests)
 * ------------------------------------------------------------------------------------------- */
/* When enabled, nrf_render_rays brackets its dominant kernels with HIP events on the caller's stream;
 * nrf_profile_read synchronises them and returns accumulated milliseconds and launch counts. */
/* NRF_PROF_MLP_COLOUR: the NeRFSmall kernel's colour-net-only launch of a hierarchical render's fine pass (its S coarse depths; see nrf_render_params.coarse_mode) --
 * a different amount of work per point than NRF_PROF_MLP's whole-network launches, so it has its own slot. */
enum { NRF_PROF_HASH = 0, NRF_PROF_MLP = 1, NRF_PROF_COMPOSITE = 2, NRF_PROF_SAMPLE = 3, NRF_PROF_OTHER = 4, NRF_PROF_SIGMA = 5, NRF_PROF_MLP_COLOUR = 6, NRF_PROF_COUNT = 7 };
/* The Chunk loop of nrf_batchify_rays / nrf_render_rows runs consecutive chunks on `lanes` internal streams (default 2, at most 4; forked from and joined to the
 * caller's stream) so that one chunk's gather-bound kernels overlap another's matrix-bound ones; results do not depend on it.  lanes = 1 restores the single-stream
 * loop (also: NRF_RENDER_LANES=1..4).  Process-wide setting, read at every call; the workspace query and the call must see the same value. */
NRF_API int nrf_set_render_lanes(int lanes);
NRF_API int nrf_get_render_lanes(void);
/* ... per renderer: 1-4 lanes for this renderer's calls whatever the process-wide setting; 0 returns it to that setting (two renderers of one process need not share it). */
NRF_API int nrf_renderer_set_lanes(nrf_renderer *r, int lanes);
/* 1 when the fp32 layer products of the training paths (classic and LeRF backward, NeRFSmall's fp32 backward) run as rocBLAS GEMMs on the fp32 matrix cores (the library is
 * looked up with dlopen at first use, preferring the copy the process already maps), 0 when they run the hand-written FMA kernels (rocBLAS absent, or NRF_FP32_GEMM=0). */
NRF_API int nrf_fp32_gemm_available(void);
/* Arithmetic of the forward / back-propagation products of the classic-NeRF and LeRF training steps (gemm_bf16x3.hip; NRF_TRAIN_GEMM=auto | f32 | bf16x3 | f16x3 in the
 * environment selects the process-wide default):
 *  -1  (default, "auto") by network family: 2 for the classic NeRF and the LeRF head, 0 for NeRFSmall's fp32 backward (the hash path's parity chain; its fast chain is
 *      the fused fp16 backward, nrf_mlp_backward_f16).
 *   0  fp32 products (rocBLAS sgemm where present, else the FMA kernels).
 *   1  bf16x3: every operand as hi + lo bf16 (fp32's exponent range, 16 significant bits), three matrix-core products per fp32 accumulator, bias / ReLU / ReLU mask
 *      fused in the epilogue.  One product within 6e-6 of its largest entry; a whole chain's weight gradients within ~1e-3 of the fp32 chain's.
 *   2  f16x3: hi + lo fp16 (22 significant bits) of power-of-two scaled operands -- each row of A by its own largest entry, B by its largest entry, undone exactly in
 *      the epilogue; one product is closer to the float64 product than sgemm's (5e-7 against 8e-7 of the largest entry at K = 256), a classic step's weight gradients
 *      end 6e-5 (norm-wise) from the rocBLAS chain's -- rocBLAS and the FMA kernels are 2e-5 apart (ReLUs decided the other way by last-bit differences). */
NRF_API int nrf_get_train_gemm(void);
NRF_API int nrf_set_train_gemm(int mode);
/* The products themselves: C [m x n] (ldc) = A [m x k] (lda) . B [n x k]^T (ldb) (+ bias [n]) (ReLU), fp32 row-major in and out */
NRF_API int nrf_gemm_nt_bf16x3(const float *d_a, int lda, int64_t m, int k, const float *d_b, int ldb, int n, float *d_c, int ldc, const float *d_bias, int relu, void *stream);
/* dW [out x in] (ld in; columns col0 .. col0 + n) += G [p x out]^T (ldg) . X [p x n] (ldx), fp32 row-major: the weight-gradient product of a layer over p points
 * (bf16x3 arithmetic -- a per-point scale cannot be undone in a sum over points; deterministic: slices of the points summed in a fixed order) */
NRF_API int nrf_gemm_tn_bf16x3(const float *d_g, int ldg, int out, const float *d_x, int ldx, int n, int64_t p, float *d_dw, int in, int col0, void *stream);
/* One layer's weight gradient as the split-precision training modes compute it, with its bias gradient: dW += G^T X as above (out >= 32) or, for a head of fewer rows,
 * by fp32 FMAs four rows per pass over X; d_db (optional) [out] += the column sums of G, out of the same pass over G where out >= 32 */
NRF_API int nrf_layer_grad_split(const float *d_g, int ldg, int out, const float *d_x, int ldx, int n, int64_t p, float *d_dw, int in, int col0, float *d_db, void *stream);
NRF_API int nrf_gemm_nt_f16x3(const float *d_a, int lda, int64_t m, int k, const float *d_b, int ldb, int n, float *d_c, int ldc, const float *d_bias, int relu, void *stream);
NRF_API int nrf_profile_enable(int on);
/* 1 while the event bracketing is on.  A throughput measurement must run with it off: an event pair around every kernel of every lane costs host time per launch and
 * separates the kernels on the device (bench.py asserts 0 before its timed region; the per-kernel times come from a separate pass).  Events are pooled: none is created on
 * a launch path after the first profiled frame. */
NRF_API int nrf_profile_is_enabled(void);
NRF_API int nrf_profile_read(double *ms /*[NRF_PROF_COUNT]*/, int64_t *launches /*[NRF_PROF_COUNT]*/, int reset);

#ifdef __cplusplus
}
#endif
#endif /* NERFPP_HIP_H */
