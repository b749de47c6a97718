"""ctypes loader for libnerfpp_hip.so (the C ABI declared in include/nerfpp_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C nerfpp_amd/csrc`.  There is no CPU
fallback: if the shared object is missing, importing the compute API raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRF_LIB_PATH") or os.path.join(_HERE, "lib", "libnerfpp_hip.so")   # NRF_LIB_PATH: tuning builds only

NRF_OK = 0
NRF_HASH_NGP, NRF_HASH_CU = 0, 1
NRF_SH_LIBTORCH, NRF_SH_CUDA = 0, 1
NRF_PREC_F32, NRF_PREC_F16_MFMA, NRF_PREC_F16_SPLIT = 0, 1, 2
NRF_DIRS_NONE, NRF_DIRS_PE, NRF_DIRS_SH_LIBTORCH, NRF_DIRS_SH_CUDA = 0, 1, 2, 3
(NRF_RNG_T_RAND, NRF_RNG_R_COARSE, NRF_RNG_THETA_COARSE, NRF_RNG_NOISE_COARSE, NRF_RNG_U_PDF, NRF_RNG_PRECOND, NRF_RNG_R_FINE, NRF_RNG_THETA_FINE,
 NRF_RNG_NOISE_FINE) = range(1, 10)       # include/nrf_rng.h
NRF_PROF_NAMES = ("hash", "mlp", "composite", "sample", "other", "sigma", "mlp_colour")
NRF_COARSE_AUTO, NRF_COARSE_FULL, NRF_COARSE_SIGMA_F32 = 0, 1, 2
NRF_OVERFLOW_AUTO, NRF_OVERFLOW_RERENDER, NRF_OVERFLOW_ERROR, NRF_OVERFLOW_DEFERRED, NRF_OVERFLOW_IGNORE = 0, 1, 2, 3, 4
NRF_ERR_NONFINITE = 5


class HashDesc(C.Structure):
    _fields_ = [("mode", C.c_int), ("n_levels", C.c_int), ("n_features", C.c_int), ("log2_hashmap_size", C.c_int),
                ("base_resolution", C.c_int), ("finest_resolution", C.c_int), ("bbox", C.c_float * 6)]


class MlpSmallDesc(C.Structure):
    _fields_ = [("input_ch", C.c_int), ("input_ch_views", C.c_int), ("num_layers", C.c_int), ("hidden_dim", C.c_int),
                ("geo_feat_dim", C.c_int), ("num_layers_color", C.c_int), ("hidden_dim_color", C.c_int),
                ("use_pred_normal", C.c_int), ("num_layers_normals", C.c_int), ("hidden_dim_normals", C.c_int)]


class MlpNerfDesc(C.Structure):
    _fields_ = [("depth", C.c_int), ("width", C.c_int), ("input_ch", C.c_int), ("input_ch_views", C.c_int),
                ("output_ch", C.c_int), ("skip", C.c_int), ("use_viewdirs", C.c_int)]


class RendererDesc(C.Structure):
    _fields_ = [("hash", C.c_void_p), ("pe_freqs", C.c_int), ("dirs_encoder", C.c_int), ("dirs_param", C.c_int), ("mlp", C.c_void_p)]


class RenderParams(C.Structure):
    _fields_ = [("n_samples", C.c_int), ("n_importance", C.c_int), ("lindisp", C.c_int), ("white_bkgr", C.c_int),
                ("precision", C.c_int), ("sum_vec", C.c_int),
                ("perturb", C.c_float), ("has_cone", C.c_int), ("cone_angle", C.c_float), ("raw_noise_std", C.c_float), ("precond_alpha", C.c_float),
                ("has_bbox", C.c_int), ("bbox", C.c_float * 6), ("seed", C.c_uint64), ("ray_base", C.c_int64), ("coarse_mode", C.c_int), ("overflow_policy", C.c_int)]


class RenderOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("d_rgb", "d_disp", "d_acc", "d_depth", "d_weights", "d_raw",
                                          "d_z_coarse", "d_raw_coarse", "d_weights_coarse", "d_z_fine")]


class LerfRendererDesc(C.Structure):
    _fields_ = [("lang_embed", C.c_void_p), ("lerf", C.c_void_p)]


class LerfOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("d_embedding", "d_disp", "d_acc", "d_depth", "d_weights", "d_relevancy", "d_z_coarse", "d_weights_coarse", "d_z_fine")]


class View(C.Structure):          # nrf_view
    _fields_ = [("h", C.c_int), ("w", C.c_int), ("K", C.c_float * 9), ("c2w", C.c_float * 12), ("has_staticcam", C.c_int), ("c2w_staticcam", C.c_float * 12),
                ("row0", C.c_int), ("rows", C.c_int), ("use_viewdirs", C.c_int), ("ndc", C.c_int), ("chunk", C.c_int), ("bbox", C.c_float * 6)]


# every symbol include/nerfpp_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "nrf_version", "nrf_last_error", "nrf_status_string",
    "nrf_get_rays", "nrf_ndc_rays", "nrf_aabb", "nrf_pack_rays", "nrf_pack_rays_viewsrc", "nrf_view_rays", "nrf_near_far_range", "nrf_near_far_range_device", "nrf_linspace", "nrf_z_vals", "nrf_points",
    "nrf_precrop_bounds", "nrf_rand_pixels", "nrf_ray_batch", "nrf_gather_pixels",
    "nrf_pe_encode", "nrf_sh_encode",
    "nrf_hash_create", "nrf_hash_destroy", "nrf_hash_output_dims", "nrf_hash_table_elems", "nrf_hash_set_table", "nrf_hash_set_primes", "nrf_hash_set_dense_budget", "nrf_hash_get_dense_budget", "nrf_hash_get_level_scales", "nrf_hash_set_level_scales",
    "nrf_hash_encode",
    "nrf_mlp_small_param_count", "nrf_mlp_nerf_param_count", "nrf_mlp_small_create", "nrf_mlp_nerf_create", "nrf_mlp_destroy",
    "nrf_mlp_output_dims", "nrf_mlp_forward",
    "nrf_mlp_lerf_param_count", "nrf_mlp_lerf_create",
    "nrf_lerf_mfma_available", "nrf_lerf_set_precision", "nrf_lerf_sigma", "nrf_lerf_render_embedding",
    "nrf_raw2outputs", "nrf_raw2weights", "nrf_raw2weights_gather", "nrf_render_clip_embedding", "nrf_sample_pdf", "nrf_fine_depths", "nrf_fine_depths_merge",
    "nrf_rng_fill", "nrf_jitter_z", "nrf_tangent_scatter", "nrf_precondition", "nrf_raw2outputs_noise", "nrf_sample_pdf_rand", "nrf_fine_depths_rand",
    "nrf_renderer_create", "nrf_renderer_destroy", "nrf_run_network_workspace_bytes", "nrf_run_network",
    "nrf_render_rays_workspace_bytes", "nrf_render_rays", "nrf_batchify_rays_workspace_bytes", "nrf_batchify_rays", "nrf_render_rows_workspace_bytes", "nrf_render_rows",
    "nrf_normalize_depth", "nrf_to_u8",
    "nrf_huber_loss", "nrf_raw2outputs_backward", "nrf_raw2outputs_backward_noise", "nrf_mask_sigma_grad", "nrf_mlp_backward_workspace_bytes", "nrf_mlp_backward", "nrf_mlp_backward_f16_workspace_bytes", "nrf_mlp_backward_f16", "nrf_mlp_backward_f16_lm", "nrf_mlp_backward_f16_flags", "nrf_hash_encode_lm_f16", "nrf_hash_encode_lm_f16_strided", "nrf_lerf_sigma_lm", "nrf_lerf_sigma_lm_strided", "nrf_lerf_render_embedding_lm", "nrf_lerf_render_embedding_lm_gather", "nrf_lerf_geo_bytes", "nrf_lerf_sigma_geo_lm_strided", "nrf_lerf_sigma_exact_available", "nrf_lerf_sigma_exact_lm_strided", "nrf_lerf_render_embedding_lm_geo", "nrf_hash_backward_packed_workspace_bytes", "nrf_hash_backward_rays_packed", "nrf_hash_backward_binned_workspace_bytes", "nrf_hash_backward_binned_workspace_bytes_for", "nrf_hash_backward_rays_binned", "nrf_mlp_set_params", "nrf_mlp_device_repack_images", "nrf_mlp_set_input_rms_hint", "nrf_mlp_set_split_scaling", "nrf_mlp_get_split_scales", "nrf_renderer_nonfinite", "nrf_hash_memory_bytes", "nrf_lerf_renderer_nonfinite",
    "nrf_hash_backward", "nrf_hash_backward_rays", "nrf_hash_tv_loss", "nrf_adam_step", "nrf_adam_step_guarded", "nrf_renderer_last_features", "nrf_mlp_backward_f16_lm_src", "nrf_mask_sigma_grad_src", "nrf_mlp_backward_f16_flags_async", "nrf_mlp_backward_f16_flags_device",
    "nrf_render_view_dims",
    "nrf_tile_partition", "nrf_comm_unique_id", "nrf_comm_create", "nrf_comm_create_timeout", "nrf_comm_wrap", "nrf_comm_destroy", "nrf_comm_world", "nrf_comm_rank", "nrf_allgather_tiles", "nrf_allreduce_grads",
    "nrf_profile_enable", "nrf_profile_is_enabled", "nrf_profile_read", "nrf_set_render_lanes", "nrf_get_render_lanes", "nrf_renderer_set_lanes", "nrf_lerf_renderer_set_lanes",
    "nrf_lerf_relevancy", "nrf_relevancy_image", "nrf_colormap_jet_u8", "nrf_colormap_jet_lut",
    "nrf_lerf_renderer_create", "nrf_lerf_renderer_destroy", "nrf_lerf_set_prompts", "nrf_lerf_render_rays_workspace_bytes", "nrf_lerf_render_rays",
    "nrf_lerf_batchify_rays_workspace_bytes", "nrf_lerf_batchify_rays", "nrf_lerf_render_rows_workspace_bytes", "nrf_lerf_render_rows",
    "nrf_fp32_gemm_available", "nrf_get_train_gemm", "nrf_set_train_gemm", "nrf_gemm_nt_bf16x3", "nrf_gemm_nt_f16x3", "nrf_gemm_tn_bf16x3", "nrf_layer_grad_split", "nrf_huber_rows_nanmean", "nrf_lerf_head_backward_workspace_bytes", "nrf_lerf_head_backward", "nrf_lerf_backward_points_workspace_bytes", "nrf_lerf_backward_points",
    "nrf_lerf_renderer_last_features", "nrf_lerf_backward_points_src", "nrf_scratch_trim",
]
NRF_COMM_ID_BYTES = 128

_lib = None


class NrfError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NrfError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH)
        L.nrf_last_error.restype = C.c_char_p
        L.nrf_status_string.restype = C.c_char_p
        L.nrf_hash_table_elems.restype = C.c_int64
        L.nrf_mlp_backward_f16_flags_device.restype = C.c_void_p
        L.nrf_hash_get_dense_budget.restype = C.c_int64
        L.nrf_mlp_small_param_count.restype = C.c_int64
        L.nrf_mlp_nerf_param_count.restype = C.c_int64
        L.nrf_mlp_lerf_param_count.restype = C.c_int64
        L.nrf_scratch_trim.restype = C.c_size_t
        L.nrf_run_network_workspace_bytes.restype = C.c_size_t
        L.nrf_render_rays_workspace_bytes.restype = C.c_size_t
        L.nrf_batchify_rays_workspace_bytes.restype = C.c_size_t
        L.nrf_render_rows_workspace_bytes.restype = C.c_size_t
        L.nrf_lerf_render_rays_workspace_bytes.restype = C.c_size_t
        L.nrf_lerf_batchify_rays_workspace_bytes.restype = C.c_size_t
        L.nrf_lerf_render_rows_workspace_bytes.restype = C.c_size_t
        L.nrf_mlp_backward_workspace_bytes.restype = C.c_size_t
        L.nrf_mlp_backward_f16_workspace_bytes.restype = C.c_size_t
        L.nrf_hash_backward_binned_workspace_bytes.restype = C.c_size_t
        L.nrf_hash_backward_binned_workspace_bytes_for.restype = C.c_size_t
        L.nrf_lerf_geo_bytes.restype = C.c_size_t
        L.nrf_hash_backward_packed_workspace_bytes.restype = C.c_size_t
        L.nrf_lerf_head_backward_workspace_bytes.restype = C.c_size_t
        L.nrf_lerf_backward_points_workspace_bytes.restype = C.c_size_t
        _lib = L
    return _lib


def check(status):
    if status != NRF_OK:
        L = lib()
        raise NrfError(f"{L.nrf_status_string(status).decode()}: {L.nrf_last_error().decode()}")
