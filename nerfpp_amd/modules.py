"""Host-side mirror of the reference's encoder / MLP plugin surface over the C ABI.

Class and method names follow the reference (BaseEmbedder.h, NeRF.h, CuHashEmbedder.h, CuSHEncoder.h, LeRF.h) so the
parity tests read like calls into the reference.  PyTorch is used only for device memory and streams: every forward is
ONE call into libnerfpp_hip.so on the current HIP stream.  Nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t):
    assert t.is_cuda, "inputs must live on the GPU (no CPU fallback)"
    return t.contiguous().float()


# ------------------------------------------------------------------------------------------------
# BaseEmbedderImpl                                                           BaseEmbedder.h:6-15
# ------------------------------------------------------------------------------------------------
class BaseEmbedder:
    def GetOutputDims(self):
        return 0

    def forward(self, x):
        """-> (embedding [P, D] fp32, keep_mask [P] bool or None)"""
        raise NotImplementedError

    __call__ = lambda self, x: self.forward(x)


class Embedder(BaseEmbedder):
    """Sinusoidal positional encoding, EmbedderImpl (NeRF.h:12-31, NeRF.cpp:4-39): Embedder(name, multires)."""

    def __init__(self, module_name, multires):
        self.name, self.multires = module_name, int(multires)

    def GetOutputDims(self):
        return 3 + 6 * self.multires

    def forward(self, x):
        x = _dev_f32(x).reshape(-1, 3)
        out = torch.empty((x.shape[0], self.GetOutputDims()), device=x.device, dtype=torch.float32)
        L.check(L.lib().nrf_pe_encode(_ptr(x), C.c_int64(x.shape[0]), self.multires, _ptr(out), _stream()))
        return out, None


class SHEncoder(BaseEmbedder):
    """LibTorch spherical harmonics, SHEncoderImpl (NeRF.h:80-132, NeRF.cpp:131-201), degree 1..5."""
    variant = L.NRF_SH_LIBTORCH

    def __init__(self, module_name, input_dim=3, degree=4):
        assert input_dim == 3
        self.name, self.degree = module_name, int(degree)

    def GetOutputDims(self):
        return self.degree * self.degree

    def forward(self, x):
        x = _dev_f32(x).reshape(-1, 3)
        out = torch.empty((x.shape[0], self.GetOutputDims()), device=x.device, dtype=torch.float32)
        L.check(L.lib().nrf_sh_encode(_ptr(x), C.c_int64(x.shape[0]), self.degree, self.variant, _ptr(out), _stream()))
        return out, None


class CuSHEncoder(SHEncoder):
    """CUDA spherical harmonics, CuSHEncoderImpl (CuSHEncoder.h:6-29, CuSHEncoder.cu:4-118), degree 1..8."""
    variant = L.NRF_SH_CUDA


class _HashBase(BaseEmbedder):
    mode = None

    def __init__(self, module_name, bounding_box, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                 finest_resolution=512):
        self.name = module_name
        bb = np.asarray(bounding_box.detach().cpu().numpy() if torch.is_tensor(bounding_box) else bounding_box, np.float32).reshape(6)
        self.BoundingBox = bb
        self.NLevels, self.NFeaturesPerLevel, self.Log2HashmapSize = int(n_levels), int(n_features_per_level), int(log2_hashmap_size)
        self.BaseResolution, self.FinestResolution = int(base_resolution), int(finest_resolution)
        desc = L.HashDesc(self.mode, self.NLevels, self.NFeaturesPerLevel, self.Log2HashmapSize, self.BaseResolution, self.FinestResolution,
                          (C.c_float * 6)(*bb.tolist()))
        self._h = C.c_void_p()
        L.check(L.lib().nrf_hash_create(C.byref(desc), C.byref(self._h)))
        self.dense_budget = 24 << 30          # the library's default bake budget (encode.h); set_dense_budget keeps this in step

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.lib().nrf_hash_destroy(h)
            except Exception:       # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    def GetOutputDims(self):
        return self.NLevels * self.NFeaturesPerLevel

    def GetBoundingBox(self):
        return self.BoundingBox

    def table_elems(self):
        return self.NLevels * (1 << self.Log2HashmapSize) * self.NFeaturesPerLevel

    def set_dense_budget(self, nbytes):
        """Bytes of baked dense image for the coarse levels of the renderer's fast path (nrf_hash_set_dense_budget; 0 = every level hashed: what a
        training loop, which re-uploads the table every step, wants)."""
        L.check(L.lib().nrf_hash_set_dense_budget(self._h, C.c_int64(int(nbytes)), _stream()))
        self.dense_budget = int(nbytes)

    def level_scales(self):
        """The per-level position scales in use (CuHashEmbedder: mul_l of CuHashEmbedder.cu:40; HashEmbedder: the floor()ed resolutions)."""
        out = np.empty(self.NLevels, np.float32)
        L.check(L.lib().nrf_hash_get_level_scales(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_level_scales(self, scales):
        """CuHashEmbedder only: replace the host-libm mul_l by the values a particular CUDA build computes on its device (nrf_hash_set_level_scales)."""
        a = np.ascontiguousarray(scales, np.float32)
        assert a.shape == (self.NLevels,)
        L.check(L.lib().nrf_hash_set_level_scales(self._h, a.ctypes.data_as(C.c_void_p), _stream()))

    def set_table(self, table):
        """fp32 embedding table in the reference's parameter layout (numpy or torch, host or device)."""
        if torch.is_tensor(table) and table.is_cuda:
            t = table.contiguous().float().reshape(-1)
            assert t.numel() == self.table_elems()
            L.check(L.lib().nrf_hash_set_table(self._h, _ptr(t), 1, _stream()))       # stream-ordered: the cast (and a re-bake into an existing image) queue behind `t`
        else:
            a = np.ascontiguousarray(table.detach().cpu().numpy() if torch.is_tensor(table) else table, np.float32).reshape(-1)
            assert a.size == self.table_elems()
            L.check(L.lib().nrf_hash_set_table(self._h, a.ctypes.data_as(C.c_void_p), 0, _stream()))
            torch.cuda.current_stream().synchronize()

    def forward(self, x):
        x = _dev_f32(x).reshape(-1, 3)
        out = torch.empty((x.shape[0], self.GetOutputDims()), device=x.device, dtype=torch.float32)
        mask = torch.empty((x.shape[0],), device=x.device, dtype=torch.uint8)
        L.check(L.lib().nrf_hash_encode(self._h, _ptr(x), C.c_int64(x.shape[0]), _ptr(out), _ptr(mask), _stream()))
        return out, mask.bool()


class HashEmbedder(_HashBase):
    """LibTorch hash grid, HashEmbedderImpl (NeRF.h:136-209, NeRF.cpp:208-318).  Table: embeddings_0..L-1, each [2^T, F]."""
    mode = L.NRF_HASH_NGP


class CuHashEmbedder(_HashBase):
    """CUDA hash grid, CuHashEmbedderImpl (CuHashEmbedder.h:8-63, CuHashEmbedder.cpp, CuHashEmbedder.cu).
    Table `embedder_embeddings` [L*2^T, F] fp32 master (cast to fp16 once at upload); per-level primes/biases."""
    mode = L.NRF_HASH_CU

    def set_primes(self, primes, biases=None):
        p = np.ascontiguousarray(primes, np.int32).reshape(-1)
        assert p.size == self.NLevels * 3
        b = None
        if biases is not None:
            b = np.ascontiguousarray(biases, np.float32).reshape(-1)
            assert b.size == self.NLevels * 3
        self.Primes, self.Biases = p, b
        L.check(L.lib().nrf_hash_set_primes(self._h, p.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None))


# ------------------------------------------------------------------------------------------------
# BaseNeRFImpl                                                               NeRF.h:33-42
# ------------------------------------------------------------------------------------------------
class BaseNeRF:
    precision = L.NRF_PREC_F32
    _m = None

    def __del__(self):
        m = getattr(self, "_m", None)
        if m:
            try:
                L.lib().nrf_mlp_destroy(m)
            except Exception:       # interpreter shutdown: module globals may already be gone
                pass
            self._m = None

    def _blob(self, params):
        if isinstance(params, (list, tuple)):   # [(name, array)] in named_parameters() order
            params = np.concatenate([np.asarray(a, np.float32).reshape(-1) for _, a in params])
        if torch.is_tensor(params):
            params = params.detach().cpu().numpy()
        return np.ascontiguousarray(params, np.float32).reshape(-1)

    def GetOutputDims(self):
        return L.lib().nrf_mlp_output_dims(self._m)

    def forward(self, x, precision=None):
        x = _dev_f32(x)
        x = x.reshape(-1, x.shape[-1])
        out = torch.empty((x.shape[0], self.GetOutputDims()), device=x.device, dtype=torch.float32)
        prec = self.precision if precision is None else precision
        L.check(L.lib().nrf_mlp_forward(self._m, _ptr(x), C.c_int64(x.shape[0]), prec, _ptr(out), _stream()))
        return out

    __call__ = forward


class NeRFSmall(BaseNeRF):
    """NeRFSmallImpl (NeRF.h:212-254, NeRF.cpp:322-412).  use_pred_normal adds the predicted-normals head (NeRF.cpp:343-347, :393-407; the executor builds it only
    when n_importance == 0, NeRFExecutor.h:487): output [rgb, sigma, normal xyz], NRF_PREC_F32 only."""

    def __init__(self, num_layers=3, hidden_dim=64, geo_feat_dim=15, num_layers_color=4, hidden_dim_color=64, use_pred_normal=False,
                 num_layers_normals=3, hidden_dim_normals=64, input_ch=3, input_ch_views=3, module_name="hashnerf", params=None):
        self.desc = L.MlpSmallDesc(input_ch, input_ch_views, num_layers, hidden_dim, geo_feat_dim, num_layers_color, hidden_dim_color,
                                   int(bool(use_pred_normal)), int(num_layers_normals) if use_pred_normal else 0, int(hidden_dim_normals) if use_pred_normal else 0)
        self.n_params = L.lib().nrf_mlp_small_param_count(C.byref(self.desc))
        if params is not None:
            self.load(params)

    def load(self, params):
        blob = self._blob(params)
        assert blob.size == self.n_params, (blob.size, self.n_params)
        self.__del__()
        self._m = C.c_void_p()
        L.check(L.lib().nrf_mlp_small_create(C.byref(self.desc), blob.ctypes.data_as(C.c_void_p), 0, _stream(), C.byref(self._m)))


class NeRF(BaseNeRF):
    """NeRFImpl (NeRF.h:44-77, NeRF.cpp:41-126); skips = {4}."""

    def __init__(self, d=8, w=256, input_ch=3, input_ch_views=3, output_ch=4, skips=(4,), use_viewdirs=False, module_name="nerf", params=None):
        skips = tuple(skips)
        assert len(skips) <= 1, "the reference only ever builds skips = {4}"
        self.desc = L.MlpNerfDesc(d, w, input_ch, input_ch_views, output_ch, skips[0] if skips else -1, int(bool(use_viewdirs)))
        self.n_params = L.lib().nrf_mlp_nerf_param_count(C.byref(self.desc))
        if params is not None:
            self.load(params)

    def load(self, params):
        blob = self._blob(params)
        assert blob.size == self.n_params, (blob.size, self.n_params)
        self.__del__()
        self._m = C.c_void_p()
        L.check(L.lib().nrf_mlp_nerf_create(C.byref(self.desc), blob.ctypes.data_as(C.c_void_p), 0, _stream(), C.byref(self._m)))


class LeRF(BaseNeRF):
    """LeRFImpl (LeRF.h:6-31, LeRF.cpp): LeRF(geo_feat_dim_le, num_layers_le, hidden_dim_le, lang_embed_dim, input_ch_le)."""

    def __init__(self, geo_feat_dim_le=32, num_layers_le=3, hidden_dim_le=64, lang_embed_dim=768, input_ch_le=0, module_name="lerf", params=None):
        self.desc = L.MlpSmallDesc(input_ch_le, 0, num_layers_le, hidden_dim_le, geo_feat_dim_le, num_layers_le, lang_embed_dim)
        self.n_params = L.lib().nrf_mlp_lerf_param_count(C.byref(self.desc))
        self.LangEmbedDim = lang_embed_dim
        if params is not None:
            self.load(params)

    def GetLangEmbedDim(self):
        return self.LangEmbedDim

    def load(self, params):
        blob = self._blob(params)
        assert blob.size == self.n_params, (blob.size, self.n_params)
        self.__del__()
        self._m = C.c_void_p()
        L.check(L.lib().nrf_mlp_lerf_create(C.byref(self.desc), blob.ctypes.data_as(C.c_void_p), 0, _stream(), C.byref(self._m)))
