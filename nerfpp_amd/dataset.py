"""Ray-batch producer of the training loop -- NeRFDataset::get_batch / GetRayBatch (NeRFDataset.cpp:44-65, :109-208) and the Blender
camera / bounds helpers the executor calls once per dataset (load_blender.h:83-124) -- SURVEY section 8f, row N2.

Image decoding, COLMAP / Blender file parsing and the CLIP pyramid are out of scope (loaders); a view here is the already
decoded record the reference keeps in NeRFDatasetParams::Views: H, W, K [3,3], Pose [3|4, 4] and the image as an fp32 [H, W, 3] tensor.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .modules import _ptr, _stream, _dev_f32


@dataclass
class View:                       # NeRFDatasetParams::Views[i]
    H: int
    W: int
    K: np.ndarray                 # [3,3]
    Pose: np.ndarray              # [3,4] or [4,4] camera-to-world
    Image: Optional[torch.Tensor] = None      # [H, W, 3] fp32 on the GPU
    Near: float = 0.0
    Far: float = 0.0


def _k9(K):
    return np.ascontiguousarray(K.detach().cpu().numpy() if torch.is_tensor(K) else K, np.float32).reshape(9)


def _m12(c2w):
    m = c2w.detach().cpu().numpy() if torch.is_tensor(c2w) else np.asarray(c2w)
    return np.ascontiguousarray(m[:3, :4], np.float32).reshape(12)


def GetRayBatch(rand_h, rand_w, H, W, K, c2w):
    """NeRFDataset::GetRayBatch (NeRFDataset.cpp:109-145): rand_h / rand_w int64 device tensors [N] -> rays_o [N,3], rays_d [N,3], cone_angle."""
    rh = rand_h.to(torch.int64).contiguous(); rw = rand_w.to(torch.int64).contiguous()
    assert rh.is_cuda and rw.is_cuda and rh.shape == rw.shape
    n = rh.numel()
    o = torch.empty((n, 3), device=rh.device, dtype=torch.float32); d = torch.empty_like(o)
    cone = C.c_float(0)
    k, m = _k9(K), _m12(c2w)
    L.check(L.lib().nrf_ray_batch(k.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), _ptr(rh), _ptr(rw), C.c_int64(n), _ptr(o), _ptr(d), C.byref(cone), _stream()))
    return o, d, torch.tensor(cone.value, dtype=torch.float32)


def CalculateBounds(h, w, current_iter, precorp_iters, precorp_frac):
    """NeRFDataset::CalculateBounds (NeRFDataset.cpp:44-65) -> (h_start, h_end, w_start, w_end), inclusive."""
    out = (C.c_int * 4)()
    L.check(L.lib().nrf_precrop_bounds(int(h), int(w), int(current_iter), int(precorp_iters), C.c_float(precorp_frac), out))
    return tuple(out)


class NeRFDataset:
    """The get_batch() side of NeRFDataset (NeRFDataset.cpp:148-208) for already decoded views.  The reference draws pixel coordinates
    with torch::randint from the global generator; here they are a pure function of (seed, iteration, element) (include/nrf_rng.h), so
    a training run is reproducible and a batch can be regenerated from its iteration number alone."""

    def __init__(self, views, batch_size, precorp_iters=0, precorp_frac=0.5, seed=0):
        self.Views, self.BatchSize, self.PrecorpIters, self.PrecorpFrac, self.Seed = list(views), int(batch_size), int(precorp_iters), float(precorp_frac), int(seed)
        self.CurrentIter, self.CurrentImageIdx = 0, 0

    def SetCurrentIter(self, i):
        self.CurrentIter = int(i)
        self.CurrentImageIdx = int(i) % len(self.Views)       # the reference cycles through prefetched images (PrefetchNextImage)

    def get_batch(self):
        v = self.Views[self.CurrentImageIdx]
        h0, h1, w0, w1 = CalculateBounds(v.H, v.W, self.CurrentIter, self.PrecorpIters, self.PrecorpFrac)
        n = self.BatchSize
        rh = torch.empty((n,), device="cuda", dtype=torch.int64); rw = torch.empty_like(rh)
        L.check(L.lib().nrf_rand_pixels(C.c_uint64(self.Seed), C.c_int64(self.CurrentIter), h0, h1, w0, w1, C.c_int64(n), _ptr(rh), _ptr(rw), _stream()))
        target = None
        if v.Image is not None:
            img = _dev_f32(v.Image)
            c = img.shape[-1]
            target = torch.empty((n, c), device=img.device, dtype=torch.float32)
            L.check(L.lib().nrf_gather_pixels(_ptr(img), v.H, v.W, c, _ptr(rh), _ptr(rw), C.c_int64(n), _ptr(target), _stream()))
        o, d, cone = GetRayBatch(rh, rw, v.H, v.W, v.K, v.Pose)
        return dict(rays_o=o, rays_d=d, cone_angle=cone, Near=v.Near, Far=v.Far, target_s=target, rand_h=rh, rand_w=rw)


def GetBoundsForObj(views):
    """load_blender.h:83-96: (near, far) = (0.15, 0.6) x the diagonal of the box around the training cameras' origins."""
    org = np.stack([np.asarray(v.Pose, np.float32)[:3, 3] for v in views])
    diff = (org.max(0) - org.min(0)).astype(np.float32)
    d = np.float32(np.sqrt(np.float32(np.sum(diff * diff, dtype=np.float32))))
    return float(np.float32(0.15 * float(d))), float(np.float32(0.6 * float(d)))


def GetBbox3dForObj(views):
    """load_blender.h:99-124: the box around o + Near*d and o + Far*d of the four corner rays of every training view."""
    mn = np.full(3, 1e8, np.float32); mx = np.full(3, -1e8, np.float32)
    for v in views:
        rh = torch.tensor([0, 0, v.H - 1, v.H - 1], device="cuda"); rw = torch.tensor([0, v.W - 1, 0, v.W - 1], device="cuda")
        o, d, _ = GetRayBatch(rh, rw, v.H, v.W, v.K, v.Pose)
        o, d = o.cpu().numpy(), d.cpu().numpy()
        for t in (np.float32(v.Near), np.float32(v.Far)):
            p = o + t * d
            mn = np.minimum(mn, p.min(0)); mx = np.maximum(mx, p.max(0))
    return np.concatenate([mn, mx]).astype(np.float32)
