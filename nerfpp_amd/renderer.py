"""Host-side mirror of RayUtils.h / Sampler.h / NeRFRenderer.h over the C ABI.

Same names, argument meaning and quirks as the reference so the parity tests read like the reference's own calls:
GetRays, NDCRays, IntersectWithAABB, SamplePDF, NeRFRenderParams, NeRFRendererOutputs, NeRFRenderResult and
NeRFRenderer.{Render, BatchifyRays, RenderRays, RunNetwork, RawToOutputs}.  All arithmetic happens in
libnerfpp_hip.so on the current HIP stream; torch only owns the buffers.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .modules import Embedder, SHEncoder, CuSHEncoder, _HashBase, _ptr, _stream, _dev_f32

ATEN_SUM_VEC = 8   # fp32 lanes of ATen's CPU sum kernel on any AVX2+/AVX-512 x86 host (REGISTER_DISPATCH registers the AVX2 build)


def _host_f32(t, n):
    a = np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, np.float32).reshape(-1)
    assert n is None or a.size == n
    return a


# ------------------------------------------------------------------------------------------------
# RayUtils.h
# ------------------------------------------------------------------------------------------------
def GetRays(h, w, k, c2w, device="cuda", row0=0, rows=None):
    """RayUtils.h:23-46 -> (rays_o [rows,w,3], rays_d [rows,w,3], cone_angle 0-dim).  row0/rows select an image row tile
    (multi-GPU sharding); ray index stays row-major, y outer."""
    rows = h - row0 if rows is None else rows
    K = _host_f32(k, 9); M = _host_f32(torch.as_tensor(c2w)[:3, :4] if torch.is_tensor(c2w) else np.asarray(c2w)[:3, :4], 12)
    o = torch.empty((rows, w, 3), device=device, dtype=torch.float32)
    d = torch.empty((rows, w, 3), device=device, dtype=torch.float32)
    cone = C.c_float(0)
    L.check(L.lib().nrf_get_rays(h, w, K.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p), row0, rows, _ptr(o), _ptr(d), C.byref(cone), _stream()))
    return o, d, torch.tensor(cone.value, dtype=torch.float32)


def NDCRays(h, w, focal, near, rays_o, rays_d, cone_angle=None):
    """RayUtils.h:49-83.  cone_angle comes back unchanged: the reference multiplies it by |d_ndc| / |rays_d| after rays_d has become d_ndc (:73-81), i.e. by
    exactly 1.0 per ray (its [.., 1] tensor holds the scalar's value everywhere)."""
    o = _dev_f32(rays_o); d = _dev_f32(rays_d)
    oo = torch.empty_like(o); od = torch.empty_like(d)
    L.check(L.lib().nrf_ndc_rays(h, w, C.c_float(focal), C.c_float(near), _ptr(o), _ptr(d), C.c_int64(o.numel() // 3), _ptr(oo), _ptr(od), _stream()))
    return oo, od, cone_angle


def IntersectWithAABB(rays_o, rays_d, bounding_box, near_plane=0.0):
    """RayUtils.h:87-126 -> (nears [N], fars [N])."""
    o = _dev_f32(rays_o).reshape(-1, 3); d = _dev_f32(rays_d).reshape(-1, 3)
    bb = _host_f32(bounding_box, 6)
    nr = torch.empty((o.shape[0],), device=o.device, dtype=torch.float32); fr = torch.empty_like(nr)
    L.check(L.lib().nrf_aabb(_ptr(o), _ptr(d), bb.ctypes.data_as(C.c_void_p), C.c_int64(o.shape[0]), C.c_float(near_plane), _ptr(nr), _ptr(fr), _stream()))
    return nr, fr


# ------------------------------------------------------------------------------------------------
# Sampler.h
# ------------------------------------------------------------------------------------------------
def SamplePDF(bins, weights, nsamples, det=True, return_inds=False, sum_vec=ATEN_SUM_VEC, u=None, seed=0, ray_base=0):
    """Sampler.h:6-43 (det == (Perturb == 0), NeRFRenderer.h:428).  det=False: `u` [N, nsamples] are the uniform draws (the reference's
    torch::rand tensor); when None they come from the library's counter RNG (seed, NRF_RNG_U_PDF, global element index)."""
    bins = _dev_f32(bins); weights = _dev_f32(weights)
    n, nb = bins.shape
    assert weights.shape == (n, nb - 1)
    samples = torch.empty((n, nsamples), device=bins.device, dtype=torch.float32)
    inds = torch.empty((n, nsamples), device=bins.device, dtype=torch.int64) if return_inds else None
    if det:
        u = torch.linspace(0.0, 1.0, nsamples, dtype=torch.float32).to(bins.device)     # Sampler.h:21, ATen's own rounding
        L.check(L.lib().nrf_sample_pdf(_ptr(bins), _ptr(weights), C.c_int64(n), nb, _ptr(u), nsamples, sum_vec, _ptr(samples), _ptr(inds), _stream()))
    else:
        u = RngFill(seed, L.NRF_RNG_U_PDF, ray_base * nsamples, n * nsamples, device=bins.device).reshape(n, nsamples) if u is None else _dev_f32(u)
        assert u.shape == (n, nsamples)
        L.check(L.lib().nrf_sample_pdf_rand(_ptr(bins), _ptr(weights), C.c_int64(n), nb, _ptr(u), nsamples, sum_vec, _ptr(samples), _ptr(inds), _stream()))
    return (samples, inds) if return_inds else samples


def RngFill(seed, rng_stream, index0, count, normal=False, device="cuda"):
    """include/nrf_rng.h: element k = draw(seed, rng_stream, index0 + k); uniform on [0,1) or standard normal."""
    out = torch.empty((int(count),), device=device, dtype=torch.float32)
    L.check(L.lib().nrf_rng_fill(C.c_uint64(int(seed)), C.c_uint32(int(rng_stream)), C.c_uint64(int(index0)), C.c_int64(int(count)), int(normal), _ptr(out), _stream()))
    return out


def JitterZ(z_vals, t_rand):
    """NeRFRenderer.h:404-417 with explicit uniform draws t_rand [N,S]."""
    z = _dev_f32(z_vals); t = _dev_f32(t_rand)
    out = torch.empty_like(z)
    L.check(L.lib().nrf_jitter_z(_ptr(z), _ptr(t), C.c_int64(z.shape[0]), z.shape[1], _ptr(out), _stream()))
    return out


def TangentScatter(pts, z_vals, cone_angle, rays_d, bounding_box=None, u_r=None, u_theta=None):
    """NeRFRenderer.h:307-362 with explicit uniform draws u_r, u_theta [N,S(,1)] (the reference's two torch::rand tensors)."""
    pts = _dev_f32(pts); z = _dev_f32(z_vals); d = _dev_f32(rays_d).reshape(-1, 3)
    n, s = z.shape
    rays = torch.cat([torch.zeros_like(d), d], -1).contiguous()          # only columns 3..5 (rays_d) are read when pts are explicit
    bb = _host_f32(bounding_box, 6) if bounding_box is not None else None
    out = torch.empty_like(pts)
    L.check(L.lib().nrf_tangent_scatter(_ptr(pts), _ptr(rays), 6, _ptr(z), C.c_int64(n), s, C.c_float(float(cone_angle)), _ptr(_dev_f32(u_r).reshape(n, s)),
                                        _ptr(_dev_f32(u_theta).reshape(n, s)), bb.ctypes.data_as(C.c_void_p) if bb is not None else None, _ptr(out), _stream()))
    return out


def StochasticPrecondition(pts, noise, alpha, bounding_box):
    """NeRFRenderer.h:433-443 + ReflectBoundary :285-304: reflect(pts + noise*alpha)."""
    pts = _dev_f32(pts); nz = _dev_f32(noise)
    bb = _host_f32(bounding_box, 6)
    out = torch.empty_like(pts)
    L.check(L.lib().nrf_precondition(_ptr(pts), _ptr(nz), C.c_float(alpha), bb.ctypes.data_as(C.c_void_p), C.c_int64(pts.numel() // 3), _ptr(out), _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# NeRFRenderer.h
# ------------------------------------------------------------------------------------------------
@dataclass
class NeRFRendererOutputs:            # NeRFRenderer.h:12-18
    RGBMap: Optional[torch.Tensor] = None
    DispMap: Optional[torch.Tensor] = None
    AccMap: Optional[torch.Tensor] = None
    Weights: Optional[torch.Tensor] = None
    DepthMap: Optional[torch.Tensor] = None


class NeRFRenderResult:               # NeRFRenderer.h:20-26
    """Outputs, Raw, Near, Far as in the reference.  Near / Far are float scalars there (two .item() host syncs per frame, NeRFRenderer.h:602-603); here the
    library leaves them in a device [2] buffer on the render's stream and the properties read it on first access, so a render call never stalls the host."""

    def __init__(self):
        self.Outputs = NeRFRendererOutputs()
        self.Raw = None
        self.Extras = {}          # intermediates exposed for stage-chained parity tests (not in the reference struct)
        self.FeatureView = None   # NeRFRenderer.feature_view() of the render call that produced this result (ray-batch branch), or None
        self._nf = (0.0, 0.0)
        self._nf_dev = None

    def _near_far(self):
        if self._nf_dev is not None:
            a = self._nf_dev.tolist()             # synchronises the producing stream
            self._nf, self._nf_dev = (a[0], a[1]), None
        return self._nf

    Near = property(lambda self: self._near_far()[0], lambda self, v: setattr(self, "_nf", (float(v), self._near_far()[1])))
    Far = property(lambda self: self._near_far()[1], lambda self, v: setattr(self, "_nf", (self._near_far()[0], float(v))))


@dataclass
class NeRFRenderParams:               # NeRFRenderer.h:28-44 (same defaults)
    NSamples: int = 64
    NImportance: int = 192
    Chunk: int = 1024 * 32
    ReturnRaw: bool = False
    LinDisp: bool = False
    Perturb: float = 0.0
    WhiteBkgr: bool = False
    RawNoiseStd: float = 0.0
    Ndc: bool = True
    UseViewdirs: bool = False
    ReturnWeights: bool = False
    ThinRay: bool = False
    RenderFactor: float = 0
    BoundingBox: Optional[object] = None
    StochasticPreconditioningAlpha: float = 0.0
    # not in the reference: MLP arithmetic (L.NRF_PREC_F32 parity mode / L.NRF_PREC_F16_MFMA fast mode)
    Precision: int = L.NRF_PREC_F32
    KeepIntermediates: object = False     # True: z / raw / weights of the coarse pass + z_fine in Extras; "depths": the same without raw_coarse
    # not in the reference: how the coarse pass is evaluated when NImportance > 0 (L.NRF_COARSE_*, include/nerfpp_hip.h).  AUTO: with
    # NRF_PREC_F16_SPLIT on the HashNeRF fast path, sigma net only in exact fp32 on the matrix cores -> the fine sample set of NRF_PREC_F32
    CoarseMode: int = L.NRF_COARSE_AUTO
    # not in the reference: what a matrix-core precision does when its network outputs are not finite (an fp16 operand left its range; L.NRF_OVERFLOW_*,
    # include/nerfpp_hip.h).  AUTO: one host synchronisation at the end of the render call, flagged chunks rendered again in NRF_PREC_F32; DEFERRED: no
    # synchronisation, the NEXT render call (or NeRFRenderer.nonfinite()) reports it
    OverflowPolicy: int = L.NRF_OVERFLOW_AUTO
    # not in the reference (it draws from torch's global RNG): seed of the counter-based draws of the stochastic branches
    Seed: int = 0


class NeRFRenderer:
    """NeRFRenderer<TEmbedder, TEmbedDirs, TNeRF> (NeRFRenderer.h:88-159)."""

    def __init__(self, embed_fn, embeddirs_fn, nerf):
        self.EmbedFn, self.EmbeddirsFn, self.NeRF = embed_fn, embeddirs_fn, nerf
        desc = L.RendererDesc()
        if isinstance(embed_fn, _HashBase):
            desc.hash = embed_fn._h; desc.pe_freqs = 0
        elif isinstance(embed_fn, Embedder):
            desc.hash = None; desc.pe_freqs = embed_fn.multires
        else:
            raise L.NrfError(f"unsupported position embedder {type(embed_fn).__name__}")
        if embeddirs_fn is None:
            desc.dirs_encoder, desc.dirs_param = L.NRF_DIRS_NONE, 0
        elif isinstance(embeddirs_fn, Embedder):
            desc.dirs_encoder, desc.dirs_param = L.NRF_DIRS_PE, embeddirs_fn.multires
        elif isinstance(embeddirs_fn, CuSHEncoder):
            desc.dirs_encoder, desc.dirs_param = L.NRF_DIRS_SH_CUDA, embeddirs_fn.degree
        elif isinstance(embeddirs_fn, SHEncoder):
            desc.dirs_encoder, desc.dirs_param = L.NRF_DIRS_SH_LIBTORCH, embeddirs_fn.degree
        else:
            raise L.NrfError(f"unsupported direction embedder {type(embeddirs_fn).__name__}")
        desc.mlp = nerf._m
        self._r = C.c_void_p()
        L.check(L.lib().nrf_renderer_create(C.byref(desc), C.byref(self._r)))
        self._ws = None

    def __del__(self):
        r = getattr(self, "_r", None)
        if r:
            try:
                L.lib().nrf_renderer_destroy(r)
            except Exception:       # interpreter shutdown: module globals may already be gone
                pass
            self._r = None

    def feature_view(self):
        """nrf_renderer_last_features: where the most recent single-chunk render of the feature-reusing fast path left the hash features of its fine depths in this
        renderer's workspace -- dict(feats, cols, keep, src: device addresses; n, sf; serial) or None.  Valid until the next render call on this renderer."""
        if not getattr(self, "_r", None):
            return None
        fp, kp, sp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cols, nn, sf, ser = C.c_int64(), C.c_int64(), C.c_int(), C.c_uint64()
        rc = L.lib().nrf_renderer_last_features(self._r, C.byref(fp), C.byref(cols), C.byref(kp), C.byref(sp), C.byref(nn), C.byref(sf), C.byref(ser))
        if rc != 0:
            return None
        return dict(feats=fp.value, cols=int(cols.value), keep=kp.value, src=sp.value, n=int(nn.value), sf=int(sf.value), serial=int(ser.value))

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty((int(nbytes),), device=device, dtype=torch.uint8)
        return self._ws

    def nonfinite(self):
        """(chunks whose matrix-core render produced non-finite network outputs, chunks rendered again in NRF_PREC_F32) since this renderer was built
        (nrf_renderer_nonfinite: completes a pending OverflowPolicy DEFERRED check first)."""
        a, b = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().nrf_renderer_nonfinite(self._r, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- protected virtuals of the reference ----
    def RunNetwork(self, inputs, view_dirs, precision=L.NRF_PREC_F32):
        """NeRFRenderer.h:164-194: inputs [N,S,3], view_dirs [N,3] or None -> raw [N,S,C]."""
        pts = _dev_f32(inputs)
        n, s = pts.shape[0], pts.shape[1]
        vd = _dev_f32(view_dirs) if view_dirs is not None and view_dirs.numel() else None
        c = self.NeRF.GetOutputDims()
        raw = torch.empty((n, s, c), device=pts.device, dtype=torch.float32)
        nb = L.lib().nrf_run_network_workspace_bytes(self._r, C.c_int64(n), s)
        ws = self._workspace(nb, pts.device)
        L.check(L.lib().nrf_run_network(self._r, _ptr(pts), _ptr(vd), C.c_int64(n), s, precision, _ptr(raw), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
        return raw

    def RawToOutputs(self, raw, cone_angle, z_vals, rays_d, raw_noise_std=0.0, white_bkgr=False, noise=None):
        """NeRFRenderer.h:199-282.  raw_noise_std > 0 needs `noise` [N,S], the normal draws the reference takes from torch::randn_like."""
        raw = _dev_f32(raw); z = _dev_f32(z_vals); d = _dev_f32(rays_d)
        if raw_noise_std > 0:
            if noise is None:
                raise L.NrfError("RawToOutputs(raw_noise_std > 0): pass the normal draws as `noise` (RngFill(..., normal=True) or your own)")
            n, s, c = raw.shape
            o = NeRFRendererOutputs(RGBMap=torch.empty((n, 3), device=raw.device), DispMap=torch.empty((n,), device=raw.device),
                                    AccMap=torch.empty((n,), device=raw.device), Weights=torch.empty((n, s), device=raw.device),
                                    DepthMap=torch.empty((n,), device=raw.device))
            L.check(L.lib().nrf_raw2outputs_noise(_ptr(raw), _ptr(z), _ptr(d), 3, C.c_int64(n), s, c, int(white_bkgr), _ptr(_dev_f32(noise)),
                                                  C.c_float(raw_noise_std), _ptr(o.RGBMap), _ptr(o.DispMap), _ptr(o.AccMap), _ptr(o.Weights), _ptr(o.DepthMap), _stream()))
            return o
        n, s, c = raw.shape
        o = NeRFRendererOutputs(RGBMap=torch.empty((n, 3), device=raw.device), DispMap=torch.empty((n,), device=raw.device),
                                AccMap=torch.empty((n,), device=raw.device), Weights=torch.empty((n, s), device=raw.device),
                                DepthMap=torch.empty((n,), device=raw.device))
        L.check(L.lib().nrf_raw2outputs(_ptr(raw), _ptr(z), _ptr(d), 3, C.c_int64(n), s, c, int(white_bkgr), _ptr(o.RGBMap), _ptr(o.DispMap),
                                        _ptr(o.AccMap), _ptr(o.Weights), _ptr(o.DepthMap), _stream()))
        return o

    # ---- public surface ----
    _lin_cache = {}

    @classmethod
    def _linspace(cls, steps, dev):
        """torch::linspace(0, 1, steps) on the device (NeRFRenderer.h:393, Sampler.h:21), ATen's own rounding; cached per (steps, device)."""
        d = torch.device(dev)
        if d.type == "cuda" and d.index is None:          # 'cuda' means the CURRENT device: key on the indexed device, or a later set_device would get the wrong card's buffer
            d = torch.device("cuda", torch.cuda.current_device())
        key = (int(steps), d)
        t = cls._lin_cache.get(key)
        if t is None:
            if len(cls._lin_cache) >= 64:
                cls._lin_cache.clear()
            t = cls._lin_cache[key] = torch.linspace(0.0, 1.0, int(steps), dtype=torch.float32).to(d)
        return t

    @staticmethod
    def _params(n_samples, n_importance, cone_angle, lin_disp, perturb, white_bkgr, raw_noise_std, stochastic_preconditioning_alpha, bounding_box, precision,
                seed, ray_base, coarse_mode, overflow_policy=L.NRF_OVERFLOW_AUTO):
        rp = L.RenderParams(int(n_samples), int(n_importance), int(lin_disp), int(white_bkgr), precision, ATEN_SUM_VEC)
        rp.perturb, rp.raw_noise_std, rp.precond_alpha = float(perturb), float(raw_noise_std), float(stochastic_preconditioning_alpha)
        if cone_angle is not None and (not torch.is_tensor(cone_angle) or cone_angle.numel()):
            rp.has_cone, rp.cone_angle = 1, float(cone_angle)
        if bounding_box is not None:
            rp.has_bbox = 1
            rp.bbox = (C.c_float * 6)(*_host_f32(bounding_box, 6).tolist())
        rp.seed, rp.ray_base = int(seed), int(ray_base)
        rp.coarse_mode = int(coarse_mode)
        rp.overflow_policy = int(overflow_policy)
        return rp

    def _alloc_outputs(self, n, s, ni, dev, return_raw, return_weights, keep_intermediates):
        """(result, nrf_render_outputs) with every buffer sized for n rays."""
        sf = s + ni
        c = self.NeRF.GetOutputDims()
        res = NeRFRenderResult()
        so = sf if ni > 0 else s
        o = res.Outputs
        o.RGBMap = torch.empty((n, 3), device=dev); o.DispMap = torch.empty((n,), device=dev); o.AccMap = torch.empty((n,), device=dev)
        o.DepthMap = torch.empty((n,), device=dev)
        o.Weights = torch.empty((n, so), device=dev) if return_weights else None
        if return_raw:
            res.Raw = torch.empty((n, so, c), device=dev)
        ro = L.RenderOutputs(_ptr(o.RGBMap), _ptr(o.DispMap), _ptr(o.AccMap), _ptr(o.DepthMap), _ptr(o.Weights), _ptr(res.Raw), None, None, None, None)
        if keep_intermediates:
            ex = res.Extras
            ex["z_coarse"] = torch.empty((n, s), device=dev)
            ex["weights_coarse"] = torch.empty((n, s), device=dev)
            ro.d_z_coarse, ro.d_weights_coarse = _ptr(ex["z_coarse"]), _ptr(ex["weights_coarse"])
            if keep_intermediates != "depths":       # asking for the coarse raw forces the whole network on the coarse pass (nrf_render_params.coarse_mode)
                ex["raw_coarse"] = torch.empty((n, s, c), device=dev)
                ro.d_raw_coarse = _ptr(ex["raw_coarse"])
            if ni > 0:
                ex["z_fine"] = torch.empty((n, sf), device=dev)
                ro.d_z_fine = _ptr(ex["z_fine"])
        return res, ro

    def RenderRays(self, ray_batch, cone_angle, n_samples, return_raw=False, lin_disp=False, perturb=0.0, n_importance=0, white_bkgr=False,
                   raw_noise_std=0.0, stochastic_preconditioning_alpha=0.0, bounding_box=None, return_weights=True,
                   precision=L.NRF_PREC_F32, keep_intermediates=False, seed=0, ray_base=0, coarse_mode=L.NRF_COARSE_AUTO):
        """NeRFRenderer.h:366-459 for one chunk of packed rays [N, 8|11].  The stochastic branches (perturb, a defined cone_angle,
        raw_noise_std, stochastic preconditioning) draw from the library's counter RNG keyed by (seed, ray_base + ray, sample)."""
        rays = _dev_f32(ray_batch)
        n, stride = rays.shape
        dev = rays.device
        s, ni = int(n_samples), int(n_importance)
        t = self._linspace(s, dev)                                   # NeRFRenderer.h:393
        u = self._linspace(ni, dev) if ni > 0 else None               # Sampler.h:21
        rp = self._params(s, ni, cone_angle, lin_disp, perturb, white_bkgr, raw_noise_std, stochastic_preconditioning_alpha, bounding_box, precision, seed,
                          ray_base, coarse_mode)
        res, ro = self._alloc_outputs(n, s, ni, dev, return_raw, return_weights, keep_intermediates)
        nb = L.lib().nrf_render_rays_workspace_bytes(self._r, C.c_int64(n), C.byref(rp))
        ws = self._workspace(nb, dev)
        L.check(L.lib().nrf_render_rays(self._r, _ptr(rays), stride, C.c_int64(n), C.byref(rp), _ptr(t), _ptr(u), C.byref(ro), _ptr(ws),
                                        C.c_size_t(ws.numel()), _stream()))
        return res

    def BatchifyRays(self, rays_flat, cone_angle, n_samples, chunk=1024 * 32, **kw):
        """NeRFRenderer.h:465-525: host loop over Chunk-sized slices, torch.cat of every defined field."""
        base = kw.pop("ray_base", 0)
        results = [self.RenderRays(rays_flat[i:i + chunk], cone_angle, n_samples, ray_base=base + i, **kw) for i in range(0, rays_flat.shape[0], chunk)]
        res = NeRFRenderResult()

        def cat(get):
            parts = [get(r) for r in results if get(r) is not None]
            return torch.cat(parts, 0) if parts else None
        o = res.Outputs
        o.RGBMap = cat(lambda r: r.Outputs.RGBMap); o.DispMap = cat(lambda r: r.Outputs.DispMap); o.AccMap = cat(lambda r: r.Outputs.AccMap)
        o.Weights = cat(lambda r: r.Outputs.Weights); o.DepthMap = cat(lambda r: r.Outputs.DepthMap)
        res.Raw = cat(lambda r: r.Raw)
        for k in (results[0].Extras if results else {}):
            res.Extras[k] = torch.cat([r.Extras[k] for r in results], 0)
        return res

    def Render(self, h, w, k, render_params: NeRFRenderParams, rays=(None, None, None), c2w=None, c2w_staticcam=None, row0=0, rows=None, device="cuda"):
        """NeRFRenderer.h:530-605.  Either a pose (c2w, full image or the row tile [row0, row0+rows)) or an explicit ray batch.
        The pose branch is ONE library call (nrf_render_rows: ray generation, view directions, NDC, AABB, packing, the Chunk loop and the
        tile's Near / Far); the ray-batch branch packs and then runs the Chunk loop in one call (nrf_batchify_rays).  In NRF_PREC_F32, and in the matrix-core
        precisions with OverflowPolicy DEFERRED / IGNORE, neither synchronises; with the default policy a matrix-core render ends with one read-back of its
        chunks' non-finite words (flagged chunks are rendered again in NRF_PREC_F32)."""
        self._last_feature_view = None
        p = render_params
        s, ni = int(p.NSamples), int(p.NImportance)
        stride = 11 if p.UseViewdirs else 8
        bb = _host_f32(p.BoundingBox, 6)
        lib = L.lib()
        if c2w is not None:
            rows = h - row0 if rows is None else rows
            dev = torch.device(device)
            sh = (rows, w, 3)
            n = rows * w
            v = L.View()
            v.h, v.w, v.row0, v.rows = int(h), int(w), int(row0), int(rows)
            v.K = (C.c_float * 9)(*_host_f32(k, 9).tolist())
            v.c2w = (C.c_float * 12)(*_host_f32(torch.as_tensor(c2w)[:3, :4] if torch.is_tensor(c2w) else np.asarray(c2w)[:3, :4], 12).tolist())
            if c2w_staticcam is not None:
                v.has_staticcam = 1
                cs = c2w_staticcam
                v.c2w_staticcam = (C.c_float * 12)(*_host_f32(torch.as_tensor(cs)[:3, :4] if torch.is_tensor(cs) else np.asarray(cs)[:3, :4], 12).tolist())
            v.use_viewdirs, v.ndc, v.chunk = int(bool(p.UseViewdirs)), int(bool(p.Ndc)), int(p.Chunk)
            v.bbox = (C.c_float * 6)(*bb.tolist())
            cone_angle = None
            if not p.ThinRay:                                            # GetRays' third result (RayUtils.h:35-43)
                kk = _host_f32(k, 9)
                cone_angle = float(np.float32((np.float32(1.0) / kk[0] + np.float32(1.0) / kk[4]) / np.float32(2.0)) * np.float32(1.1))
            rp = self._params(s, ni, cone_angle, p.LinDisp, p.Perturb, p.WhiteBkgr, p.RawNoiseStd, p.StochasticPreconditioningAlpha, p.BoundingBox, p.Precision,
                              p.Seed, 0, p.CoarseMode, p.OverflowPolicy)
            res, ro = self._alloc_outputs(n, s, ni, dev, p.ReturnRaw, p.ReturnWeights, p.KeepIntermediates)
            rays_ = torch.empty((n, stride), device=dev, dtype=torch.float32)
            nf = torch.empty((2,), device=dev, dtype=torch.float32)
            nb = lib.nrf_render_rows_workspace_bytes(self._r, C.byref(v), C.byref(rp))
            ws = self._workspace(nb, dev)
            L.check(lib.nrf_render_rows(self._r, C.byref(v), C.byref(rp), _ptr(self._linspace(s, dev)), _ptr(self._linspace(ni, dev)) if ni > 0 else None,
                                        C.byref(ro), _ptr(rays_), _ptr(nf), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
        else:
            if c2w_staticcam is not None:
                raise L.NrfError("c2w_staticcam replaces the camera of a POSE render (NeRFRenderer.h:554-558); with an explicit ray batch pass those rays yourself")
            rays_o, rays_d, cone_angle = rays
            rays_o, rays_d = _dev_f32(rays_o), _dev_f32(rays_d)
            sh = tuple(rays_d.shape)
            dev = rays_d.device
            view_src = rays_d.reshape(-1, 3)
            if p.Ndc:
                # cone rays: NDCRays' scale factor divides the new direction's norm by itself (RayUtils.h:73-81: rays_d is already the NDC direction there), exactly 1.0,
                # so cone_angle keeps its value for every ray
                kk = _host_f32(k, 9)
                rays_o, rays_d, _ = NDCRays(h, w, float(kk[0]), 1.0, rays_o, rays_d, None)                       # :567
            o = rays_o.reshape(-1, 3).contiguous(); d = rays_d.reshape(-1, 3).contiguous()
            n = o.shape[0]
            rays_ = torch.empty((n, stride), device=dev, dtype=torch.float32)
            if p.Ndc and p.UseViewdirs:                                  # viewdirs are taken before the warp (:549-561)
                L.check(lib.nrf_pack_rays_viewsrc(_ptr(o), _ptr(d), _ptr(view_src.contiguous()), bb.ctypes.data_as(C.c_void_p), C.c_int64(n), _ptr(rays_), _stream()))
            else:
                L.check(lib.nrf_pack_rays(_ptr(o), _ptr(d), bb.ctypes.data_as(C.c_void_p), C.c_int64(n), int(p.UseViewdirs), _ptr(rays_), _stream()))   # :549-583
            rp = self._params(s, ni, None if p.ThinRay else cone_angle, p.LinDisp, p.Perturb, p.WhiteBkgr, p.RawNoiseStd, p.StochasticPreconditioningAlpha,
                              p.BoundingBox, p.Precision, p.Seed, 0, p.CoarseMode, p.OverflowPolicy)
            res, ro = self._alloc_outputs(n, s, ni, dev, p.ReturnRaw, p.ReturnWeights, p.KeepIntermediates)
            nb = lib.nrf_batchify_rays_workspace_bytes(self._r, C.c_int64(n), int(p.Chunk), C.byref(rp))
            ws = self._workspace(nb, dev)
            L.check(lib.nrf_batchify_rays(self._r, _ptr(rays_), stride, C.c_int64(n), int(p.Chunk), C.byref(rp), _ptr(self._linspace(s, dev)),
                                          _ptr(self._linspace(ni, dev)) if ni > 0 else None, C.byref(ro), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
            self._last_feature_view = self.feature_view()          # (None unless this was a single-chunk render of the feature-reusing fast path)
            nf = None
            if n > 0:
                nf = torch.empty((2,), device=dev, dtype=torch.float32)          # Near / Far stay on the device until someone reads them (a training loop never does):
                L.check(lib.nrf_near_far_range_device(_ptr(rays_), C.c_int64(n), stride, _ptr(nf), _stream()))   # :602-603 without the reference's two host stalls
        out = res.Outputs
        out.RGBMap = out.RGBMap.reshape(sh)                                                               # :591-592
        if len(sh) > 2:
            out.DispMap = out.DispMap.reshape(sh[0], sh[1]); out.DepthMap = out.DepthMap.reshape(sh[0], sh[1])   # :594-600
        res._nf_dev = nf
        res.FeatureView = getattr(self, "_last_feature_view", None)          # (not in Extras: those are tensors)
        res.Extras["rays_flat"] = rays_
        return res


# ------------------------------------------------------------------------------------------------
# Image-space tail of NeRFExecutor::RenderPath (NeRFExecutor.h:690, :698-700; TorchTensorToCVMat, NeRFRenderer.h:58-68)
# ------------------------------------------------------------------------------------------------
def NormalizeDepth(depth_map, near, far):
    """(DepthMap - Near) / (Far - Near), NeRFExecutor.h:690."""
    d = _dev_f32(depth_map)
    out = torch.empty_like(d)
    L.check(L.lib().nrf_normalize_depth(_ptr(d), C.c_int64(d.numel()), C.c_float(near), C.c_float(far), _ptr(out), _stream()))
    return out


def TorchTensorToCVMat(tensor_image):
    """NeRFRenderer.h:58-68 up to the cv::Mat wrap: squeeze, mul(255).clamp(0,255).to(u8); returns the uint8 tensor (on the GPU)."""
    t = _dev_f32(tensor_image).squeeze()
    out = torch.empty(t.shape, device=t.device, dtype=torch.uint8)
    L.check(L.lib().nrf_to_u8(_ptr(t), C.c_int64(t.numel()), _ptr(out), _stream()))
    return out


def RenderViewBuffers(result):
    """The three 8-bit images RenderPath writes for a pose (NeRFExecutor.h:690-700): rgb, disparity, Near/Far-normalised depth."""
    o = result.Outputs
    return TorchTensorToCVMat(o.RGBMap), TorchTensorToCVMat(o.DispMap), TorchTensorToCVMat(NormalizeDepth(o.DepthMap, result.Near, result.Far))


def RenderViewDims(h, w, k, render_factor):
    """The render-factor step of NeRFExecutor::RenderView (NeRFExecutor.h:618-627) -> (h1, w1, K1 [3,3] float32)."""
    K = _host_f32(k, 9)
    K1 = np.empty(9, np.float32)
    h1, w1 = C.c_int(0), C.c_int(0)
    L.check(L.lib().nrf_render_view_dims(int(h), int(w), K.ctypes.data_as(C.c_void_p), C.c_float(float(render_factor)), C.byref(h1), C.byref(w1),
                                         K1.ctypes.data_as(C.c_void_p)))
    return h1.value, w1.value, K1.reshape(3, 3)


def RenderView(renderer, render_pose, w, h, k, rparams: NeRFRenderParams, **kw):
    """NeRFExecutor::RenderView (NeRFExecutor.h:609-650), argument order (pose, w, h, k, params) as there: with RenderFactor != 0 the
    frame is rendered at (h, w) / RenderFactor with fx, fy, cx, cy divided by it.  `renderer` is a NeRFRenderer or a LeRFRenderer mirror;
    row0 / rows (multi-GPU tiles) refer to the downsampled frame."""
    h1, w1, k1 = RenderViewDims(h, w, k, rparams.RenderFactor)
    pose = torch.as_tensor(render_pose)[:3, :4] if torch.is_tensor(render_pose) else np.asarray(render_pose)[:3, :4]
    return renderer.Render(h1, w1, k1, rparams, c2w=pose, **kw)


def RenderPath(renderer, render_poses, h, w, focal, k, rparams: NeRFRenderParams):
    """NeRFExecutor::RenderPath (NeRFExecutor.h:653-737) up to cv::imwrite: per pose the three 8-bit buffers (rgb, disparity, Near/Far-normalised
    depth).  Reproduced quirk: RenderFactor scales h, w and the local `focal` (:657-662) but Render() receives the ORIGINAL k (:668)."""
    h1, w1, _ = RenderViewDims(h, w, k, rparams.RenderFactor)
    return [RenderViewBuffers(renderer.Render(h1, w1, k, rparams, c2w=(torch.as_tensor(p)[:3, :4] if torch.is_tensor(p) else np.asarray(p)[:3, :4])))
            for p in render_poses]


# ------------------------------------------------------------------------------------------------
# LeRFRenderer.h / LeRFRenderer.cpp  (BASELINE config 5: language-embedded radiance field render pass)
# ------------------------------------------------------------------------------------------------
@dataclass
class LeRFRendererOutputs:            # LeRFRenderer.h:9-18
    LangEmbedding: Optional[torch.Tensor] = None            # [N, S, E]
    RenderedLangEmbedding: Optional[torch.Tensor] = None    # [N, E]
    DispMapLE: Optional[torch.Tensor] = None
    AccMapLE: Optional[torch.Tensor] = None
    WeightsLE: Optional[torch.Tensor] = None
    DepthMapLE: Optional[torch.Tensor] = None
    Relevancy: Optional[torch.Tensor] = None                # [N, 2] (LeRFRenderer.cpp:79; set when prompts are given -- parity unpinned, see Relevancy())


class LeRFRenderResult(NeRFRenderResult):               # LeRFRenderer.h:20-26 (Near / Far read lazily from the device, as NeRFRenderResult)
    def __init__(self):
        super().__init__()
        self.Outputs = LeRFRendererOutputs()


def RenderCLIPEmbedding(embeds, weights):
    """LeRFRenderer.h:45-54: embeds [N,S,>=E] (row stride may exceed E), weights [N,S] or [N,S,1] -> normalize(sum_s w*e) [N,E]."""
    e = _dev_f32(embeds); w = _dev_f32(weights).reshape(e.shape[0], e.shape[1])
    n, s, stride = e.shape
    return _clip_embedding(e, stride, stride, w)


def _clip_embedding(e, stride, dim, w):
    n, s = w.shape
    out = torch.empty((n, dim), device=e.device, dtype=torch.float32)
    L.check(L.lib().nrf_render_clip_embedding(_ptr(e), stride, dim, _ptr(w), C.c_int64(n), s, _ptr(out), _stream()))
    return out


def Relevancy(embeds, positives, negatives, positive_id=0):
    """Relevancy(embeds [N, E], positives [P, E], negatives [Q, E]) -> [N, 2] (call sites LeRFRenderer.cpp:79, NeRFExecutor.h:824).  The function's source is external
    (DeliriumV01D/RuCLIP): this is the published LERF relevancy score it mirrors -- PARITY UNPINNED (include/nerfpp_hip.h, nrf_lerf_relevancy)."""
    e = _dev_f32(embeds)

    def phrases(x):           # the prompt embeddings come from the host's text encoder: a host array or a tensor on any device
        x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, np.float32))
        return x.to(device=e.device, dtype=torch.float32).reshape(-1, e.shape[1]).contiguous()
    pos, neg = phrases(positives), phrases(negatives)
    out = torch.empty((e.shape[0], 2), device=e.device, dtype=torch.float32)
    L.check(L.lib().nrf_lerf_relevancy(_ptr(e), C.c_int64(e.shape[0]), int(e.shape[1]), _ptr(pos), int(pos.shape[0]), _ptr(neg), int(neg.shape[0]), int(positive_id),
                                       _ptr(out), _stream()))
    return out


def RelevancyImage(relevancy):
    """The relevancy picture RenderPath writes (NeRFExecutor.h:713-719): rel[..., 0] * 255 -> u8 -> COLORMAP_JET, [..., 3] bytes in B, G, R order (parity unpinned)."""
    r = _dev_f32(relevancy)
    sh = r.shape[:-1]
    n = int(np.prod(sh)) if len(sh) else 1
    out = torch.empty((n, 3), device=r.device, dtype=torch.uint8)
    L.check(L.lib().nrf_relevancy_image(_ptr(r), C.c_int64(n), int(r.shape[-1]), _ptr(out), _stream()))
    return out.reshape(*sh, 3)


class LeRFRenderer:
    """LeRFRenderer (LeRFRenderer.h:57-132, LeRFRenderer.cpp): CuHashEmbedder -> LeRF head -> sigma_le weights -> rendered CLIP
    embedding -> relevancy.  The default configuration (fused matrix-core passes on level-major features) renders a pose or a ray batch with ONE library call
    (nrf_lerf_render_rows / nrf_lerf_batchify_rays); every other configuration is stage-composed over the C ABI."""

    def __init__(self, lang_embed_fn, lerf, lerf_positives=None, lerf_negatives=None, point_chunk=1 << 16, fused=True, precision=L.NRF_PREC_F16_SPLIT):
        """precision: arithmetic of the fused matrix-core passes, L.NRF_PREC_F16_SPLIT (default: hi + lo fp16 operand pairs, fp32-grade like LeRFImpl::forward)
        or L.NRF_PREC_F16_MFMA (plain fp16 operands, ~2x faster network, ~1e-3 relative)."""
        self.LangEmbedFn, self.Lerf = lang_embed_fn, lerf
        self.LerfPositives, self.LerfNegatives = lerf_positives, lerf_negatives
        self.point_chunk = point_chunk          # bounds the [P, E+1] raw tensor (3 KB per point at E = 768)
        # matrix-core path: the LeRF head fused with its render pass (mlp_lerf_mfma.hip); raw_le [N, S, E+1] is never formed
        self.fused = bool(fused) and bool(L.lib().nrf_lerf_mfma_available(lerf._m))
        self.hand_over_geo = True           # split precision, reuse path: the embedding pass takes the sigma net's output from the sigma pass (False: re-evaluates it)
        # split precision: the COARSE pass's sigma_le in exact fp32 on the matrix cores (sigma_lerf_f32.hip) -- its weights choose the fine samples through a
        # discontinuous function, so the fine sample set of the timed mode is then the fp32 stage path's own, bit for bit (False: split arithmetic, for A/B tests)
        self.exact_coarse = True
        self.compose_through_map = True     # reuse path: sigma_le of the sorted depths read through the merge map by the compositing kernel (False: gathered by torch first, A/B)
        self.reuse_features = True          # level-major fused path: encode every sample point once per render (False: the plain two-pass evaluation, for A/B tests)
        self.precision = int(precision)
        # streams of Render's Chunk loop.  ONE by default: the LeRF kernels gain nothing from sharing the CUs (800x800 frame, same call: 140-143 ms on one lane,
        # 146-147 on two; docs/history/profiles/round4/r4g_lerf_lane_chunk_sweep.log); NRF_LERF_LANES overrides
        self.lanes = max(1, min(4, int(os.environ.get("NRF_LERF_LANES", "1"))))
        self._lane_streams = None
        if self.fused:
            self.set_precision(precision)
        # level-major fp16 features straight into the matrix-core kernels' operand fragments (CuHashEmbedder, 16 levels x 8 features); False = fp32 rows
        self.level_major = self.fused and getattr(lang_embed_fn, "mode", None) == L.NRF_HASH_CU and lang_embed_fn.NLevels == 16 and lang_embed_fn.NFeaturesPerLevel == 8
        # the pass as library calls (lerf_render.hip); single_call = False keeps the stage-composed host loop below (A/B tests)
        self.single_call = True
        self.keep_intermediates = True      # Extras: z_coarse / weights_coarse / z_fine of the whole batch (parity tests; a timing loop turns it off)
        self._r = C.c_void_p()
        self._ws = None
        if self.level_major:
            d = L.LerfRendererDesc(lang_embed_fn._h, lerf._m)
            L.check(L.lib().nrf_lerf_renderer_create(C.byref(d), C.byref(self._r)))
            if lerf_positives is not None and lerf_negatives is not None:
                self.SetLeRFPrompts(lerf_positives, lerf_negatives)

    def __del__(self):
        try:
            if getattr(self, "_r", None):
                L.lib().nrf_lerf_renderer_destroy(self._r)
                self._r = None
        except Exception:
            pass

    def GetLeRFPrompts(self):                        # LeRFRenderer.h:85
        return self.LerfPositives, self.LerfNegatives

    def SetLeRFPrompts(self, lerf_positives, lerf_negatives):
        """LeRFRenderer.h:86: [P, E] / [Q, E] phrase embeddings (the text encoder is the host's)."""
        self.LerfPositives, self.LerfNegatives = lerf_positives, lerf_negatives
        if self._r:
            if lerf_positives is None or lerf_negatives is None:
                L.check(L.lib().nrf_lerf_set_prompts(self._r, None, 0, None, 0, 0, _stream()))
            else:
                pos = _host_f32(lerf_positives, None); neg = _host_f32(lerf_negatives, None)
                E = self.Lerf.GetLangEmbedDim()
                L.check(L.lib().nrf_lerf_set_prompts(self._r, pos.ctypes.data_as(C.c_void_p), pos.size // E, neg.ctypes.data_as(C.c_void_p), neg.size // E, 0, _stream()))

    def _single_call_ok(self, p):
        s, ni = int(p.NSamples), int(p.NImportance)
        return bool(self.single_call and self._r and self.fused and self.level_major and self.reuse_features and self.hand_over_geo and self.compose_through_map
                    and not p.ReturnRaw and ni > 0 and s % 32 == 0 and (s + ni) % 32 == 0 and p.ThinRay and p.Perturb == 0 and p.RawNoiseStd == 0
                    and not p.Ndc and p.StochasticPreconditioningAlpha == 0)            # as HipLeRFRenderer::Render (include/nerfpp_torch.h): both take the inherited path

    def _render_single_call(self, h, w, k, p, rays, c2w, row0, rows):
        """LeRFRenderer::Render as one C call: nrf_lerf_render_rows for a pose, nrf_lerf_batchify_rays for a ray batch."""
        lib = L.lib()
        s, ni = int(p.NSamples), int(p.NImportance)
        sf, E = s + ni, self.Lerf.GetLangEmbedDim()
        stride = 11 if p.UseViewdirs else 8
        bb = _host_f32(p.BoundingBox, 6)
        rp = L.RenderParams(s, ni, int(p.LinDisp), 0, self.precision, ATEN_SUM_VEC)
        rp.coarse_mode = L.NRF_COARSE_AUTO if self.exact_coarse else L.NRF_COARSE_FULL
        dev = torch.device("cuda", torch.cuda.current_device())
        o = d = None
        if c2w is not None:
            rows = h - row0 if rows is None else rows
            n = rows * w
        else:
            o = _dev_f32(rays[0]).reshape(-1, 3).contiguous(); d = _dev_f32(rays[1]).reshape(-1, 3).contiguous()
            n, dev = o.shape[0], o.device
        res = LeRFRenderResult()
        want = bool(p.ReturnWeights)          # LeRFRenderer.cpp:180-185: without ReturnWeights the weights and the rendered embedding are dropped
        f32 = dict(device=dev, dtype=torch.float32)
        out = res.Outputs
        out.DispMapLE = torch.empty((n,), **f32); out.AccMapLE = torch.empty((n,), **f32); out.DepthMapLE = torch.empty((n,), **f32)
        ro = L.LerfOutputs()
        ro.d_disp, ro.d_acc, ro.d_depth = _ptr(out.DispMapLE), _ptr(out.AccMapLE), _ptr(out.DepthMapLE)
        if want:
            out.WeightsLE = torch.empty((n, sf), **f32); out.RenderedLangEmbedding = torch.empty((n, E), **f32)
            ro.d_weights, ro.d_embedding = _ptr(out.WeightsLE), _ptr(out.RenderedLangEmbedding)
        if self.LerfPositives is not None and self.LerfNegatives is not None:
            out.Relevancy = torch.empty((n, 2), **f32)
            ro.d_relevancy = _ptr(out.Relevancy)
        if p.KeepIntermediates or self.keep_intermediates:
            res.Extras["z_fine"] = torch.empty((n, sf), **f32); res.Extras["z_coarse"] = torch.empty((n, s), **f32); res.Extras["weights_coarse"] = torch.empty((n, s), **f32)
            ro.d_z_fine, ro.d_z_coarse, ro.d_weights_coarse = _ptr(res.Extras["z_fine"]), _ptr(res.Extras["z_coarse"]), _ptr(res.Extras["weights_coarse"])
        rays_ = torch.empty((n, stride), **f32)
        t, u = NeRFRenderer._linspace(s, dev), NeRFRenderer._linspace(ni, dev)

        def workspace(nb):
            if self._ws is None or self._ws.numel() < nb or self._ws.device != dev:
                self._ws = None
                self._ws = torch.empty((int(nb),), device=dev, dtype=torch.uint8)
            return self._ws
        L.check(lib.nrf_lerf_renderer_set_lanes(self._r, int(self.lanes)))          # this renderer's own lane count (default 1: see __init__)
        self._issue_single_call(lib, h, w, k, p, c2w, row0, rows, bb, rp, ro, res, rays_, t, u, workspace, f32, stride, n, o, d)
        res.Extras["rays_flat"] = rays_
        res.FeatureView = self.feature_view() if c2w is None else None          # (ray-batch branch; None unless this call rendered exactly one chunk)
        return res

    def feature_view(self):
        """nrf_lerf_renderer_last_features: where the most recent one-chunk render left the language features of its fine depths in this renderer's workspace --
        dict(feats, cols, keep, src: device addresses; n, sf; serial) or None.  Valid until the next render call on this renderer (LeRFTrainer.backward reads it)."""
        if not getattr(self, "_r", None):
            return None
        fp, kp, sp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cols, nn, sf, ser = C.c_int64(), C.c_int64(), C.c_int(), C.c_uint64()
        rc = L.lib().nrf_lerf_renderer_last_features(self._r, C.byref(fp), C.byref(cols), C.byref(kp), C.byref(sp), C.byref(nn), C.byref(sf), C.byref(ser))
        if rc != 0:
            return None
        return dict(feats=fp.value, cols=int(cols.value), keep=kp.value, src=sp.value, n=int(nn.value), sf=int(sf.value), serial=int(ser.value))

    def _issue_single_call(self, lib, h, w, k, p, c2w, row0, rows, bb, rp, ro, res, rays_, t, u, workspace, f32, stride, n, o, d):
        if c2w is not None:
            v = L.View()
            v.h, v.w, v.row0, v.rows = int(h), int(w), int(row0), int(rows)
            v.K = (C.c_float * 9)(*_host_f32(k, 9).tolist())
            v.c2w = (C.c_float * 12)(*_host_f32(torch.as_tensor(c2w)[:3, :4] if torch.is_tensor(c2w) else np.asarray(c2w)[:3, :4], 12).tolist())
            v.use_viewdirs, v.ndc, v.chunk = int(bool(p.UseViewdirs)), 0, int(p.Chunk)
            v.bbox = (C.c_float * 6)(*bb.tolist())
            nf = torch.empty((2,), **f32)
            ws = workspace(lib.nrf_lerf_render_rows_workspace_bytes(self._r, C.byref(v), C.byref(rp)))
            L.check(lib.nrf_lerf_render_rows(self._r, C.byref(v), C.byref(rp), _ptr(t), _ptr(u), C.byref(ro), _ptr(rays_), _ptr(nf), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
            res._nf_dev = nf
        else:
            L.check(lib.nrf_pack_rays(_ptr(o), _ptr(d), bb.ctypes.data_as(C.c_void_p), C.c_int64(n), int(p.UseViewdirs), _ptr(rays_), _stream()))
            ws = workspace(lib.nrf_lerf_batchify_rays_workspace_bytes(self._r, C.c_int64(n), int(p.Chunk), C.byref(rp)))
            L.check(lib.nrf_lerf_batchify_rays(self._r, _ptr(rays_), stride, C.c_int64(n), int(p.Chunk), C.byref(rp), _ptr(t), _ptr(u), C.byref(ro), _ptr(ws),
                                               C.c_size_t(ws.numel()), _stream()))
            nfd = torch.empty((2,), device=rays_.device, dtype=torch.float32)
            L.check(lib.nrf_near_far_range_device(_ptr(rays_), C.c_int64(n), stride, _ptr(nfd), _stream()))
            res._nf_dev = nfd
        res.Extras["rays_flat"] = rays_
        return res

    def set_precision(self, precision):
        L.check(L.lib().nrf_lerf_set_precision(self.Lerf._m, int(precision)))
        self.precision = int(precision)

    @property
    def precision_name(self):
        return {L.NRF_PREC_F16_MFMA: "f16", L.NRF_PREC_F16_SPLIT: "f16x3"}.get(self.precision, "f32")

    def _exact_coarse_on(self):
        return bool(self.exact_coarse) and self.precision == L.NRF_PREC_F16_SPLIT and bool(L.lib().nrf_lerf_sigma_exact_available(self.Lerf._m))

    def _sigma_fused(self, pts, exact=False):
        """sigma_le [N,S] (keep-masked) and the hash features [N*S, in] of the sample points, sigma net on the matrix cores (exact: in exact fp32, the coarse pass)."""
        n, s = pts.shape[0], pts.shape[1]
        sig = torch.empty((n, s), device=pts.device, dtype=torch.float32)
        if self.level_major:
            flat = pts.reshape(-1, 3).contiguous()
            x = torch.empty((16, n * s, 8), device=pts.device, dtype=torch.float16)
            ku8 = torch.empty((n * s,), device=pts.device, dtype=torch.uint8)
            L.check(L.lib().nrf_hash_encode_lm_f16(self.LangEmbedFn._h, _ptr(flat), C.c_int64(n * s), _ptr(x), _ptr(ku8), _stream()))
            if exact:
                L.check(L.lib().nrf_lerf_sigma_exact_lm_strided(self.Lerf._m, _ptr(x), C.c_int64(n * s), _ptr(ku8), C.c_int64(n * s), _ptr(sig), None, C.c_int64(0), _stream()))
            else:
                L.check(L.lib().nrf_lerf_sigma_lm(self.Lerf._m, _ptr(x), _ptr(ku8), C.c_int64(n * s), _ptr(sig), _stream()))
            return sig, x
        x, keep = self.LangEmbedFn.forward(pts.reshape(-1, 3))
        ku8 = keep.to(torch.uint8)
        L.check(L.lib().nrf_lerf_sigma(self.Lerf._m, _ptr(x), _ptr(ku8), C.c_int64(n * s), _ptr(sig), _stream()))
        return sig, x

    def _render_fine_reusing(self, rays, stride, z, pts, rays_d, ni):
        """Both passes of a hierarchical render with every sample point encoded ONCE (level-major fused path): the fine pass's S + N_importance depths contain the
        S coarse ones, whose features and sigma_le exist already -- the hash encode and the sigma net run on the N_importance new samples only, the embedding pass
        reads every depth's feature column through the merge map (nrf_fine_depths_merge).  Same kernels on the same inputs: results equal the plain two-pass
        evaluation bit for bit.  Returns (coarse outputs, fine outputs, z_fine)."""
        n, s = z.shape
        sf, dev = s + ni, z.device
        cols = n * sf
        lib, h, m = L.lib(), self.LangEmbedFn._h, self.Lerf._m
        x = torch.empty((16, cols, 8), device=dev, dtype=torch.float16)
        keep = torch.empty((cols,), device=dev, dtype=torch.uint8)
        sig = torch.empty((cols,), device=dev, dtype=torch.float32)           # [coarse n*s | new n*ni], the table's column order
        L.check(lib.nrf_hash_encode_lm_f16_strided(h, _ptr(pts), C.c_int64(n * s), _ptr(x), C.c_int64(cols), _ptr(keep), _stream()))
        # split precision: the sigma pass also leaves (sigma, geo32) per column, and the embedding pass starts at LE0 from it (nrf_lerf_*_geo)
        geo = torch.empty((int(lib.nrf_lerf_geo_bytes(C.c_int64(cols))),), device=dev, dtype=torch.uint8) if self.precision == L.NRF_PREC_F16_SPLIT and self.hand_over_geo else None
        def sigma_pass(x_ptr, keep_t, count, sig_t, col0):
            if geo is None:
                L.check(lib.nrf_lerf_sigma_lm_strided(m, x_ptr, C.c_int64(cols), _ptr(keep_t), C.c_int64(count), _ptr(sig_t), _stream()))
            else:
                L.check(lib.nrf_lerf_sigma_geo_lm_strided(m, x_ptr, C.c_int64(cols), _ptr(keep_t), C.c_int64(count), _ptr(sig_t), C.c_void_p(geo.data_ptr() + col0 * 32),
                                                          C.c_int64(cols), _stream()))
        if self._exact_coarse_on():
            # coarse columns: sigma_le in exact fp32 (== the fp32 stage path bit for bit), and -- with the hand-over -- the sigma net's geo output split from the exact values
            L.check(lib.nrf_lerf_sigma_exact_lm_strided(m, _ptr(x), C.c_int64(cols), _ptr(keep), C.c_int64(n * s), _ptr(sig), _ptr(geo), C.c_int64(cols), _stream()))
        else:
            sigma_pass(_ptr(x), keep, n * s, sig, 0)
        out1 = self._weights_from_sigma(sig[:n * s].view(n, s), z, rays_d)
        u = NeRFRenderer._linspace(ni, dev)                                   # cached: a fresh host tensor's .to(device) is a synchronous copy per chunk
        zf = torch.empty((n, sf), device=dev); src = torch.empty((n, sf), device=dev, dtype=torch.int32); z_new = torch.empty((n, ni), device=dev)
        L.check(lib.nrf_fine_depths_merge(_ptr(z), _ptr(out1.WeightsLE), C.c_int64(n), s, _ptr(u), ni, ATEN_SUM_VEC, _ptr(zf), _ptr(src), _ptr(z_new), _stream()))
        pts_new = torch.empty((n, ni, 3), device=dev)
        L.check(lib.nrf_points(_ptr(rays), stride, _ptr(z_new), C.c_int64(n), ni, _ptr(pts_new), _stream()))
        x_new = C.c_void_p(x.data_ptr() + n * s * 8 * 2)                      # column n*s of level 0
        L.check(lib.nrf_hash_encode_lm_f16_strided(h, _ptr(pts_new), C.c_int64(n * ni), x_new, C.c_int64(cols), _ptr(keep[n * s:]), _stream()))
        sigma_pass(x_new, keep[n * s:], n * ni, sig[n * s:], n * s)
        if self.compose_through_map:
            o = self._weights_from_sigma(sig, zf, rays_d, src)                 # sigma_le stays in column order; the compositing kernel reads through the merge map
        else:
            o = self._weights_from_sigma(sig[src.reshape(-1).long()].view(n, sf), zf, rays_d)
        E = self.Lerf.GetLangEmbedDim()
        acc = torch.empty((n, E), device=dev, dtype=torch.float32)
        if geo is None:
            L.check(lib.nrf_lerf_render_embedding_lm_gather(m, _ptr(x), C.c_int64(cols), _ptr(src), _ptr(o.WeightsLE), C.c_int64(n), sf, _ptr(acc), _stream()))
        else:
            L.check(lib.nrf_lerf_render_embedding_lm_geo(m, _ptr(x), C.c_int64(cols), _ptr(src), _ptr(geo), C.c_int64(cols), _ptr(o.WeightsLE), C.c_int64(n), sf, _ptr(acc), _stream()))
        ones = torch.ones((n, 1), device=dev, dtype=torch.float32)
        o.RenderedLangEmbedding = _clip_embedding(acc, E, E, ones)
        return out1, o, zf

    def _weights_from_sigma(self, sig, z, rays_d, src=None):
        """sigma_le -> weights / depth / disp / acc (nrf_raw2weights); src: sample (ray, j)'s sigma_le is sig[src[ray, j]] (the merge map), else sig[ray, j]."""
        n, s = z.shape
        o = LeRFRendererOutputs(WeightsLE=torch.empty((n, s), device=sig.device), DepthMapLE=torch.empty((n,), device=sig.device),
                                DispMapLE=torch.empty((n,), device=sig.device), AccMapLE=torch.empty((n,), device=sig.device))
        if src is None:
            L.check(L.lib().nrf_raw2weights(_ptr(sig), 1, 0, _ptr(z), _ptr(rays_d), 3, C.c_int64(n), s, _ptr(o.WeightsLE), _ptr(o.DepthMapLE), _ptr(o.DispMapLE),
                                            _ptr(o.AccMapLE), _stream()))
        else:
            L.check(L.lib().nrf_raw2weights_gather(_ptr(sig), 1, 0, _ptr(src), _ptr(z), _ptr(rays_d), 3, C.c_int64(n), s, _ptr(o.WeightsLE), _ptr(o.DepthMapLE),
                                                   _ptr(o.DispMapLE), _ptr(o.AccMapLE), _stream()))
        return o

    def _render_fused(self, pts, z, rays_d, want_embedding, exact=False):
        sig, x = self._sigma_fused(pts, exact)
        o = self._weights_from_sigma(sig, z, rays_d)
        if want_embedding:
            n, s = sig.shape
            E = self.Lerf.GetLangEmbedDim()
            acc = torch.empty((n, E), device=pts.device, dtype=torch.float32)
            fn = L.lib().nrf_lerf_render_embedding_lm if self.level_major else L.lib().nrf_lerf_render_embedding
            L.check(fn(self.Lerf._m, _ptr(x), _ptr(o.WeightsLE), C.c_int64(n), s, _ptr(acc), _stream()))
            ones = torch.ones((n, 1), device=pts.device, dtype=torch.float32)
            o.RenderedLangEmbedding = _clip_embedding(acc, E, E, ones)          # the final normalize of RenderCLIPEmbedding (LeRFRenderer.h:53)
        return o

    def RunLENetwork(self, inputs):
        """LeRFRenderer.cpp:5-25: [N,S,3] -> [N,S,E+1], sigma_le zeroed where the embedder's keep_mask is false."""
        pts = _dev_f32(inputs)
        flat = pts.reshape(-1, 3)
        outs = []
        for i in range(0, flat.shape[0], self.point_chunk):
            emb, keep = self.LangEmbedFn.forward(flat[i:i + self.point_chunk])
            o = self.Lerf.forward(emb, L.NRF_PREC_F32)
            o[~keep, -1] = 0
            outs.append(o)
        out = torch.cat(outs, 0) if outs else torch.empty((0, self.Lerf.GetOutputDims()), device=pts.device)
        return out.reshape(pts.shape[0], pts.shape[1], -1)

    def RawToLEOutputs(self, raw_le, z_vals_le, rays_d, lang_embed_dim=768, raw_noise_std=0.0):
        """LeRFRenderer.cpp:27-82 without Relevancy."""
        if raw_noise_std > 0:
            raise L.NrfError("raw_noise_std > 0 is the training-time noise branch; not built")
        raw = _dev_f32(raw_le); z = _dev_f32(z_vals_le); d = _dev_f32(rays_d)
        n, s, c = raw.shape
        o = LeRFRendererOutputs(LangEmbedding=raw[..., :lang_embed_dim], WeightsLE=torch.empty((n, s), device=raw.device),
                                DepthMapLE=torch.empty((n,), device=raw.device), DispMapLE=torch.empty((n,), device=raw.device),
                                AccMapLE=torch.empty((n,), device=raw.device))
        L.check(L.lib().nrf_raw2weights(_ptr(raw), c, lang_embed_dim, _ptr(z), _ptr(d), 3, C.c_int64(n), s, _ptr(o.WeightsLE), _ptr(o.DepthMapLE),
                                        _ptr(o.DispMapLE), _ptr(o.AccMapLE), _stream()))
        o.RenderedLangEmbedding = _clip_embedding(raw, c, lang_embed_dim, o.WeightsLE)
        return o

    def RenderRays(self, ray_batch, cone_angle, n_samples, return_raw=False, lin_disp=False, perturb=0.0, n_importance=0, white_bkgr=False,
                   raw_noise_std=0.0, stochastic_preconditioning_alpha=0.0, bounding_box=None, return_weights=True):
        """LeRFRenderer.cpp:85-187 (deterministic path)."""
        if perturb > 0 or raw_noise_std > 0 or stochastic_preconditioning_alpha > 0 or (cone_angle is not None and torch.is_tensor(cone_angle) and cone_angle.numel()):
            raise L.NrfError("perturb / noise / preconditioning / TangentScatter are RNG branches; render with ThinRay=True, Perturb=0")
        rays = _dev_f32(ray_batch)
        n, stride = rays.shape
        dev = rays.device
        s, ni = int(n_samples), int(n_importance)
        E = self.Lerf.GetLangEmbedDim()
        t = NeRFRenderer._linspace(s, dev)
        z = torch.empty((n, s), device=dev); pts = torch.empty((n, s, 3), device=dev)
        L.check(L.lib().nrf_z_vals(_ptr(rays), stride, C.c_int64(n), _ptr(t), s, int(lin_disp), _ptr(z), _stream()))
        L.check(L.lib().nrf_points(_ptr(rays), stride, _ptr(z), C.c_int64(n), s, _ptr(pts), _stream()))
        rays_d = rays[:, 3:6].contiguous()
        res = LeRFRenderResult()
        if self.fused and self.level_major and self.reuse_features and not return_raw and ni > 0 and s % 32 == 0 and (s + ni) % 32 == 0 and n * (s + ni) < (1 << 31):
            out1, res.Outputs, zf = self._render_fine_reusing(rays, stride, z, pts, rays_d, ni)
            res.Extras["z_fine"] = zf; res.Extras["z_coarse"] = z; res.Extras["weights_coarse"] = out1.WeightsLE
            if not return_weights:
                res.Outputs.WeightsLE = None; res.Outputs.RenderedLangEmbedding = None      # LeRFRenderer.cpp:180-185
            return res
        if self.fused and not return_raw and s % 32 == 0 and (ni == 0 or (s + ni) % 32 == 0):
            # coarse pass: only sigma_le is consumed (the reference also renders a coarse embedding, LeRFRenderer.cpp:139, and drops it)
            out1 = self._render_fused(pts, z, rays_d, want_embedding=(ni == 0), exact=(ni > 0 and self.level_major and self._exact_coarse_on()))
            res.Outputs = out1
            if ni > 0:
                u = NeRFRenderer._linspace(ni, dev)
                zf = torch.empty((n, s + ni), device=dev)
                L.check(L.lib().nrf_fine_depths(_ptr(z), _ptr(out1.WeightsLE), C.c_int64(n), s, _ptr(u), ni, ATEN_SUM_VEC, _ptr(zf), _stream()))
                ptsf = torch.empty((n, s + ni, 3), device=dev)
                L.check(L.lib().nrf_points(_ptr(rays), stride, _ptr(zf), C.c_int64(n), s + ni, _ptr(ptsf), _stream()))
                res.Outputs = self._render_fused(ptsf, zf, rays_d, want_embedding=True)
                res.Extras["z_fine"] = zf
            res.Extras["z_coarse"] = z
            if not return_weights:
                res.Outputs.WeightsLE = None; res.Outputs.RenderedLangEmbedding = None      # LeRFRenderer.cpp:180-185
            return res
        raw = self.RunLENetwork(pts)
        out1 = self.RawToLEOutputs(raw, z, rays_d, E)
        res.Outputs = out1 if ni == 0 else None        # (the reference leaves Outputs undefined when n_importance == 0; the coarse ones are returned here)
        if ni > 0:
            u = torch.linspace(0.0, 1.0, ni, dtype=torch.float32).to(dev)
            zf = torch.empty((n, s + ni), device=dev)
            L.check(L.lib().nrf_fine_depths(_ptr(z), _ptr(out1.WeightsLE), C.c_int64(n), s, _ptr(u), ni, ATEN_SUM_VEC, _ptr(zf), _stream()))
            ptsf = torch.empty((n, s + ni, 3), device=dev)
            L.check(L.lib().nrf_points(_ptr(rays), stride, _ptr(zf), C.c_int64(n), s + ni, _ptr(ptsf), _stream()))
            raw = self.RunLENetwork(ptsf)
            res.Outputs = self.RawToLEOutputs(raw, zf, rays_d, E)
            res.Extras["z_fine"] = zf
        res.Extras["z_coarse"] = z
        if return_raw:
            res.Raw = raw
        if not return_weights:
            res.Outputs.WeightsLE = None; res.Outputs.LangEmbedding = None; res.Outputs.RenderedLangEmbedding = None   # LeRFRenderer.cpp:180-185
        return res

    def Render(self, h, w, k, render_params: NeRFRenderParams, rays=(None, None, None), c2w=None, row0=0, rows=None):
        """LeRFRenderer.cpp:265-330."""
        p = render_params
        if self._single_call_ok(p) and (c2w is not None or (rays[0] is not None and torch.as_tensor(rays[0]).numel() > 0)):
            return self._render_single_call(h, w, k, p, rays, c2w, row0, rows)
        if c2w is not None:
            rays_o, rays_d, cone_angle = GetRays(h, w, k, c2w, row0=row0, rows=rows)
        else:
            rays_o, rays_d, cone_angle = rays
        bb = _host_f32(p.BoundingBox, 6)
        o = _dev_f32(rays_o).reshape(-1, 3).contiguous(); d = _dev_f32(rays_d).reshape(-1, 3).contiguous()
        n = o.shape[0]
        stride = 11 if p.UseViewdirs else 8
        rays_ = torch.empty((n, stride), device=o.device, dtype=torch.float32)
        L.check(L.lib().nrf_pack_rays(_ptr(o), _ptr(d), bb.ctypes.data_as(C.c_void_p), C.c_int64(n), int(p.UseViewdirs), _ptr(rays_), _stream()))
        def one(i):
            return self.RenderRays(rays_[i:i + p.Chunk], None if p.ThinRay else cone_angle, p.NSamples, return_raw=p.ReturnRaw, lin_disp=p.LinDisp,
                                   perturb=p.Perturb, n_importance=p.NImportance, white_bkgr=p.WhiteBkgr, raw_noise_std=p.RawNoiseStd,
                                   return_weights=p.ReturnWeights)
        starts = list(range(0, n, p.Chunk))
        if self.lanes >= 2 and len(starts) >= 2 and rays_.is_cuda:
            # the Chunk loop on two lanes (as nrf_batchify_rays does for the NeRF renderers): consecutive chunks on two streams forked from and joined to the current one, so that
            # one chunk's gather-bound F = 8 hash encode shares the CUs with another's matrix-bound passes.  Same kernels on the same slices: same results.
            cur = torch.cuda.current_stream()
            if self._lane_streams is None:
                self._lane_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            for st in self._lane_streams:
                st.wait_stream(cur)
            parts = []
            for j, i in enumerate(starts):
                with torch.cuda.stream(self._lane_streams[j & 1]):
                    parts.append(one(i))
            for st in self._lane_streams:
                cur.wait_stream(st)
            for q in parts:            # the results are read (concatenated) on the current stream: tell the allocator before the lanes may reuse their memory
                for t in list(vars(q.Outputs).values()) + list(q.Extras.values()) + [q.Raw]:
                    if torch.is_tensor(t):
                        t.record_stream(cur)
        else:
            parts = [one(i) for i in starts]
        res = LeRFRenderResult()
        for name in ("RenderedLangEmbedding", "DispMapLE", "AccMapLE", "WeightsLE", "DepthMapLE"):
            vals = [getattr(q.Outputs, name) for q in parts if getattr(q.Outputs, name) is not None]
            setattr(res.Outputs, name, torch.cat(vals, 0) if vals else None)
        for k_ in (parts[0].Extras if parts else {}):
            res.Extras[k_] = torch.cat([q.Extras[k_] for q in parts], 0)
        nfd = torch.empty((2,), device=rays_.device, dtype=torch.float32)
        L.check(L.lib().nrf_near_far_range_device(_ptr(rays_), C.c_int64(n), stride, _ptr(nfd), _stream()))
        res._nf_dev = nfd
        if self.LerfPositives is not None and self.LerfNegatives is not None and res.Outputs.RenderedLangEmbedding is not None:
            res.Outputs.Relevancy = Relevancy(res.Outputs.RenderedLangEmbedding, self.LerfPositives, self.LerfNegatives)      # LeRFRenderer.cpp:79
        res.Extras["rays_flat"] = rays_
        return res
