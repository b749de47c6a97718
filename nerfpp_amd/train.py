"""One optimisation step of NeRFExecutor::Train (NeRFExecutor.h:862-995) over the C ABI -- SURVEY section 8f, row N1.

    render the ray batch (NeRFRenderer::Render with an explicit ray batch, :876) -> huber_loss(RGBMap, target) (:883)
    -> loss.backward() (:923) -> Adam(lr, betas (0.9, 0.99), eps 1e-15).step() (:539, :985)

Gradients flow through the FINE pass only (z_samples are detached, NeRFRenderer.h:429; rays and depths carry no parameters).
PyTorch owns the buffers (fp32 master parameters, Adam moments) and nothing else: every arithmetic step is a call into
libnerfpp_hip.so on the current HIP stream.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .modules import _HashBase, CuHashEmbedder, NeRFSmall, _ptr, _stream, _dev_f32
from .renderer import NeRFRenderer, NeRFRenderParams, RngFill, StochasticPrecondition, TangentScatter


class Trainer:
    def __init__(self, embedder: _HashBase, embeddirs, mlp: NeRFSmall, table, mlp_blob, learning_rate=5e-4, betas=(0.9, 0.99), eps=1e-15,
                 tv_loss_weight=0.0, seed=0, mlp_backward="f32", hash_backward="f32", grad_sync=None):
        if not isinstance(embedder, _HashBase) or not isinstance(mlp, NeRFSmall):
            raise L.NrfError("Trainer is built for hash-grid + NeRFSmall scenes (the reference's HashNeRF training configuration)")
        self.embedder, self.embeddirs, self.mlp = embedder, embeddirs, mlp
        self.renderer = NeRFRenderer(embedder, embeddirs, mlp)
        dev = "cuda"
        self.table = torch.as_tensor(np.ascontiguousarray(table, np.float32).reshape(-1) if not torch.is_tensor(table) else table.reshape(-1)).to(dev).contiguous()
        self.blob = torch.as_tensor(np.ascontiguousarray(mlp_blob, np.float32).reshape(-1) if not torch.is_tensor(mlp_blob) else mlp_blob.reshape(-1)).to(dev).contiguous()
        assert self.table.numel() == embedder.table_elems() and self.blob.numel() == mlp.n_params
        self.m_table, self.v_table = torch.zeros_like(self.table), torch.zeros_like(self.table)
        self.m_blob, self.v_blob = torch.zeros_like(self.blob), torch.zeros_like(self.blob)
        self.g_table, self.g_blob = torch.zeros_like(self.table), torch.zeros_like(self.blob)
        self.lr, self.betas, self.eps, self.t = float(learning_rate), betas, float(eps), 0
        # TotalVariationLoss of the LibTorch HashEmbedder, weight 1e-6 in the reference for the first half of training (NeRFExecutor.h:896-913)
        self.tv_loss_weight, self.seed = float(tv_loss_weight), int(seed)
        self.tv_loss = torch.zeros((1,), device=dev)
        # "f32": layer-wise fp32 kernels (the parity path, pinned to the reference's autograd); "f16": one fused matrix-core kernel
        # (fp16 operands, fp32 accumulation, device-side loss scaling -- mlp_small_bwd_mfma.hip)
        if mlp_backward not in ("f32", "f16"):
            raise L.NrfError("mlp_backward must be 'f32' or 'f16'")
        self.mlp_backward = mlp_backward
        # "f32": one float atomic per feature; "packed": both features of an entry in one 64-bit fixed-point atomic (nrf_hash_backward_rays_packed)
        if hash_backward not in ("f32", "packed"):
            raise L.NrfError("hash_backward must be 'f32' or 'packed'")
        if hash_backward == "packed" and embedder.NFeaturesPerLevel != 2:
            raise L.NrfError("hash_backward='packed' needs 2 features per level")
        self.hash_backward = hash_backward
        self.grad_sync = grad_sync          # callable(g_table, g_blob) reducing the gradients across data-parallel ranks in place, or None
        self._hws = None
        self._ws = None
        if isinstance(embedder, CuHashEmbedder):
            embedder.set_dense_budget(0)        # the baked dense pyramid of the render fast path would be re-baked after every step
        self._push_params()

    def _push_params(self):
        self.embedder.set_table(self.table)
        L.check(L.lib().nrf_mlp_set_params(self.mlp._m, _ptr(self.blob), 1, _stream()))

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty((int(nbytes),), device="cuda", dtype=torch.uint8)
        return self._ws

    def backward(self, res, target, n_samples_out, white_bkgr, params=None, cone_angle=None):
        """loss + gradients of one rendered batch (fills self.g_table / self.g_blob); returns device tensor [huber, mse].
        `params` (the NeRFRenderParams of the forward) tells which stochastic branches ran: their counter-based draws are regenerated here
        (same seed, stream and element index, include/nrf_rng.h), so the backward sees exactly the forward's sample points and densities."""
        lib = L.lib()
        rays = res.Extras["rays_flat"]
        n, stride = rays.shape
        rgb = res.Outputs.RGBMap.reshape(n, 3).contiguous()
        tgt = _dev_f32(target).reshape(n, 3)
        loss_mse = torch.empty((2,), device=rgb.device); g_rgb = torch.empty_like(rgb)
        L.check(lib.nrf_huber_loss(_ptr(rgb), _ptr(tgt), C.c_int64(rgb.numel()), _ptr(loss_mse), _ptr(g_rgb), _stream()))
        raw = res.Raw; s = n_samples_out
        z = res.Extras["z_fine"] if "z_fine" in res.Extras else res.Extras["z_coarse"]
        assert raw.shape == (n, s, 4) and z.shape == (n, s)
        g_raw = torch.empty_like(raw)
        p = params
        seed = int(p.Seed) if p is not None else 0
        fine = "z_fine" in res.Extras
        noise_std = float(p.RawNoiseStd) if p is not None else 0.0
        noise = RngFill(seed, L.NRF_RNG_NOISE_FINE if fine else L.NRF_RNG_NOISE_COARSE, 0, n * s, normal=True, device=rays.device) if noise_std > 0 else None
        L.check(lib.nrf_raw2outputs_backward_noise(_ptr(raw), _ptr(z), C.c_void_p(rays.data_ptr() + 12), stride, C.c_int64(n), s, 4, int(white_bkgr), _ptr(noise),
                                                   C.c_float(noise_std), _ptr(g_rgb), _ptr(g_raw), _stream()))
        pts = torch.empty((n * s, 3), device=rays.device)
        L.check(lib.nrf_points(_ptr(rays), stride, _ptr(z), C.c_int64(n), s, _ptr(pts), _stream()))
        if p is not None and fine and p.StochasticPreconditioningAlpha > 0:          # NeRFRenderer.h:433-443 (fine pass only)
            pn = RngFill(seed, L.NRF_RNG_PRECOND, 0, n * s * 3, normal=True, device=rays.device)
            pts = StochasticPrecondition(pts, pn, float(p.StochasticPreconditioningAlpha), p.BoundingBox)
        if cone_angle is not None:                                                  # TangentScatter, NeRFRenderer.h:307-362
            ur = RngFill(seed, L.NRF_RNG_R_FINE if fine else L.NRF_RNG_R_COARSE, 0, n * s, device=rays.device)
            ut = RngFill(seed, L.NRF_RNG_THETA_FINE if fine else L.NRF_RNG_THETA_COARSE, 0, n * s, device=rays.device)
            pts = TangentScatter(pts.reshape(n, s, 3), z, float(cone_angle), rays[:, 3:6].contiguous(), p.BoundingBox if p is not None else None, ur, ut).reshape(n * s, 3)
        dirs, _ = self.embeddirs.forward(rays[:, 8:11].contiguous())
        in_ch = self.embedder.GetOutputDims()
        g_x = torch.empty((n * s, in_ch), device=rays.device)
        self.g_blob.zero_(); self.g_table.zero_()
        lm = (self.mlp_backward == "f16" and isinstance(self.embedder, CuHashEmbedder) and self.embedder.NLevels == 16 and self.embedder.NFeaturesPerLevel == 2
              and dirs.shape[1] == 16)
        if lm:
            # the fast path's own layout: level-major fp16 hash features + one fp16 direction row per ray; no [p, 48] fp32 input is formed
            feats = torch.empty((16, n * s, 2), device=rays.device, dtype=torch.float16)
            keep_u8 = torch.empty((n * s,), device=rays.device, dtype=torch.uint8)
            ptsc = pts.contiguous()
            L.check(lib.nrf_hash_encode_lm_f16(self.embedder._h, _ptr(ptsc), C.c_int64(n * s), _ptr(feats), _ptr(keep_u8), _stream()))
            dirs16 = dirs.to(torch.float16).contiguous()
            L.check(lib.nrf_mask_sigma_grad(_ptr(keep_u8), C.c_int64(n * s), 4, _ptr(g_raw), _stream()))
            nb = lib.nrf_mlp_backward_f16_workspace_bytes(self.mlp._m, C.c_int64(n * s))
            ws = self._workspace(nb)
            L.check(lib.nrf_mlp_backward_f16_lm(self.mlp._m, _ptr(feats), _ptr(dirs16), s, _ptr(g_raw), C.c_int64(n * s), _ptr(self.g_blob), _ptr(g_x), _ptr(ws),
                                                C.c_size_t(ws.numel()), _stream()))
            x = None
        else:
            emb, keep = self.embedder.forward(pts)
            x = torch.cat([emb, dirs[:, None, :].expand(n, s, dirs.shape[1]).reshape(n * s, -1)], 1).contiguous()
            keep_u8 = keep.to(torch.uint8)
            L.check(lib.nrf_mask_sigma_grad(_ptr(keep_u8), C.c_int64(n * s), 4, _ptr(g_raw), _stream()))
            ws_fn, bw_fn = (lib.nrf_mlp_backward_f16_workspace_bytes, lib.nrf_mlp_backward_f16) if self.mlp_backward == "f16" else (lib.nrf_mlp_backward_workspace_bytes, lib.nrf_mlp_backward)
            nb = ws_fn(self.mlp._m, C.c_int64(n * s))
            ws = self._workspace(nb)
            L.check(bw_fn(self.mlp._m, _ptr(x), _ptr(g_raw), C.c_int64(n * s), _ptr(self.g_blob), _ptr(g_x), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
        if self.hash_backward == "packed":
            nbh = lib.nrf_hash_backward_packed_workspace_bytes(self.embedder._h)
            if self._hws is None or self._hws.numel() < nbh:
                self._hws = torch.empty((int(nbh),), device="cuda", dtype=torch.uint8)
            L.check(lib.nrf_hash_backward_rays_packed(self.embedder._h, _ptr(pts), C.c_int64(n), s, _ptr(g_x), _ptr(self.g_table), _ptr(self._hws), C.c_size_t(self._hws.numel()), _stream()))
        else:
            L.check(lib.nrf_hash_backward_rays(self.embedder._h, _ptr(pts), C.c_int64(n), s, _ptr(g_x), _ptr(self.g_table), _stream()))
        self.last = dict(g_rgb=g_rgb, g_raw=g_raw, g_x=g_x, x=x, pts=pts)
        return loss_mse

    @staticmethod
    def _rng_u32(seed, stream, idx):
        """include/nrf_rng.h's nrf_rng_u32 on the host (three draws per level and step do not need a kernel)."""
        m = (1 << 64) - 1
        x = (seed + idx * 0x9E3779B97F4A7C15) & m
        x ^= (stream * 0xD1B54A32D192ED03) & m
        x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & m
        x ^= x >> 27; x = (x * 0x94D049BB133111EB) & m
        x ^= x >> 31
        return x >> 32

    def add_tv_loss(self):
        """loss += w * TotalVariationLoss(level) for every level (NeRF.h:255-300): accumulates into self.g_table and self.tv_loss."""
        e = self.embedder
        if self.tv_loss_weight <= 0 or e.mode != L.NRF_HASH_NGP:
            return
        self.tv_loss.zero_()
        b = math.exp((math.log(e.FinestResolution) - math.log(e.BaseResolution)) / (e.NLevels - 1))          # NeRF.h:265
        for level in range(e.NLevels):
            res = int(math.floor(b ** level * e.BaseResolution))
            cube = int(math.floor(min(max(float(np.float32(res) / np.float32(10.0)), e.BaseResolution - 1), e.FinestResolution - 1)))      # :269-273
            span = max(res - cube, 1)
            mv = np.array([self._rng_u32(self.seed, 18, (self.t * e.NLevels + level) * 3 + a) * span >> 32 for a in range(3)], np.int32)   # randint(0, res - cube), :276
            L.check(L.lib().nrf_hash_tv_loss(e._h, _ptr(self.table), level, mv.ctypes.data_as(C.c_void_p), cube, C.c_float(self.tv_loss_weight), _ptr(self.tv_loss),
                                             _ptr(self.g_table), _stream()))

    def step(self, rays_o, rays_d, target, render_params: NeRFRenderParams, cone_angle=None):
        """Optimizer->zero_grad(); Render; huber; backward; Optimizer->step() (NeRFExecutor.h:866-985)."""
        p = render_params
        if not p.ThinRay and cone_angle is None:
            raise L.NrfError("Trainer.step: ThinRay = False needs the batch's cone_angle (GetRayBatch / GetRays)")
        p.ReturnRaw, p.KeepIntermediates = True, True
        cone = None if p.ThinRay else cone_angle
        res = self.renderer.Render(0, 0, None, p, rays=(rays_o, rays_d, cone))
        s_out = p.NSamples + p.NImportance
        loss_mse = self.backward(res, target, s_out, p.WhiteBkgr, params=p, cone_angle=cone)
        self.add_tv_loss()
        if self.grad_sync is not None:                     # data-parallel replicas: mean of the ranks' gradients (nerfpp_amd/dist.py::GradSync)
            self.grad_sync(self.g_table, self.g_blob)
        self.t += 1
        b1, b2 = self.betas
        for prm, g, m, v in ((self.table, self.g_table, self.m_table, self.v_table), (self.blob, self.g_blob, self.m_blob, self.v_blob)):
            L.check(L.lib().nrf_adam_step(_ptr(prm), _ptr(g), _ptr(m), _ptr(v), C.c_int64(prm.numel()), C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                          C.c_float(self.eps), self.t, _stream()))
        self._push_params()
        return loss_mse, res

    @staticmethod
    def psnr(mse):
        return -10.0 * math.log(max(float(mse), 1e-30)) / math.log(10.0)       # NeRFExecutor.h:893
