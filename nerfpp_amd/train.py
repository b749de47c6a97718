"""One optimisation step of NeRFExecutor::Train (NeRFExecutor.h:862-995) over the C ABI -- SURVEY section 8f, row N1.

    render the ray batch (NeRFRenderer::Render with an explicit ray batch, :876) -> huber_loss(RGBMap, target) (:883)
    -> loss.backward() (:923) -> Adam(lr, betas (0.9, 0.99), eps 1e-15).step() (:539, :985)

Gradients flow through the FINE pass only (z_samples are detached, NeRFRenderer.h:429; rays and depths carry no parameters).
PyTorch owns the buffers (fp32 master parameters, Adam moments) and nothing else: every arithmetic step is a call into
libnerfpp_hip.so on the current HIP stream.
"""
import copy
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .modules import _HashBase, CuHashEmbedder, NeRF, NeRFSmall, _ptr, _stream, _dev_f32
from .renderer import NeRFRenderer, NeRFRenderParams, RngFill, StochasticPrecondition, TangentScatter


class Trainer:
    def __init__(self, embedder: _HashBase, embeddirs, mlp: NeRFSmall, table, mlp_blob, learning_rate=5e-4, betas=(0.9, 0.99), eps=1e-15,
                 tv_loss_weight=0.0, seed=0, mlp_backward="f32", hash_backward="f32", grad_sync=None, train_dense_budget=256 << 20):
        # two configurations of the reference's train loop (NeRFExecutor.h:862-995): hash grid + NeRFSmall (main.cpp:220-221) and the classic PE(10) / PE(4) + NeRF 8x256
        # (NeRFImpl is a legal TNeRF of the same loop; its embedders have no parameters: `table` is None / empty there)
        self.has_table = isinstance(embedder, _HashBase)
        if self.has_table != isinstance(mlp, NeRFSmall) or not (self.has_table or isinstance(mlp, NeRF)):
            raise L.NrfError("Trainer is built for hash-grid + NeRFSmall scenes and for positional-encoding + NeRF (classic) scenes")
        if not self.has_table and (mlp_backward != "f32" or hash_backward != "f32"):
            raise L.NrfError("the classic model trains through the fp32 layer kernels (mlp_backward = hash_backward = 'f32')")
        self.embedder, self.embeddirs, self.mlp = embedder, embeddirs, mlp
        self.renderer = NeRFRenderer(embedder, embeddirs, mlp)
        dev = "cuda"
        if not self.has_table:
            table = np.zeros((0,), np.float32)
        self.table = torch.as_tensor(np.ascontiguousarray(table, np.float32).reshape(-1) if not torch.is_tensor(table) else table.reshape(-1)).to(dev).contiguous()
        self.blob = torch.as_tensor(np.ascontiguousarray(mlp_blob, np.float32).reshape(-1) if not torch.is_tensor(mlp_blob) else mlp_blob.reshape(-1)).to(dev).contiguous()
        assert self.table.numel() == (embedder.table_elems() if self.has_table else 0) and self.blob.numel() == mlp.n_params
        self.m_table, self.v_table = torch.zeros_like(self.table), torch.zeros_like(self.table)
        self.m_blob, self.v_blob = torch.zeros_like(self.blob), torch.zeros_like(self.blob)
        self.g_table, self.g_blob = torch.zeros_like(self.table), torch.zeros_like(self.blob)
        self.lr, self.betas, self.eps, self.t = float(learning_rate), betas, float(eps), 0
        self._skipped, self._h_flags, self._flags_event, self._d_flags = 0, None, None, None          # deferred overflow words of a guarded step (_settle_flags)
        self.learning_rate0 = float(learning_rate)          # Params.LearningRate: the base of the exponential decay (NeRFExecutor.h:992-996)
        # TotalVariationLoss of the LibTorch HashEmbedder, weight 1e-6 in the reference for the first half of training (NeRFExecutor.h:896-913)
        self.tv_loss_weight, self.seed = float(tv_loss_weight), int(seed)
        self.tv_loss = torch.zeros((1,), device=dev)
        # "f32": layer-wise fp32 kernels (the parity path, pinned to the reference's autograd); "f16": one fused matrix-core kernel
        # (fp16 operands, fp32 accumulation, device-side loss scaling -- mlp_small_bwd_mfma.hip)
        if mlp_backward not in ("f32", "f16"):
            raise L.NrfError("mlp_backward must be 'f32' or 'f16'")
        self.mlp_backward = mlp_backward
        # "f32": one float atomic per feature; "packed": both features of an entry in one 64-bit fixed-point atomic (nrf_hash_backward_rays_packed);
        # "binned": the same fixed-point contributions merged per table range in LDS before they reach memory (nrf_hash_backward_rays_binned; equals "packed" bit for bit)
        if hash_backward not in ("f32", "packed", "binned"):
            raise L.NrfError("hash_backward must be 'f32', 'packed' or 'binned'")
        if hash_backward in ("packed", "binned") and embedder.NFeaturesPerLevel != 2:
            raise L.NrfError("hash_backward='%s' needs 2 features per level" % hash_backward)
        self.hash_backward = hash_backward
        self.grad_sync = grad_sync          # callable(g_table, g_blob) reducing the gradients across data-parallel ranks in place, or None
        self._hws = None
        self._ws = None
        # the baked dense pyramid of the render fast path is re-baked after every step's table upload: GBs at the render default -- so while training only the coarse
        # levels that fit `train_dense_budget` stay baked (256 MB: re-baking them costs ~0.1 ms per step and their lookups are the 2-load fast path instead of 8 hashed
        # corners: step 9.3 -> 8.5 ms, same bits; 0 / 16 / 64 / 1024 MB: 9.3 / 8.9 / 8.7 / 8.7 ms, docs/history/profiles/round4/r4m_train_dense_budget.log).  For ANY hash embedder --
        # the LibTorch HashEmbedder (the reference's TV-loss training configuration) included
        self._dense_budget_before = getattr(embedder, "dense_budget", None) if self.has_table else None
        if not self.has_table:
            self._push_params()
            return
        # never MORE than the embedder already had: a host that built it with budget 0 (or a small one) keeps the levels it turned off -- as the C++ drop-in does
        # (HipNeRFRenderer: min(previous budget, TrainDenseBudget))
        prev = self._dense_budget_before
        embedder.set_dense_budget(int(train_dense_budget) if prev is None else min(int(prev), int(train_dense_budget)))
        self._push_params()

    def close(self):
        """Give the embedder its baked dense image back (the budget it had before this Trainer switched it off): renders after training -- evaluation frames, a
        bench measurement -- then take the fast path again.  Idempotent."""
        if getattr(self, "_dense_budget_before", None) is not None:
            self.embedder.set_dense_budget(self._dense_budget_before)
            self._dense_budget_before = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _push_params(self):
        if self.has_table:
            self.embedder.set_table(self.table)
        L.check(L.lib().nrf_mlp_set_params(self.mlp._m, _ptr(self.blob), 1, _stream()))

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty((int(nbytes),), device="cuda", dtype=torch.uint8)
        return self._ws

    def backward(self, res, target, n_samples_out, white_bkgr, params=None, cone_angle=None, defer_flags=False):
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
        self.g_blob.zero_(); self.g_table.zero_()
        if not self.has_table:
            # classic model (NeRFRenderer.h:175-184 with Embedder / Embedder / NeRF): the encodings carry no parameters, so the chain ends at the network's own gradient
            emb, _ = self.embedder.forward(pts)
            x = torch.cat([emb, dirs[:, None, :].expand(n, s, dirs.shape[1]).reshape(n * s, -1)], 1).contiguous()
            nb = lib.nrf_mlp_backward_workspace_bytes(self.mlp._m, C.c_int64(n * s))
            ws = self._workspace(nb)
            L.check(lib.nrf_mlp_backward(self.mlp._m, _ptr(x), _ptr(g_raw), C.c_int64(n * s), _ptr(self.g_blob), None, _ptr(ws), C.c_size_t(ws.numel()), _stream()))
            self.last = dict(g_rgb=g_rgb, g_raw=g_raw, g_x=None, x=x, pts=pts)
            self.overflow = False
            self._d_flags = None
            return loss_mse
        g_x = torch.empty((n * s, in_ch), device=rays.device)
        lm = (self.mlp_backward == "f16" and isinstance(self.embedder, CuHashEmbedder) and self.embedder.NLevels == 16 and self.embedder.NFeaturesPerLevel == 2
              and dirs.shape[1] == 16)
        if lm:
            # the fast path's own layout: level-major fp16 hash features + one fp16 direction row per ray; no [p, 48] fp32 input is formed
            dirs16 = dirs.to(torch.float16).contiguous()
            nb = lib.nrf_mlp_backward_f16_workspace_bytes(self.mlp._m, C.c_int64(n * s))
            ws = self._workspace(nb)
            # ... and where the forward render of THIS batch was the renderer's last call (a single-chunk render of the feature-reusing fast path), the features it
            # encoded are still in its workspace -- coarse columns, new samples' columns, the merge map: the same points, the same kernel, the same bits -- and the
            # fine points are not encoded a second time (0.6 ms of a 5.9 ms step)
            view = self._render_features(res, n, s) if (fine and cone_angle is None and not (p is not None and p.StochasticPreconditioningAlpha > 0)) else None
            if view is not None:
                L.check(lib.nrf_mask_sigma_grad_src(C.c_void_p(view["keep"]), C.c_void_p(view["src"]), C.c_int64(n * s), 4, _ptr(g_raw), _stream()))
                L.check(lib.nrf_mlp_backward_f16_lm_src(self.mlp._m, C.c_void_p(view["feats"]), C.c_int64(view["cols"]), C.c_void_p(view["src"]), _ptr(dirs16), s, _ptr(g_raw),
                                                        C.c_int64(n * s), _ptr(self.g_blob), _ptr(g_x), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
            else:
                feats = torch.empty((16, n * s, 2), device=rays.device, dtype=torch.float16)
                keep_u8 = torch.empty((n * s,), device=rays.device, dtype=torch.uint8)
                ptsc = pts.contiguous()
                L.check(lib.nrf_hash_encode_lm_f16(self.embedder._h, _ptr(ptsc), C.c_int64(n * s), _ptr(feats), _ptr(keep_u8), _stream()))
                L.check(lib.nrf_mask_sigma_grad(_ptr(keep_u8), C.c_int64(n * s), 4, _ptr(g_raw), _stream()))
                L.check(lib.nrf_mlp_backward_f16_lm(self.mlp._m, _ptr(feats), _ptr(dirs16), s, _ptr(g_raw), C.c_int64(n * s), _ptr(self.g_blob), _ptr(g_x), _ptr(ws),
                                                    C.c_size_t(ws.numel()), _stream()))
            self.reused_render_features = view is not None
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
        if self.hash_backward == "binned":
            nbh = lib.nrf_hash_backward_binned_workspace_bytes_for(self.embedder._h, C.c_int64(n), s)      # records for this batch, not for a whole 2^18-point pass
            if self._hws is None or self._hws.numel() < nbh:
                self._hws = torch.empty((int(nbh),), device="cuda", dtype=torch.uint8)
            L.check(lib.nrf_hash_backward_rays_binned(self.embedder._h, _ptr(pts), C.c_int64(n), s, _ptr(g_x), _ptr(self.g_table), _ptr(self._hws), C.c_size_t(self._hws.numel()), _stream()))
        elif self.hash_backward == "packed":
            nbh = lib.nrf_hash_backward_packed_workspace_bytes(self.embedder._h)
            if self._hws is None or self._hws.numel() < nbh:
                self._hws = torch.empty((int(nbh),), device="cuda", dtype=torch.uint8)
            L.check(lib.nrf_hash_backward_rays_packed(self.embedder._h, _ptr(pts), C.c_int64(n), s, _ptr(g_x), _ptr(self.g_table), _ptr(self._hws), C.c_size_t(self._hws.numel()), _stream()))
        else:
            L.check(lib.nrf_hash_backward_rays(self.embedder._h, _ptr(pts), C.c_int64(n), s, _ptr(g_x), _ptr(self.g_table), _stream()))
        self.last = dict(g_rgb=g_rgb, g_raw=g_raw, g_x=g_x, x=x, pts=pts)
        # fp16 gradient chain: was anything in it, or anything it produced, not finite?  (nrf_mlp_backward_f16_flags; the caller skips the step then.)  Without gradient
        # exchange the host does not wait for the answer: the optimizer step is guarded ON THE DEVICE by the two words (nrf_adam_step_guarded) and the host reads them
        # from pinned memory before the next step begins (_settle_flags) -- a wait in the middle of every step was 0.3 ms of GPU idle time in a 6 ms step.
        self.overflow = False
        self._d_flags = None
        if self.mlp_backward == "f16":
            if self.grad_sync is None and defer_flags:
                if self._h_flags is None:
                    self._h_flags = torch.zeros((2,), dtype=torch.int32).pin_memory()
                L.check(lib.nrf_mlp_backward_f16_flags_async(_ptr(self._ws), C.c_void_p(self._h_flags.data_ptr()), _stream()))
                self._flags_event = torch.cuda.Event(); self._flags_event.record(torch.cuda.current_stream())
                self._d_flags = lib.nrf_mlp_backward_f16_flags_device(_ptr(self._ws))
            else:
                fl = (C.c_uint32 * 2)()
                L.check(lib.nrf_mlp_backward_f16_flags(_ptr(self._ws), fl, _stream()))
                self.overflow = bool(fl[0] or fl[1])
        return loss_mse

    def _render_features(self, res, n, s):
        """The renderer's feature view (NeRFRenderer.feature_view) if it still belongs to `res`: recorded on the result (FeatureView) by the Render call that produced it, and the
        renderer has rendered nothing since (same chunk serial).  None otherwise, and with reuse_render_features = False."""
        if not getattr(self, "reuse_render_features", True):
            return None
        mine = getattr(res, "FeatureView", None)
        now = self.renderer.feature_view() if hasattr(self.renderer, "feature_view") else None
        if not mine or not now or mine != now or now["n"] != n or now["sf"] != s:
            return None
        return now

    def _settle_flags(self):
        """A guarded step whose overflow words have not been looked at yet: wait for their copy (issued behind that step's backward: long done when the next step
        begins), and if the chain had overflowed -- the device skipped the update -- take the step count back."""
        ev = getattr(self, "_flags_event", None)
        if ev is None:
            return
        ev.synchronize()
        self._flags_event = None
        self.overflow = bool(int(self._h_flags[0]) or int(self._h_flags[1]))
        if self.overflow:
            self.t -= 1
            self._skipped += 1

    @property
    def skipped_steps(self):
        self._settle_flags()
        return self._skipped

    @skipped_steps.setter
    def skipped_steps(self, v):
        self._skipped = int(v)

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
        if self.tv_loss_weight <= 0 or not self.has_table or e.mode != L.NRF_HASH_NGP:
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

    def step(self, rays_o, rays_d, target, render_params: NeRFRenderParams, cone_angle=None, global_step=None, n_iters=None, lrate_decay=None):
        """Optimizer->zero_grad(); Render; huber; backward; Optimizer->step(); learning-rate decay (NeRFExecutor.h:866-996).
        global_step / n_iters / lrate_decay (the executor's loop counter, Params.NIters and Params.LRateDecay): when given, the TV regulariser is added only
        for global_step < n_iters / 2 (:896-913) and after the step lr = learning_rate * 0.1^(global_step / (lrate_decay * 1000)) (:992-996).
        The counter-based draws of the stochastic branches are keyed by (Seed, step): every iteration draws afresh, as the reference does from torch's global
        generator.  The caller's render_params object is not modified."""
        if not render_params.ThinRay and cone_angle is None:
            raise L.NrfError("Trainer.step: ThinRay = False needs the batch's cone_angle (GetRayBatch / GetRays)")
        self._settle_flags()                               # (the previous step's overflow words: the step count below must be final)
        p = copy.copy(render_params)
        # the fine depths, not the coarse pass's raw rows: gradients flow through the fine pass only (NeRFRenderer.h:429), and without a raw_coarse output the render is the
        # default path -- sigma net alone in the coarse pass, hash features kept per column for the fine pass -- whose features the backward then reads (feature_view)
        p.ReturnRaw, p.KeepIntermediates = True, "depths"
        p.Seed = (int(render_params.Seed) + 0x9E3779B97F4A7C15 * self.t) & ((1 << 64) - 1)
        cone = None if p.ThinRay else cone_angle
        res = self.renderer.Render(0, 0, None, p, rays=(rays_o, rays_d, cone))
        s_out = p.NSamples + p.NImportance
        loss_mse = self.backward(res, target, s_out, p.WhiteBkgr, params=p, cone_angle=cone, defer_flags=True)
        if global_step is None or n_iters is None or global_step < n_iters / 2:
            self.add_tv_loss()
        # A non-finite gradient in the fp16 chain: no optimizer step (the moments would be poisoned for good).  With data-parallel replicas the flag is made
        # collective BEFORE any gradient is exchanged (GradSync.reduce_or_skip): every rank skips, none sums a peer's inf, all keep the same t / seed / lr.
        skip = bool(self.overflow)
        if self.grad_sync is not None:                     # mean of the ranks' gradients (nerfpp_amd/dist.py::GradSync)
            if hasattr(self.grad_sync, "reduce_or_skip"):
                skip = self.grad_sync.reduce_or_skip(skip, self.g_table, self.g_blob)
            elif not skip:
                self.grad_sync(self.g_table, self.g_blob)   # a plain callable: single-process hooks
        if skip:
            self._skipped += 1
            return loss_mse, res
        self.t += 1                                        # (a guarded step: taken back by _settle_flags if the device skipped the update)
        b1, b2 = self.betas
        dfl, nfl = (C.c_void_p(self._d_flags), 2) if self._d_flags else (None, 0)
        for prm, g, m, v in ((self.table, self.g_table, self.m_table, self.v_table), (self.blob, self.g_blob, self.m_blob, self.v_blob)):
            if prm.numel():
                L.check(L.lib().nrf_adam_step_guarded(_ptr(prm), _ptr(g), _ptr(m), _ptr(v), C.c_int64(prm.numel()), C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                                      C.c_float(self.eps), self.t, dfl, nfl, _stream()))
        self._push_params()
        if global_step is not None and lrate_decay:
            self.lr = self.learning_rate0 * math.pow(0.1, float(global_step) / (float(lrate_decay) * 1000.0))      # :992-996
        return loss_mse, res

    # ---- checkpoint interchange (NeRFExecutor::SaveCheckpoint / the restore branch of Initialize, NeRFExecutor.h:1055-1070, :540-566) ----
    def _param_layout(self):
        """[(name, offset, shape)] of the embedder's and the model's parameters in the reference's optimizer order (embedder first, :508-535)."""
        if not self.has_table:
            # NeRFImpl's registration order (NeRF.cpp:76-91): pts_linears_i, views_linears_0, feature_linear, alpha_linear, rgb_linear | output_linear; weight then bias
            d, off, mlp = self.mlp.desc, 0, []
            def lin(name, o, i):
                nonlocal off
                mlp.append((f"model_{name}.weight", off, (o, i))); off += o * i
                mlp.append((f"model_{name}.bias", off, (o,))); off += o
            for l in range(d.depth):
                lin(f"pts_linears_{l}", d.width, d.input_ch if l == 0 else (d.width + d.input_ch if l - 1 == d.skip else d.width))
            if d.use_viewdirs:
                lin("views_linears_0", d.width // 2, d.input_ch_views + d.width); lin("feature_linear", d.width, d.width); lin("alpha_linear", 1, d.width); lin("rgb_linear", 3, d.width // 2)
            else:
                lin("output_linear", d.output_ch, d.width + d.input_ch)
            assert off == self.blob.numel()
            return [], mlp
        e, out = self.embedder, []
        rows, F = 1 << e.Log2HashmapSize, e.NFeaturesPerLevel
        if e.mode == L.NRF_HASH_NGP:
            emb = [(f"{e.name}_embeddings_{l}.weight", l * rows * F, (rows, F)) for l in range(e.NLevels)]
        else:
            emb = [(f"{e.name}_embeddings", 0, (rows * e.NLevels, F))]
        d, off, mlp = self.mlp.desc, 0, []
        for l in range(d.num_layers):
            o, i = ((1 + d.geo_feat_dim) if l == d.num_layers - 1 else d.hidden_dim), (d.input_ch if l == 0 else d.hidden_dim)
            mlp.append((f"model_sigma_net_{l}.weight", off, (o, i))); off += o * i
        for l in range(d.num_layers_color):
            o, i = (3 if l == d.num_layers_color - 1 else d.hidden_dim_color), ((d.input_ch_views + d.geo_feat_dim) if l == 0 else d.hidden_dim_color)
            mlp.append((f"model_color_net_{l}.weight", off, (o, i))); off += o * i
        assert off == self.blob.numel()
        return emb, mlp

    def SaveCheckpoint(self, path, global_step=0):
        """embedder_checkpoint.pt, model_checkpoint.pt, start_checkpoint.pt and optimizer_checkpoint.pt as the reference writes them: its executor restores
        from such a directory (all four must exist, :541-546), Adam moments and step included."""
        self._settle_flags()
        from . import checkpoint as CK
        from collections import OrderedDict
        emb, mlp = self._param_layout()
        cut = lambda t, off, shape: t[off:off + int(np.prod(shape))].reshape(shape).detach().cpu().numpy()
        bufs = None
        if self.has_table and self.embedder.mode == L.NRF_HASH_CU:
            e = self.embedder
            ls = ((1 << e.Log2HashmapSize) >> 4) << 4
            bufs = OrderedDict([(f"{e.name}_primes", np.asarray(e.Primes, np.int32).reshape(e.NLevels, 1, 3)),
                                (f"{e.name}_biases", (np.zeros((e.NLevels, 3), np.float32) if e.Biases is None else np.asarray(e.Biases, np.float32).reshape(e.NLevels, 3))),
                                (f"{e.name}_feat_local_size", np.full(e.NLevels, ls, np.int32)), (f"{e.name}_feat_local_idx", (np.arange(e.NLevels) * ls).astype(np.int32))])
        moments = None if self.t == 0 else \
            [(cut(self.m_table, o, sh), cut(self.v_table, o, sh)) for _, o, sh in emb] + [(cut(self.m_blob, o, sh), cut(self.v_blob, o, sh)) for _, o, sh in mlp]
        if moments is None:
            moments = [None] * (len(emb) + len(mlp))
        # (the positional encodings have no parameters: the reference writes an embedder_checkpoint.pt of an empty module there; restoring does not need it, :541-546)
        CK.SaveCheckpoint(path, embedder=OrderedDict((n, cut(self.table, o, sh)) for n, o, sh in emb) if self.has_table else None, embedder_buffers=bufs,
                          model=OrderedDict((n, cut(self.blob, o, sh)) for n, o, sh in mlp), global_step=global_step,
                          optimizer=dict(moments=moments, step=self.t, lr=self.lr, betas=self.betas, eps=self.eps))

    def LoadCheckpoint(self, path):
        """Restore parameters, Adam moments / step / lr and return the start step -- only under the reference's own condition (:541-546: start, optimizer and
        model files all present); otherwise nothing is touched and None is returned (the reference then initialises afresh)."""
        from . import checkpoint as CK
        if not CK.WouldRestore(path):
            return None
        ck = CK.LoadCheckpoint(path)
        emb, mlp = self._param_layout()
        dev = self.table.device
        put = lambda dst, off, a: dst[off:off + a.size].copy_(torch.as_tensor(np.ascontiguousarray(a, np.float32).reshape(-1)).to(dev))
        if "embedder" in ck:
            for n, o, sh in emb:
                put(self.table, o, ck["embedder"][n])
        for n, o, sh in mlp:
            put(self.blob, o, ck["model"][n])
        opt = ck["optimizer"]
        assert len(opt["moments"]) == len(emb) + len(mlp), "optimizer_checkpoint.pt does not match this model's parameter list"
        for (n, o, sh), mv in zip(emb + mlp, opt["moments"]):
            m_dst, v_dst = (self.m_table, self.v_table) if (n, o, sh) in emb else (self.m_blob, self.v_blob)
            if mv is None:
                m_dst[o:o + int(np.prod(sh))].zero_(); v_dst[o:o + int(np.prod(sh))].zero_()
            else:
                put(m_dst, o, mv[0]); put(v_dst, o, mv[1])
        self.t, self.lr = int(opt["step"]), float(opt["lr"])
        self.betas, self.eps = tuple(opt["betas"]), float(opt["eps"])
        self._push_params()
        return ck.get("start", 0)

    @staticmethod
    def psnr(mse):
        return -10.0 * math.log(max(float(mse), 1e-30)) / math.log(10.0)       # NeRFExecutor.h:893


# ------------------------------------------------------------------------------------------------
# N1, LeRF branch of the optimisation step (NeRFExecutor.h:955-982)
# ------------------------------------------------------------------------------------------------
def HuberRowsNanmean(pred, target, delta=1.25, want_grad=True):
    """lang_loss = huber_loss(pred, target, reduction none, delta).sum(-1).nanmean() (NeRFExecutor.h:970-974) -> (loss [1] on the device, d loss / d pred or None)."""
    p = _dev_f32(pred); t = _dev_f32(target)
    n, e = p.shape
    loss = torch.empty((1,), device=p.device)
    grad = torch.empty_like(p) if want_grad else None
    L.check(L.lib().nrf_huber_rows_nanmean(_ptr(p), _ptr(t), C.c_int64(n), int(e), C.c_float(delta), _ptr(loss), _ptr(grad), _stream()))
    return loss, grad


def LeRFHeadBackward(lerf, emb, keep, z, rays_d, g_rendered, noise=None, noise_std=0.0, want_g_emb=True):
    """Backward of the fine pass of LeRFRenderer::RenderRays downstream of the language grid (nrf_lerf_head_backward): emb [n*s, in], keep [n*s] bool / uint8 or None,
    z [n, s], rays_d [n, 3], g_rendered [n, E] -> dict(g_params [blob], g_emb [n*s, in], rendered [n, E], weights [n, s]) (the recomputed fp32 forward rides along)."""
    emb = _dev_f32(emb); z = _dev_f32(z); d = _dev_f32(rays_d).reshape(-1, 3).contiguous(); g = _dev_f32(g_rendered)
    n, s = z.shape
    E = lerf.GetLangEmbedDim()
    k8 = None if keep is None else (keep if keep.dtype == torch.uint8 else keep.to(torch.uint8)).contiguous()
    g_params = torch.zeros((lerf.n_params,), device=emb.device)
    g_emb = torch.empty_like(emb) if want_g_emb else None
    rendered = torch.empty((n, E), device=emb.device); weights = torch.empty((n, s), device=emb.device)
    nz = None if noise is None else _dev_f32(noise)
    nb = L.lib().nrf_lerf_head_backward_workspace_bytes(lerf._m, C.c_int64(n), int(s))
    ws = torch.empty((int(nb),), device=emb.device, dtype=torch.uint8)
    L.check(L.lib().nrf_lerf_head_backward(lerf._m, _ptr(emb), _ptr(k8), _ptr(z), _ptr(d), 3, C.c_int64(n), int(s), _ptr(nz), C.c_float(noise_std), _ptr(g), _ptr(g_params),
                                           _ptr(g_emb), _ptr(rendered), _ptr(weights), _ptr(ws), C.c_size_t(ws.numel()), _stream()))
    return dict(g_params=g_params, g_emb=g_emb, rendered=rendered, weights=weights)


class LeRFTrainer:
    """The LeRF half of NeRFExecutor::Train's loop body (NeRFExecutor.h:955-985): LeRFRenderer->Render on the ray batch -> lang_loss (huber, delta 1.25, row sums,
    nanmean) -> lang_loss.backward() into the LeRF head and the language grid -> Adam (the executor's one optimizer, lr / betas (0.9, 0.99) / eps 1e-15, :539).
    The render is the library's fused pass; the backward is ONE library call (nrf_lerf_backward_points: the language grid's encode, the head's recomputed fp32 forward and
    backward, the grid's scatter).  PyTorch owns the buffers (fp32 master parameters, Adam moments) only."""

    def __init__(self, renderer, lang_table, lerf_blob, learning_rate=5e-4, betas=(0.9, 0.99), eps=1e-15, delta=1.25):
        from .renderer import LeRFRenderer
        if not isinstance(renderer, LeRFRenderer) or not renderer._r:
            raise L.NrfError("LeRFTrainer needs a LeRFRenderer on the library path (CuHashEmbedder L16 F8 language grid + a LeRF head of the built family)")
        self.renderer, self.embedder, self.lerf = renderer, renderer.LangEmbedFn, renderer.Lerf
        dev = "cuda"
        as_dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32).reshape(-1) if not torch.is_tensor(a) else a.reshape(-1)).to(dev).contiguous()
        self.table, self.blob = as_dev(lang_table), as_dev(lerf_blob)
        assert self.table.numel() == self.embedder.table_elems() and self.blob.numel() == self.lerf.n_params
        self.m_table, self.v_table = torch.zeros_like(self.table), torch.zeros_like(self.table)
        self.m_blob, self.v_blob = torch.zeros_like(self.blob), torch.zeros_like(self.blob)
        self.g_table, self.g_blob = torch.zeros_like(self.table), torch.zeros_like(self.blob)
        self.lr, self.betas, self.eps, self.delta, self.t = float(learning_rate), betas, float(eps), float(delta), 0
        self._ws = None
        # while the table moves only the coarse levels that fit 256 MB stay baked (as Trainer), never more than the embedder had
        self._dense_budget_before = getattr(self.embedder, "dense_budget", None)
        if self._dense_budget_before is not None:
            self.embedder.set_dense_budget(min(int(self._dense_budget_before), 256 << 20))
        self._push_params()

    def close(self):
        if getattr(self, "_dense_budget_before", None) is not None:
            self.embedder.set_dense_budget(self._dense_budget_before)
            self._dense_budget_before = None

    def _push_params(self):
        self.embedder.set_table(self.table)
        L.check(L.lib().nrf_mlp_set_params(self.lerf._m, _ptr(self.blob), 1, _stream()))

    def backward(self, res, target, params, cone_angle=None):
        """lang_loss and the gradients of one rendered batch (fills self.g_table / self.g_blob); returns the loss [1] on the device."""
        lib = L.lib()
        rays = res.Extras["rays_flat"]
        n, stride = rays.shape
        z = res.Extras["z_fine"]
        s = z.shape[1]
        rendered = res.Outputs.RenderedLangEmbedding.reshape(n, -1).contiguous()
        loss, g = HuberRowsNanmean(rendered, target, self.delta)
        pts = torch.empty((n * s, 3), device=rays.device)
        L.check(lib.nrf_points(_ptr(rays), stride, _ptr(z), C.c_int64(n), s, _ptr(pts), _stream()))
        p = params
        seed = int(p.Seed)
        if p.StochasticPreconditioningAlpha > 0:                                     # LeRFRenderer.cpp:155-164
            pn = RngFill(seed, L.NRF_RNG_PRECOND, 0, n * s * 3, normal=True, device=rays.device)
            pts = StochasticPrecondition(pts, pn, float(p.StochasticPreconditioningAlpha), p.BoundingBox)
        if cone_angle is not None:                                                   # TangentScatter (:166)
            ur = RngFill(seed, L.NRF_RNG_R_FINE, 0, n * s, device=rays.device); ut = RngFill(seed, L.NRF_RNG_THETA_FINE, 0, n * s, device=rays.device)
            pts = TangentScatter(pts.reshape(n, s, 3), z, float(cone_angle), rays[:, 3:6].contiguous(), p.BoundingBox, ur, ut).reshape(n * s, 3)
        noise_std = float(p.RawNoiseStd)
        noise = RngFill(seed, L.NRF_RNG_NOISE_FINE, 0, n * s, normal=True, device=rays.device) if noise_std > 0 else None
        self.g_blob.zero_(); self.g_table.zero_()
        nb = lib.nrf_lerf_backward_points_workspace_bytes(self.renderer._r, C.c_int64(n), int(s))
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty((int(nb),), device="cuda", dtype=torch.uint8)
        pts = pts.contiguous()
        # the language features of exactly these points are what the forward render has just encoded (thin rays, no preconditioning: pts = o + d z_fine): read them through
        # its merge map instead of encoding the points again -- if `res` is that renderer's LAST render (same chunk serial) and a one-chunk one (nrf_lerf_renderer_last_features)
        view = None
        if getattr(self, "reuse_render_features", True) and cone_angle is None and not p.StochasticPreconditioningAlpha > 0:
            mine, now = getattr(res, "FeatureView", None), self.renderer.feature_view()
            if mine is not None and now is not None and mine["serial"] == now["serial"] and now["n"] == n and now["sf"] == s:
                view = now
        self.reused_render_features = view is not None
        if view is not None:
            L.check(lib.nrf_lerf_backward_points_src(self.renderer._r, C.c_void_p(view["feats"]), C.c_int64(view["cols"]), C.c_void_p(view["keep"]), C.c_void_p(view["src"]), _ptr(pts),
                                                     _ptr(z), C.c_void_p(rays.data_ptr() + 12), stride, C.c_int64(n), int(s), _ptr(noise), C.c_float(noise_std), _ptr(g),
                                                     _ptr(self.g_blob), _ptr(self.g_table), _ptr(self._ws), C.c_size_t(self._ws.numel()), _stream()))
        else:
            L.check(lib.nrf_lerf_backward_points(self.renderer._r, _ptr(pts), _ptr(z), C.c_void_p(rays.data_ptr() + 12), stride, C.c_int64(n), int(s), _ptr(noise),
                                                 C.c_float(noise_std), _ptr(g), _ptr(self.g_blob), _ptr(self.g_table), _ptr(self._ws), C.c_size_t(self._ws.numel()), _stream()))
        self.last = dict(g_rendered=g, pts=pts)
        return loss

    def step(self, rays_o, rays_d, target_lang_embedding, render_params: NeRFRenderParams, cone_angle=None):
        """Render; lang_loss; backward; Adam step on the language grid and the head (NeRFExecutor.h:955-985) -> (lang_loss [1] on the device, the render result)."""
        p = copy.copy(render_params)
        p.KeepIntermediates, p.ReturnWeights = True, True
        p.Seed = (int(render_params.Seed) + 0x9E3779B97F4A7C15 * self.t) & ((1 << 64) - 1)
        cone = None if p.ThinRay else cone_angle
        res = self.renderer.Render(0, 0, None, p, rays=(rays_o, rays_d, cone))
        loss = self.backward(res, target_lang_embedding, p, cone)
        self.t += 1
        b1, b2 = self.betas
        for prm, g, m, v in ((self.table, self.g_table, self.m_table, self.v_table), (self.blob, self.g_blob, self.m_blob, self.v_blob)):
            L.check(L.lib().nrf_adam_step(_ptr(prm), _ptr(g), _ptr(m), _ptr(v), C.c_int64(prm.numel()), C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                          C.c_float(self.eps), self.t, _stream()))
        self._push_params()
        return loss, res
