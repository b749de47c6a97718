"""How much of the hash encode's time is the ORDER of its points?  The frame's fine-pass sample points in ray-major order (64 lanes = 64 consecutive samples
of one ray: what the renderer feeds today) against patch-major order (64 lanes = one sample index of an 8x8 pixel patch), same kernel (nrf_dbg_hash_lm)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R
H = W = 800; ROWS = 64
sc = S.make_hash_scene(mode="cu"); K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
rp = S.lego_render_params(sc["bbox"], 64, 128, 131072, L.NRF_PREC_F16_SPLIT, KeepIntermediates="depths")
res = sc["renderer"].Render(H, W, K, rp, c2w=c2w, row0=368, rows=ROWS)
rays = res.Extras["rays_flat"]; zf = res.Extras["z_fine"]
n, s = zf.shape
pts = rays[:, None, 0:3] + rays[:, None, 3:6] * zf[..., None]                      # [n, s, 3] ray-major
pp = pts.reshape(ROWS // 8, 8, W // 8, 8, s, 3).permute(0, 2, 4, 1, 3, 5).contiguous()   # [patch_y, patch_x, sample, 8, 8, 3]: patch-major, 64 pixels of a patch adjacent
lib = L.lib(); h = sc["embedder"]._h
p = n * s
feats = torch.empty((16, p, 2), device="cuda", dtype=torch.float16); keep = torch.empty((p,), device="cuda", dtype=torch.uint8)
def run(x, name):
    x = x.reshape(-1, 3).contiguous()
    for _ in range(2): L.check(lib.nrf_dbg_hash_lm(h, C.c_void_p(x.data_ptr()), C.c_int64(p), 0, 0, -1, C.c_void_p(feats.data_ptr()), C.c_void_p(keep.data_ptr()), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): L.check(lib.nrf_dbg_hash_lm(h, C.c_void_p(x.data_ptr()), C.c_int64(p), 0, 0, -1, C.c_void_p(feats.data_ptr()), C.c_void_p(keep.data_ptr()), None))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-12s %.3f ms for %.1f M points = %.2f G points/s; scaled to a frame's 122.9 M: %.2f ms" % (name, ms, p / 1e6, p / ms / 1e6, ms * 122.88e6 / p), float(feats.float().abs().mean()))
run(pts, "ray-major")
run(pp, "patch-major")
