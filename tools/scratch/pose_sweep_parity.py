"""Default split-precision render vs NRF_PREC_F32 (= the CPU oracle bit for bit) over a sweep of poses of the bench scene: whole 800x800 frames, every pixel value.
usage: python tools/scratch/pose_sweep_parity.py [n_poses] [hash|classic]   (classic: a 24-row band per pose -- its fp32 path is fp32 vector FMAs, ~0.5 s per band)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import scene as S, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
wl = sys.argv[2] if len(sys.argv) > 2 else "hash"
sc = S.make_hash_scene(mode="cu") if wl == "hash" else S.make_classic_scene()
chunk = 131072 if wl == "hash" else 8192
band = {} if wl == "hash" else dict(row0=388, rows=24)
K = S.lego_K(800, 800)
worst = 0.0
for i in range(n):
    theta, phi, rad = -180.0 + 360.0 * i / n, -30.0 + 25.0 * np.sin(i), 4.0 + 0.5 * np.cos(2 * i)
    c2w = S.pose_spherical(theta, phi, rad)
    a = sc["renderer"].Render(800, 800, K, S.lego_render_params(sc["bbox"], 64, 128, chunk, L.NRF_PREC_F16_SPLIT, KeepIntermediates="depths"), c2w=c2w, **band)
    b = sc["renderer"].Render(800, 800, K, S.lego_render_params(sc["bbox"], 64, 128, chunk, L.NRF_PREC_F32, KeepIntermediates="depths"), c2w=c2w, **band)
    err = (a.Outputs.RGBMap - b.Outputs.RGBMap).abs().max().item()
    zeq = bool(torch.equal(a.Extras["z_fine"], b.Extras["z_fine"]))
    derr = (a.Outputs.DepthMap - b.Outputs.DepthMap).abs().max().item()
    acc = b.Outputs.AccMap.mean().item()
    worst = max(worst, err)
    print(f"pose {i:2d} theta {theta:7.1f} phi {phi:6.1f} r {rad:4.2f}: max |rgb - fp32| {err:.3e}  depth {derr:.3e}  z_fine identical {zeq}  mean acc {acc:.3f}", flush=True)
    assert zeq and err < 1e-4
print("worst", worst)
