"""Training step time with nrf_mlp_set_params refreshing the weight images on the device (default) vs on the host (NRF_MLP_HOST_REPACK=1, read at handle creation).
usage (GPU box): python tools/scratch/train_repack_ab.py"""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import torch
    from nerfpp_amd import _lib as L, scene as S, renderer as R
    from nerfpp_amd.train import Trainer
    H = W = 800; N = 16384
    sc = S.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, N, device="cuda") * (H * W // N)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    tgt = torch.rand((N, 3), device="cuda")
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned")
    print("device-repacked images:", L.lib().nrf_mlp_device_repack_images(sc["mlp"]._m), flush=True)
    for rep in range(3):
        for _ in range(3): tr.step(o, d, tgt, rp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): lm, _ = tr.step(o, d, tgt, rp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"  step {dt*1e3:6.2f} ms  loss {float(lm[0]):.6f}", flush=True)
else:
    for host in ("0", "1", "0", "1"):
        print("NRF_MLP_HOST_REPACK =", host, flush=True)
        subprocess.run([sys.executable, __file__, "w"], env=dict(os.environ, NRF_MLP_HOST_REPACK=host), timeout=300)
