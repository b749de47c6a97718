"""HashNeRF frame with the Chunk loop's lanes on disjoint halves of the chip (NRF_LANE_CU_MASK=1) against lanes that share it; chunk sizes with an even chunk count.
usage (GPU box): python tools/scratch/lane_mask_ab.py"""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import hashlib, torch
    from nerfpp_amd import _lib as L, scene as S
    sc = S.make_hash_scene(mode="cu"); r = sc["renderer"]
    K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    for chunk in (80000, 160000, 131072, 64000):
        rp = S.lego_render_params(sc["bbox"], 64, 128, chunk, L.NRF_PREC_F16_SPLIT)
        for _ in range(3): out = r.Render(800, 800, K, rp, c2w=c2w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = r.Render(800, 800, K, rp, c2w=c2w)
        torch.cuda.synchronize()
        print(f"  chunk {chunk:7d}: {(time.perf_counter() - t0) / 10 * 1e3:6.2f} ms / frame  {hashlib.sha256(out.Outputs.RGBMap.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
else:
    for env in ({"NRF_RENDER_LANES": "1"}, {"NRF_RENDER_LANES": "2"}, {"NRF_RENDER_LANES": "2", "NRF_LANE_CU_MASK": "1"}, {"NRF_RENDER_LANES": "4", "NRF_LANE_CU_MASK": "1"}):
        print(env, flush=True)
        subprocess.run([sys.executable, __file__, "w"], env=dict(os.environ, **env), timeout=300)
