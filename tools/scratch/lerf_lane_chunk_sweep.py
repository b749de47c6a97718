"""LeRF frame (800x800, 64+128, one nrf_lerf_render_rows call, prompts set) over lanes x Chunk, same call.  usage (GPU box): python tools/scratch/lerf_lane_chunk_sweep.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from nerfpp_amd import _lib as L, scene, renderer as R
H = W = 800
sc = scene.make_lerf_scene(); r = sc["renderer"]
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
rng = np.random.RandomState(79)
pos = rng.randn(1, 768).astype(np.float32); pos /= np.linalg.norm(pos); neg = rng.randn(3, 768).astype(np.float32); neg /= np.linalg.norm(neg, axis=1, keepdims=True)
r.SetLeRFPrompts(pos, neg); r.keep_intermediates = False
ref = None
for rep in range(2):
    for chunk in (8192, 16384, 24576, 32768, 49152):
        for lanes in (1,):
            r.lanes = lanes
            p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=chunk, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
            out = r.Render(H, W, K, p, c2w=c2w); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                out = r.Render(H, W, K, p, c2w=c2w)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 4 * 1e3
            e = out.Outputs.RenderedLangEmbedding
            if ref is None: ref = e.clone()
            print(f"chunk {chunk:7d} lanes {lanes} : {dt:7.2f} ms / frame   same embedding as first config: {bool(torch.equal(e, ref))}", flush=True)
            del out
