"""LeRF frame time per library build (NRF_LIB_PATH set by the caller): 4 frames after a warm tile; prints ms per frame, the LeRF pass times and a digest of the embedding"""
import sys, os, time, hashlib, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
sc = S.make_lerf_scene(); r = sc["renderer"]
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
r.Render(800, 800, K, p, c2w=c2w, row0=0, rows=82); torch.cuda.synchronize()
r.Render(800, 800, K, p, c2w=c2w); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): res = r.Render(800, 800, K, p, c2w=c2w)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
e = res.Outputs.RenderedLangEmbedding
torch.save(e.cpu(), os.environ.get("LERF_OUT", "/tmp/lerf_emb.pt"))
print(os.environ.get("NRF_LIB_PATH", "default")[-40:], "ms/frame %.2f" % (dt * 1e3), hashlib.sha256(e.cpu().numpy().tobytes()).hexdigest()[:8], flush=True)
