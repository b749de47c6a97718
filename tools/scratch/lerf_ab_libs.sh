# same-call alternating A/B of the LeRF frame over library builds: usage (GPU box): bash tools/scratch/lerf_ab_libs.sh <name>...   (tune/<name>/libnerfpp_hip.so; "default" = the tree's)
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
  python - <<'PY'
import sys, time, os; sys.path.insert(0, ".")
import numpy as np, torch
from nerfpp_amd import _lib as L, scene, renderer as R
sc = scene.make_lerf_scene(); r = sc["renderer"]; r.keep_intermediates = False
K = scene.lego_K(800, 800); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
out = r.Render(800, 800, K, p, c2w=c2w); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): out = r.Render(800, 800, K, p, c2w=c2w)
torch.cuda.synchronize()
import hashlib
print(os.environ.get("NRF_LIB_PATH", "default").split("/")[-2] if os.environ.get("NRF_LIB_PATH") else "default", "%.2f ms / frame" % ((time.perf_counter() - t0) / 4 * 1e3),
      hashlib.sha256(out.Outputs.RenderedLangEmbedding.cpu().numpy().tobytes()).hexdigest()[:12], flush=True)
PY
done; done
