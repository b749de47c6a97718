# rocprofv3 kernel summary of the LeRF frame (split precision only, 3 frames): usage on the GPU box: bash tools/scratch/lerf_prof.sh <tag>
tag=${1:-lerfprof}
ROOTD=$PWD
cat > /tmp/lerf_only.py <<PY
import sys; sys.path.insert(0, "$ROOTD")
import numpy as np, torch
from nerfpp_amd import _lib as L, scene, renderer as R
sc = scene.make_lerf_scene(); r = sc["renderer"]; r.keep_intermediates = False
K = scene.lego_K(800, 800); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
rng = np.random.RandomState(79)
pos = rng.randn(1, 768).astype(np.float32); pos /= np.linalg.norm(pos); neg = rng.randn(3, 768).astype(np.float32); neg /= np.linalg.norm(neg, axis=1, keepdims=True)
r.SetLeRFPrompts(pos, neg)
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
for _ in range(3):
    out = r.Render(800, 800, K, p, c2w=c2w)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag} -- python3 /tmp/lerf_only.py > $ROOTD/gpurun_out/${tag}.log 2>&1
cd $ROOTD
f=$(ls gpurun_out/${tag}/*/*_kernel_stats.csv | head -1); cp $f gpurun_out/${tag}_kernel_stats.csv
g=$(ls gpurun_out/${tag}/*/*_kernel_trace.csv | head -1); python3 - "$g" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"][:60]
    a = agg.setdefault(k, dict(n=0, vgpr=r.get("VGPR_Count"), agpr=r.get("Accum_VGPR_Count"), lds=r.get("LDS_Block_Size"), grid=r.get("Grid_Size"), wg=r.get("Workgroup_Size")))
    a["n"] += 1
for k, a in agg.items(): print(k, a)
PY
rm -rf gpurun_out/${tag}
head -14 gpurun_out/${tag}_kernel_stats.csv | cut -c1-170
