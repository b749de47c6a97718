# rocprofv3 kernel summary of the HashNeRF training step (N1 row; 16 384 rays, 64 + 128, matrix-core backward, binned hash backward; 12 steps): bash tools/scratch/train_kernel_stats.sh <tag>
tag=${1:-trainprof}
ROOTD=$PWD
cat > /tmp/train_only.py <<PY
import sys; sys.path.insert(0, "$ROOTD")
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
for _ in range(12): tr.step(o, d, tgt, rp)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag} -- python3 /tmp/train_only.py > $ROOTD/gpurun_out/${tag}.log 2>&1
cd $ROOTD
f=$(ls gpurun_out/${tag}/*/*_kernel_stats.csv | head -1); cp $f gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}
head -40 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
