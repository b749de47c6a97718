"""How far apart are two fp32-GRADE implementations of the classic training step's gradient?  The same step (same rays, same weights) with the layer products as
rocBLAS sgemm, as the hand-written FMA kernels (NRF_FP32_GEMM=0), as f16x3 and as bf16x3 split-precision matrix-core products; one process per configuration
(the rocBLAS switch is read once).  Prints every pair's max-over-max and norm-wise difference of the parameter gradient."""
import itertools, json, os, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CONFIGS = {"rocblas": {"NRF_TRAIN_GEMM": "f32"}, "fma": {"NRF_TRAIN_GEMM": "f32", "NRF_FP32_GEMM": "0"}, "f16x3": {"NRF_TRAIN_GEMM": "f16x3"}, "bf16x3": {"NRF_TRAIN_GEMM": "bf16x3"}}
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from nerfpp_amd import _lib as L, scene as S, renderer as R
    from nerfpp_amd.train import Trainer
    K = S.lego_K(200, 200); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(200, 200, K, c2w)
    o = o.reshape(-1, 3)[::20][:1500].contiguous(); d = d.reshape(-1, 3)[::20][:1500].contiguous()
    tgt = torch.rand((o.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    sc = S.make_classic_scene()
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], None, sc["mlp_blob"], learning_rate=5e-4)
    rp = R.NeRFRenderParams(NSamples=32, NImportance=32, Chunk=2048, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX,
                            Precision=L.NRF_PREC_F32)
    tr.step(o, d, tgt, rp)
    torch.save(tr.g_blob.cpu(), sys.argv[2])
    sys.exit(0)
import torch
tmp = tempfile.mkdtemp()
g = {}
for name, env in CONFIGS.items():
    path = os.path.join(tmp, name + ".pt")
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=dict(os.environ, **env), check=True, timeout=300, stderr=subprocess.DEVNULL)
    g[name] = torch.load(path).double()
for a, b in itertools.combinations(CONFIGS, 2):
    da = g[a] - g[b]
    print(json.dumps({"pair": a + " vs " + b, "max_over_max": float(da.abs().max() / g[b].abs().max()), "norm_wise": float(da.norm() / g[b].norm())}), flush=True)
