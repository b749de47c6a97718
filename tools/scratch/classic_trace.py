"""diagnostic build (EXTRA=-DNRF_NERF_TRACE, NRF_LIB_PATH): where wave 0 of each workgroup of k_mlp_nerf_split spends its cycles; arg `zero`: all-zero weights"""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, modules as M, renderer as R
H = W = 800
sc = S.make_classic_scene(); K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
r = sc["renderer"]
if len(sys.argv) > 1 and sys.argv[1] == "zero":
    mlp = M.NeRF(8, 256, sc["embedder"].GetOutputDims(), sc["embeddirs"].GetOutputDims(), 5, (4,), True, "model", params=np.zeros_like(sc["mlp_blob"]))
    r = R.NeRFRenderer(sc["embedder"], sc["embeddirs"], mlp)
rp = S.lego_render_params(sc["bbox"], 64, 128, 8192, L.NRF_PREC_F16_SPLIT)
r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize()
lib = C.CDLL(os.environ["NRF_LIB_PATH"])
buf = (C.c_ulonglong * (256 * 8))()
lib.nrf_dbg_nerf_trace(None, 1)
t0 = time.perf_counter(); r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
lib.nrf_dbg_nerf_trace(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
it = a[:, 5].sum()
names = ["tile loops", "-", "wait+barrier", "input encoding", "iteration total"]
print("frame %.1f ms; iterations %d; per iteration (cycles, mean over workgroups); clock %.2f GHz" % (dt * 1e3, int(it), a[:, 4].sum() / 256 / dt / 1e9))
for i, n in enumerate(names): print("  %-16s %9.0f" % (n, a[:, i].sum() / it))
t4 = a[:, 6].sum() / it / 8; t16 = a[:, 7].sum() / it / 48
step = (t16 - t4) / 12
print("  one 4-step loop %.0f, one 16-step loop %.0f cycles -> %.1f cycles per k-step (3 MFMAs = 96 pipe cycles) + %.0f fixed per loop" % (t4, t16, step, t4 - 4 * step))
