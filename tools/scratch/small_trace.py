"""diagnostic build (EXTRA='-DNRF_SMALL_TRACE [-DNRF_SPLIT_WAVES=4]', NRF_LIB_PATH): where wave 0 of each workgroup of the split NeRFSmall kernel spends its cycles, per layer"""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
H = W = 800
sc = S.make_hash_scene(mode="cu"); K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
r = sc["renderer"]
rp = S.lego_render_params(sc["bbox"], 64, 128, 131072, L.NRF_PREC_F16_SPLIT)
r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize()
lib = C.CDLL(os.environ["NRF_LIB_PATH"])
buf = (C.c_ulonglong * (256 * 12))()
lib.nrf_dbg_small_trace(None, 1)
NP = len(L.NRF_PROF_NAMES); ms = (C.c_double * NP)(); cnt = (C.c_int64 * NP)()
L.lib().nrf_profile_enable(1); L.lib().nrf_profile_read(ms, cnt, 1)
t0 = time.perf_counter(); r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
L.lib().nrf_profile_read(ms, cnt, 1); L.lib().nrf_profile_enable(0)
lib.nrf_dbg_small_trace(buf, 0)
mlp_ms = ms[L.NRF_PROF_NAMES.index("mlp")]
a = np.array(buf[:], dtype=np.float64).reshape(256, 12)
it = a[:, 10].sum()
names = ["sigma L0 (16 MFMA)", "sigma L1 (48)", "sigma L2 (24)", "geo split", "colour L0 (24)", "colour L1 (48)", "colour L2 (48)", "colour L3 (24)", "store + hand-over", "iteration total"]
ideal = [16 * 32, 48 * 32, 24 * 32, 0, 24 * 32, 48 * 32, 48 * 32, 24 * 32, 0, 232 * 32]
print("frame %.2f ms; iterations (wave 0 of each workgroup) %d; cycles per iteration, mean over workgroups; MFMA-pipe cycles of the section's own matrix instructions beside it" % (dt * 1e3, int(it)))
for i, n in enumerate(names):
    print("  %-22s %8.0f   (pipe %5d)" % (n, a[:, i].sum() / it, ideal[i]))
print("  kernel time %.3f ms over %d launches; stamped cycles per workgroup %.3e -> the cycle counter ran at %.2f GHz of kernel time" % (mlp_ms, cnt[L.NRF_PROF_NAMES.index("mlp")], a[:, 9].sum() / 256, a[:, 9].sum() / 256 / (mlp_ms * 1e-3) / 1e9))
