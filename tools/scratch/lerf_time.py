"""LeRF render pass timing per precision: whole 800x800 frame, per-kernel ms (MLP slot = the fused LeRF passes)"""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
sc = S.make_lerf_scene(); r = sc["renderer"]
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
ref = None
for prec in (L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA):
    r.set_precision(prec)
    r.Render(800, 800, K, p, c2w=c2w, row0=0, rows=41); torch.cuda.synchronize()
    lib = L.lib(); lib.nrf_profile_enable(1); ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
    t0 = time.perf_counter(); res = r.Render(800, 800, K, p, c2w=c2w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
    emb = res.Outputs.RenderedLangEmbedding
    print(r.precision_name, "s/frame %.3f" % dt, "units/s %.3e" % (640000 * 256 / dt), {n: round(ms[i], 1) for i, n in enumerate(L.NRF_PROF_NAMES)}, float(emb.abs().mean()), flush=True)
    if ref is None: ref = emb
    else:
        cos = (emb * ref).sum(1)
        print("f16 vs split: cos min %.6f median %.8f" % (float(cos.min()), float(cos.median())))
