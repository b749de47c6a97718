"""NeRFSmall architectures inside and outside the matrix-core family (sigma net 2-3 layers, colour net 2-4, SH degree 4 / 8 = 16 / 64 direction features, hidden 64 or
not) x both hash encoders x precisions: the split render stays within split-precision distance of NRF_PREC_F32, the fp16 one within fp16 distance, both Chunk-invariant and
finite; NRF_PREC_F32 == the CPU oracle bit for bit.  usage (GPU box): python tools/scratch/arch_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R, modules as M, synth
from oracle import capi as O
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1618)          # second argument: another seed
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for case in range(cases):
    mode = ("cu", "ngp")[int(rng.integers(0, 2))]
    nl = int(rng.choice([2, 3])); nlc = int(rng.choice([2, 3, 4])); deg = int(rng.choice([4, 4, 8])) if mode == "cu" else 4
    hidden = int(rng.choice([64, 64, 64, 32])); geo = int(rng.choice([15, 15, 7]))
    msgs = []; outs = {}
    try:
        bbox = np.asarray(S.LEGO_BBOX, np.float32)
        Lv, F, T = 16, 2, 14
        table = S.synth_hash_table(Lv, T, F, 100 + case, 0.5)
        if mode == "cu":
            emb = M.CuHashEmbedder("embedder", bbox, Lv, F, T, 16, 512); emb.set_primes(np.array(S.CU_PRIMES[:3 * Lv], np.int32)); dirs = M.CuSHEncoder("embeddirs", 3, deg)
        else:
            emb = M.HashEmbedder("embedder", bbox, Lv, F, T, 16, 512); dirs = M.SHEncoder("embeddirs", 3, deg)
        emb.set_table(table)
        in_ch, in_views = Lv * F, deg * deg
        params = S.synth_linear_stack(S.small_shapes(in_ch, in_views, nl, hidden, geo, nlc, hidden), 2000 + case, 1.6, 0.0, {f"sigma_net_{nl - 1}": 12.0})
        blob = np.concatenate([a.reshape(-1) for _, a in params])
        mlp = M.NeRFSmall(nl, hidden, geo, nlc, hidden, False, 3, 64, in_ch, in_views, "model", params=blob)
        r = R.NeRFRenderer(emb, dirs, mlp)
        h, w = int(rng.integers(20, 60)), int(rng.integers(20, 60))
        K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), -30.0, 4.0)
        for prec in (L.NRF_PREC_F32, L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA):
            a = r.Render(h, w, K, S.lego_render_params(bbox, 64, 128, h * w, prec), c2w=c2w)
            b = r.Render(h, w, K, S.lego_render_params(bbox, 64, 128, max(64, h * w // 3 + 1), prec), c2w=c2w)
            ra, rb = a.Outputs.RGBMap, b.Outputs.RGBMap
            if not bool(torch.isfinite(ra).all()): msgs.append(f"precision {prec}: non-finite")
            if not torch.equal(ra, rb): msgs.append(f"precision {prec}: depends on Chunk")
            outs[prec] = ra.cpu().numpy().reshape(-1, 3)
            if prec == L.NRF_PREC_F32: rays = a.Extras["rays_flat"].cpu().numpy() if a.Extras and "rays_flat" in a.Extras else None
        p_split = S.psnr(outs[L.NRF_PREC_F16_SPLIT], outs[L.NRF_PREC_F32]); p_f16 = S.psnr(outs[L.NRF_PREC_F16_MFMA], outs[L.NRF_PREC_F32])
        if p_split < 95: msgs.append(f"split vs fp32 only {p_split:.1f} dB")
        if p_f16 < 30: msgs.append(f"fp16 vs fp32 only {p_f16:.1f} dB")
        info = f"split {p_split:.1f} dB, fp16 {p_f16:.1f} dB"
    except Exception as e:
        # a network outside the built matrix-core family (hidden != 64) is REFUSED in the two matrix-core precisions (loudly: no silent fp32 fall-back); NRF_PREC_F32 rendered it above
        if hidden != 64 and "outside the built matrix-core family" in str(e) and L.NRF_PREC_F32 in outs and np.isfinite(outs[L.NRF_PREC_F32]).all():
            info = "fp32 rendered, matrix-core precisions refused (expected: outside the built family)"
        else:
            msgs.append(f"EXCEPTION {type(e).__name__}: {str(e)[:200]}"); info = ""
    bad += bool(msgs)
    print(f"case {case:2d}: {mode} sigma {nl} colour {nlc} hidden {hidden} geo {geo} SH {deg}: {'ok ' + info if not msgs else 'FAIL ' + '; '.join(msgs)}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
