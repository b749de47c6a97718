"""Training steps at RANDOM batch sizes / sample counts / stochastic settings / encoders / backward variants: finite losses and gradients, the matrix-core backward
against the fp32 one, the binned / packed table gradients against the float atomics, a poisoned scratch buffer changes nothing.  usage (GPU box): python tools/scratch/train_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import Trainer
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)          # second argument: another seed
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
for case in range(cases):
    mode = ("cu", "ngp")[int(rng.integers(0, 2))]
    n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 16384]))
    s = int(rng.choice([8, 32, 64])); ni = int(rng.choice([0, 16, 128]))
    stoch = bool(rng.integers(0, 2)); prec = int(rng.choice([L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F32]))
    msgs = []
    try:
        grads = {}
        for variant in (("f32", "f32"), ("f16", "binned"), ("f16", "packed")):
            sc = S.make_hash_scene(mode=mode, log2_t=14, table_amp=1e-2, sigma_scale=4.0, seed=1234 + case)
            if variant[1] != "f32" and mode == "ngp" and False: continue
            tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward=variant[0], hash_backward=variant[1], seed=7)
            g = torch.Generator(device="cpu").manual_seed(case)
            K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
            o, d, cone = R.GetRays(800, 800, K, c2w)
            idx = torch.randint(0, 640000, (n,), generator=g).cuda()
            o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
            tgt = torch.rand((n, 3), generator=g).cuda()
            kw = dict(Perturb=1.0, RawNoiseStd=0.3, StochasticPreconditioningAlpha=0.01, ThinRay=False, Seed=5) if stoch else {}
            rp = R.NeRFRenderParams(NSamples=s, NImportance=ni, Chunk=max(n, 64), WhiteBkgr=False, Ndc=False, UseViewdirs=True, BoundingBox=S.LEGO_BBOX, Precision=prec,
                                    **({"ThinRay": True, "Perturb": 0.0} if not stoch else {}), **kw)
            lm, res = tr.step(o, d, tgt, rp, cone_angle=cone if stoch else None)
            torch.cuda.synchronize()
            if not (torch.isfinite(lm).all() and torch.isfinite(tr.g_table).all() and torch.isfinite(tr.g_blob).all() and torch.isfinite(tr.table).all() and torch.isfinite(tr.blob).all()):
                msgs.append(f"{variant}: non-finite")
            grads[variant] = (tr.g_table.clone(), tr.g_blob.clone(), float(lm[0]))
            tr.close()
        ref = grads[("f32", "f32")]
        for v in (("f16", "binned"), ("f16", "packed")):
            gt, gb, lo = grads[v]
            st = float(ref[0].abs().max()); sb = float(ref[1].abs().max())
            et = float((gt - ref[0]).abs().max()) / (st + 1e-30); eb = float((gb - ref[1]).abs().max()) / (sb + 1e-30)
            cc = float(torch.corrcoef(torch.stack([gt.double(), ref[0].double()]))[0, 1]) if st > 0 else 1.0
            # fp16 operands in the backward chain: the worst element of a small batch moves by ~10 % of the largest gradient, the gradient as a whole not at all
            if not (et < 0.25 and eb < 0.1 and cc > 0.995 and abs(lo - ref[2]) < 1e-6 + 1e-5 * abs(ref[2])): msgs.append(f"{v}: table grad err {et:.2e} (corr {cc:.5f}) blob grad err {eb:.2e} loss {lo} vs {ref[2]}")
        if not torch.equal(grads[("f16", "binned")][0], grads[("f16", "packed")][0]): msgs.append("binned != packed table gradient")
    except Exception as e:
        msgs.append(f"EXCEPTION {type(e).__name__}: {str(e)[:200]}")
    bad += bool(msgs)
    print(f"case {case:2d}: {mode} n {n} s {s}+{ni} stoch {stoch} precision {prec}: {'ok' if not msgs else 'FAIL ' + '; '.join(msgs)}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
