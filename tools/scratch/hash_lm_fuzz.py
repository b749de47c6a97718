"""The renderer's level-major hash encode (baked dense pyramid + hashed levels, hash_fast.hip) against the generic encoder (encode.hip) and the CPU oracle at RANDOM grid
configurations: table sizes 2^10..2^19, base / finest resolutions, per-level biases on / off (CuHashEmbedder.cpp:28-49), dense-image budgets from 0 to everything (so the
split between baked and hashed levels falls anywhere), points inside, on and outside the box, F = 2 (the fast path) and F = 8 (the LeRF encoder).
usage (GPU box): python tools/scratch/hash_lm_fuzz.py [cases]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, modules as M, synth
from oracle import capi as O
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 60221023)          # second argument: another seed
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = L.lib()
P = lambda t: C.c_void_p(t.data_ptr())
bad = 0
for case in range(cases):
    mode = ("cu", "ngp")[int(rng.integers(0, 2))]
    F = 2 if (mode == "ngp" or rng.integers(0, 3)) else 8
    Lv = 16; T = int(rng.choice([10, 12, 15, 17, 19])); base = int(rng.choice([2, 4, 16, 32])); fin = int(rng.choice([64, 300, 512, 1024, 2048]))
    lo = np.array(rng.uniform(-2, -0.5, 3), np.float32); hi = lo + np.array(rng.uniform(1.0, 4.0, 3), np.float32)
    bbox = np.concatenate([lo, hi]).astype(np.float32)
    p = int(rng.choice([1, 63, 64, 65, 4097, 50000]))
    x = rng.uniform(lo - 0.3, hi + 0.3, (p, 3)).astype(np.float32)
    x[rng.integers(0, p, max(1, p // 50))] = hi                      # exactly on the upper corner
    x[rng.integers(0, p, max(1, p // 50))] = lo
    budget = int(rng.choice([0, 1 << 20, 64 << 20, 1 << 30, 8 << 30]))
    msgs = []
    try:
        table = synth.synth_sym(int(rng.integers(1, 10000)), (Lv * (1 << T) * F,), np.float32(0.5))
        if mode == "cu":
            e = M.CuHashEmbedder("e", bbox, Lv, F, T, base, fin)
            primes = np.array(S.CU_PRIMES[:3 * Lv], np.int32)
            biases = rng.uniform(0, 1, (Lv, 3)).astype(np.float32) if rng.integers(0, 2) else None
            e.set_primes(primes, biases)
        else:
            e = M.HashEmbedder("e", bbox, Lv, F, T, base, fin); biases = None
        e.set_dense_budget(budget)
        e.set_table(table)
        xd = torch.from_numpy(x).cuda()
        gen, keep = e.forward(xd)                                     # generic encoder [p, L*F] fp32 (CU: fp16-rounded values)
        if mode == "cu":
            ls = ((1 << T) >> 4) << 4
            ref, rk = O.hash_cu(x, O.f32_to_f16(table), primes, np.arange(Lv, dtype=np.int32) * ls, np.full(Lv, ls, np.int32), biases if biases is not None else np.zeros((Lv, 3), np.float32),
                                bbox, O.hash_cu_scales(Lv, base, fin), Lv, F)
            if not np.array_equal(gen.cpu().numpy(), ref): msgs.append("generic encoder != oracle")
            feats = torch.empty((Lv, p, F), device="cuda", dtype=torch.float16); k8 = torch.empty((p,), device="cuda", dtype=torch.uint8)
            L.check(lib.nrf_hash_encode_lm_f16(e._h, P(xd), C.c_int64(p), P(feats), P(k8), None))
            lm = feats.float().permute(1, 0, 2).reshape(p, Lv * F)
            if not torch.equal(lm, gen): msgs.append(f"level-major fast path != generic encoder ({int((lm != gen).sum())} of {lm.numel()} values)")
            if not torch.equal(k8.bool(), keep): msgs.append("keep mask differs")
        else:
            ref, rk = O.hash_ngp(x, table, bbox, Lv, F, T, base, fin)
            if not np.array_equal(gen.cpu().numpy(), ref): msgs.append("generic encoder != oracle")
            # the fast path of this encoder is reached through the renderer; compare the split render's raw with the stage-wise evaluation on a tiny frame
    except Exception as ex:
        msgs.append(f"EXCEPTION {type(ex).__name__}: {str(ex)[:200]}")
    bad += bool(msgs)
    print(f"case {case:2d}: {mode} F {F} T {T} res {base}..{fin} bias {biases is not None} budget {budget >> 20} MB p {p} dense levels {getattr(e, 'dense_levels', '?') if 'e' in dir() else '?'}: {'ok' if not msgs else 'FAIL ' + '; '.join(msgs)}", flush=True)
    del e
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
