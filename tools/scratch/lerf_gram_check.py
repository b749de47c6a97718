"""The head's backward with the last layer in its Gram form against the layer-wise form on the autograd goldens' inputs: the two must agree to fp32 noise and must NOT be
the same bits (else the Gram path was not taken).  usage (GPU box): python tools/scratch/lerf_gram_check.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from nerfpp_amd import _lib as L, modules as M, train as T
import test_gpu_parity as TG
lib = L.lib(); lib.nrf_dbg_lerf_train_gram.restype = C.c_int
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for tag in ("train_lerf", "train_lerf_l3", "train_lerf_main"):
    c, blob, g, where = TG._lerf_golden_case(tag)
    lerf = M.LeRF(c["geo"], c["n_layers"], c["hidden"], c["embed"], c["in_ch"], "lang_model", params=blob)
    out = {}
    for form in (1, 0):
        lib.nrf_dbg_lerf_train_gram(form)
        r = T.LeRFHeadBackward(lerf, dev(g["emb"]), dev(g["keep"].astype(np.uint8)), dev(g["z"]), dev(g["d"]), dev(g["grad_rendered"]))
        out[form] = {k: r[k].cpu().numpy() for k in ("g_params", "g_emb", "rendered", "weights")}
    lib.nrf_dbg_lerf_train_gram(1)
    line = []
    for k in ("g_params", "g_emb", "rendered", "weights"):
        a, b = out[1][k], out[0][k]
        line.append("%s max|d| %.2e of max %.2e%s" % (k, float(np.abs(a - b).max()), float(np.abs(b).max()), " (same bits)" if np.array_equal(a, b) else ""))
    print("%-16s n %d s %d hidden %d embed %d: " % (tag, c["n"], c["s"], c["hidden"], c["embed"]) + "; ".join(line), flush=True)
