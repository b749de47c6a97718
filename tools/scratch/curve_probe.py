import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import test_gpu_parity as T
from nerfpp_amd import _lib, modules, renderer, scene
from nerfpp_amd.synth import load_manifest
api = type("Api", (), dict(L=_lib, M=modules, R=renderer, S=scene))
man = load_manifest(os.path.join(os.path.dirname(T.__file__), "golden", "manifest.txt"))
for mb, prec in (("f32", _lib.NRF_PREC_F32), ("f16", _lib.NRF_PREC_F16_SPLIT), ("f32", _lib.NRF_PREC_F16_SPLIT)):
    g, loss, mse, sk = T._train_curve_run(api, man, mb, prec)
    rel = np.abs(loss - g["loss"]) / g["loss"]
    print(mb, prec, "max", rel.max(), "median", np.median(rel), "argmax", rel.argmax(), "skipped", sk)
    print(np.array2string(rel, precision=2, max_line_width=250))
    print("ref ", np.array2string(g["loss"][::7], precision=5)); print("ours", np.array2string(loss[::7], precision=5))
