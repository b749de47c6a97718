"""Experiment: do the hash-encode (vector-memory bound) and MLP (MFMA bound) kernels of independent ray tiles overlap when issued on two HIP streams?"""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
from nerfpp_amd.renderer import NeRFRenderer
H = W = 800
for precname, prec in (("f16x3", L.NRF_PREC_F16_SPLIT),):
    sc = S.make_hash_scene(mode="cu")
    K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    for nstreams, chunk in ((1, 131072), (2, 131072), (2, 32768), (3, 32768), (4, 32768), (4, 16384)):
        rs = [NeRFRenderer(sc["embedder"], sc["embeddirs"], sc["mlp"]) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        rp = S.lego_render_params(sc["bbox"], 64, 128, chunk, prec)
        rows = H // nstreams
        def frame():
            outs = []
            for i, (r, st) in enumerate(zip(rs, streams)):
                with torch.cuda.stream(st):
                    outs.append(r.Render(H, W, K, rp, c2w=c2w, row0=i * rows, rows=rows if i < nstreams - 1 else H - i * rows).Outputs.RGBMap)
            return outs
        for _ in range(2): frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): o = frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(precname, "streams", nstreams, "chunk", chunk, "ms/frame %.2f" % (dt * 1e3), "units/s %.3e" % (H * W * 256 / dt), flush=True)
