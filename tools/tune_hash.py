#!/usr/bin/env python3
"""GPU micro-bench of the hash-encode kernel variants on ray-shaped point sets (tuning aid, not a test)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfpp_amd import _lib as L, scene
from nerfpp_amd.renderer import GetRays

P = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def main():
    lib = L.lib()
    lib.nrf_dbg_hash_lm.restype = C.c_int
    sc = scene.make_hash_scene(mode="cu")
    h = sc["embedder"]._h
    K = scene.lego_K(800, 800); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = GetRays(800, 800, K, c2w, row0=360, rows=80)      # 64 000 rays
    n = o.shape[0] * o.shape[1]
    rays = torch.empty((n, 11), device="cuda")
    bb = sc["bbox"]
    L.check(lib.nrf_pack_rays(P(o), P(d), bb.ctypes.data_as(C.c_void_p), C.c_int64(n), 1, P(rays), None))
    for S, label in ((64, "coarse 64/ray uniform"), (192, "192/ray uniform")):
        t = torch.linspace(0, 1, S).cuda()
        z = torch.empty((n, S), device="cuda")
        L.check(lib.nrf_z_vals(P(rays), 11, C.c_int64(n), P(t), S, 0, P(z), None))
        pts = torch.empty((n * S, 3), device="cuda")
        L.check(lib.nrf_points(P(rays), 11, P(z), C.c_int64(n), S, P(pts), None))
        npts = n * S
        ref = torch.empty((npts, 32), device="cuda"); keep = torch.empty((npts,), device="cuda", dtype=torch.uint8)
        med, mn = timeit(lambda: L.check(lib.nrf_hash_encode(h, P(pts), C.c_int64(npts), P(ref), P(keep), None)))
        print(f"[{label}] {npts/1e6:.2f} M pts  generic row-major fp32: {med:.3f} ms (min {mn:.3f})  {npts/med/1e6:.2f} Gpts/s  alg {npts*588/med/1e6:.0f} GB/s")
        feats = torch.empty((16, npts, 2), device="cuda", dtype=torch.float16)
        for lv in range(16):
            med, mn = timeit(lambda: L.check(lib.nrf_dbg_hash_lm(h, P(pts), C.c_int64(npts), 0, lv, lv + 1, P(feats), P(keep), None)), n=5)
            print(f"      level {lv:2d} alone: {med*1e3:7.1f} us", end="" if lv % 4 != 3 else "\n")
        for variant in (0, 1, 8, 16, 24, 9, 17):
            feats.zero_()
            med, mn = timeit(lambda: L.check(lib.nrf_dbg_hash_lm(h, P(pts), C.c_int64(npts), variant, 0, -1, P(feats), P(keep), None)))
            same = bool((feats.permute(1, 0, 2).reshape(npts, 32).float() == ref).all())
            print(f"    variant {variant} (ppt={1 + (variant & 1)}, xcd={(variant >> 1) & 3}, gather={(variant >> 3) & 3}): {med:.3f} ms (min {mn:.3f})  {npts/med/1e6:.2f} Gpts/s  "
                  f"alg {npts*588/med/1e6:.0f} GB/s ({npts*588/med/1e6/8000*100:.1f}% of 8 TB/s)  bit-identical={same}")


if __name__ == "__main__":
    main()
