"""Some boxes of the pool render the default frame (2 lanes x 131 072-ray chunks) at 45-130 ms instead of 22 (docs/history/profiles/round4/r4d_lane_sweep_first_box.log, r4E_*): this
probe, run first thing on a fresh box, times the frame at several chunk sizes in ONE process, with the per-kernel HIP-event times and the host time inside Render, so that a
slow box tells what is slow (every kernel? the host?) and whether a smaller chunk escapes it.   usage (GPU box): python tools/scratch/slow_mode_hunt.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene as S
t_start = time.perf_counter()
sc = S.make_hash_scene(mode="cu"); r = sc["renderer"]
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
lib = L.lib(); n = len(L.NRF_PROF_NAMES)
print(f"scene ready after {time.perf_counter() - t_start:.1f} s; free / total GB {torch.cuda.mem_get_info()[0] / 2**30:.1f} / {torch.cuda.mem_get_info()[1] / 2**30:.1f}", flush=True)
for rep in range(2):
    for chunk in (131072, 65536, 98304, 131072, 65536):
        rp = S.lego_render_params(sc["bbox"], 64, 128, chunk, L.NRF_PREC_F16_SPLIT)
        r.Render(800, 800, K, rp, c2w=c2w); torch.cuda.synchronize()
        host = 0.0; t0 = time.perf_counter()
        for _ in range(5):
            th = time.perf_counter(); r.Render(800, 800, K, rp, c2w=c2w); host += time.perf_counter() - th
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
        lib.nrf_profile_enable(1); lib.nrf_profile_read(ms, cnt, 1)
        r.Render(800, 800, K, rp, c2w=c2w); torch.cuda.synchronize()
        lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
        k = {nm: round(ms[i], 2) for i, nm in enumerate(L.NRF_PROF_NAMES) if ms[i] > 0}
        print(f"{'SLOW ' if dt > 30 else ''}chunk {chunk:7d}: {dt:7.2f} ms / frame, host inside Render {host / 5 * 1e3:6.2f} ms; one frame's kernels (HIP events, two lanes share the chip): {k}", flush=True)
