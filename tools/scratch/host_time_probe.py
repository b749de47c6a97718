"""Host time of one frame's library call (nrf_render_rows through the Python mirror) on an IDLE queue vs back to back: is the 5 ms `host_ms_per_tile` of the bench
launch work, or the host waiting for queue space?   usage: python tools/scratch/host_time_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfpp_amd import _lib as L, scene
from benchlib.costs import H, W, NS, NI

sc = scene.make_hash_scene(mode="cu")
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
r = sc["renderer"]
for chunk in (65536, 131072):
    rp = scene.lego_render_params(sc["bbox"], NS, NI, chunk, L.NRF_PREC_F16_SPLIT)
    for lanes in (1, 2):
        L.check(L.lib().nrf_set_render_lanes(lanes))
        for _ in range(5):
            r.Render(H, W, K, rp, c2w=c2w)
        torch.cuda.synchronize()
        idle = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); r.Render(H, W, K, rp, c2w=c2w); idle.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); b2b = []
        for _ in range(20):
            t1 = time.perf_counter(); r.Render(H, W, K, rp, c2w=c2w); b2b.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / 20
        print(f"chunk {chunk} lanes {lanes}: host idle-queue {1e3*min(idle):.2f}/{1e3*sorted(idle)[5]:.2f} ms (min/median), back-to-back {1e3*sorted(b2b)[10]:.2f} ms median, frame {1e3*tot:.2f} ms", flush=True)
