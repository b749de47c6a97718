"""What one rank of the strong-scaling step does at N = 1, 2, 4, 8: its 800/N-row tile of ONE frame, back to back (20 steps, one sync at the end): GPU time per
tile against frame/N (the scaling the kernels themselves allow, before the collective), and the host time spent inside Render per tile."""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
H = W = 800
sc = S.make_hash_scene(mode="cu"); r = sc["renderer"]
K = S.lego_K(H, W)
rp = S.lego_render_params(sc["bbox"], 64, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 65536, L.NRF_PREC_F16_SPLIT)
pose = S.pose_spherical(-180.0, -30.0, 4.0)
base = None
for n in (1, 2, 4, 8):
    rows = H // n
    for rank in sorted({0, n // 2, n - 1}):
        f = lambda: r.Render(H, W, K, rp, c2w=pose, row0=rank * rows, rows=rows)
        for _ in range(3): f()
        torch.cuda.synchronize()
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(20):
            th = time.perf_counter(); f(); host += time.perf_counter() - th
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        if n == 1: base = dt
        print("N=%d rank %d: %3d rows  %.3f ms per tile (frame/N = %.3f, kernel-side speedup %.2fx of %d), host %.3f ms per tile" % (n, rank, rows, dt, base / n, base / dt, n, host / 20 * 1e3), flush=True)
