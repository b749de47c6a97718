"""Host-side cost of the weak-scaling step's shape: one whole frame in one Render call vs N row tiles of N different frames in N calls (same ray count)."""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
H = W = 800
sc = S.make_hash_scene(mode="cu"); r = sc["renderer"]
K = S.lego_K(H, W)
rp = S.lego_render_params(sc["bbox"], 64, 128, 131072, L.NRF_PREC_F16_SPLIT)
poses = [S.pose_spherical(-180.0 + 9.0 * k, -30.0, 4.0) for k in range(8)]
def timeit(f, reps=6):
    for _ in range(2): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print("one frame, one call: %.2f ms" % timeit(lambda: r.Render(H, W, K, rp, c2w=poses[0])))
for n in (2, 4, 8):
    rows = H // n
    print("%d tiles of %d rows (rank 1 of %d), %d calls: %.2f ms" % (n, rows, n, n, timeit(lambda: [r.Render(H, W, K, rp, c2w=poses[k], row0=rows, rows=rows) for k in range(n)])))
