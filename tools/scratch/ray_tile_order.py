"""Does the ORDER OF THE RAYS of a chunk matter to the hash encode?  Waves stay ray-major (64 consecutive samples of one ray: coalesced feature stores, what the
renderer feeds today); what changes is which rays follow each other in the launch: row-major scanlines (800 rays between vertical neighbours) against T x T pixel
tiles (vertical neighbours T rays apart: their lines are still in the L2 instead of the Infinity Cache).  Same kernel (nrf_dbg_hash_lm), same points, same bits."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
H = W = 800; ROWS = 80
sc = S.make_hash_scene(mode="cu"); K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
rp = S.lego_render_params(sc["bbox"], 64, 128, 65536, L.NRF_PREC_F16_SPLIT, KeepIntermediates="depths")
res = sc["renderer"].Render(H, W, K, rp, c2w=c2w, row0=360, rows=ROWS)
rays = res.Extras["rays_flat"]; zf = res.Extras["z_fine"]
n, s = zf.shape
lib = L.lib(); h = sc["embedder"]._h


def run(x, name, base=None):
    x = x.reshape(-1, 3).contiguous(); p = x.shape[0]
    feats = torch.empty((16, p, 2), device="cuda", dtype=torch.float16); keep = torch.empty((p,), device="cuda", dtype=torch.uint8)
    call = lambda: L.check(lib.nrf_dbg_hash_lm(h, C.c_void_p(x.data_ptr()), C.c_int64(p), 0, 0, -1, C.c_void_p(feats.data_ptr()), C.c_void_p(keep.data_ptr()), None))
    for _ in range(2): call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print("%-28s %.3f ms for %.1f M points (%.2f G points/s)%s  checksum %.6e" % (name, best, p / 1e6, p / best / 1e6,
          "" if base is None else "  x%.3f of row-major" % (best / base), float(feats.float().sum())), flush=True)
    return best


def tiled(pts, ty, tx):        # pts [ROWS, W, s, 3] -> rays in ty x tx pixel tiles, tiles row-major, rays row-major inside a tile
    return pts.reshape(ROWS // ty, ty, W // tx, tx, pts.shape[2], 3).permute(0, 2, 1, 3, 4, 5).contiguous()


for label, z in (("fine (192 sorted depths)", zf), ("coarse (64 uniform depths)", zf[:, ::3].contiguous() if False else res.Extras.get("z_coarse", None))):
    if z is None:
        near, far = rays[:, 6:7], rays[:, 7:8]
        t = torch.linspace(0, 1, 64, device="cuda")[None]
        z = near * (1 - t) + far * t
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(ROWS, W, z.shape[1], 3)
    print("--", label)
    b = run(pts, "row-major")
    for ty, tx in ((8, 8), (16, 16), (4, 4), (8, 32), (16, 4), (80, 1), (80, 4)):
        run(tiled(pts, ty, tx), "tiles %dx%d" % (ty, tx), b)
    b2 = run(pts, "row-major (again)")
