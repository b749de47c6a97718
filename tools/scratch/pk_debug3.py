"""does the baked hash encode give different features while matrix-core kernels run on another stream?"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, _lib as L
from nerfpp_amd.modules import _ptr
sc = S.make_hash_scene(mode="cu"); e = sc["embedder"]
n = 4_000_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
pts = (torch.rand((n, 3), device="cuda", generator=g) * 3.0 - 1.5).contiguous()
ref = torch.empty((16, n, 2), device="cuda", dtype=torch.float16); k = torch.empty((n,), device="cuda", dtype=torch.uint8)
L.check(L.lib().nrf_hash_encode_lm_f16(e._h, _ptr(pts), C.c_int64(n), _ptr(ref), _ptr(k), None)); torch.cuda.synchronize()
xin = torch.randn((2_000_000, 48), device="cuda") * 0.1
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = []
for mode in ("alone", "beside the split-precision MLP", "beside the exact fp32 sigma pass (render)"):
    outs = []
    if mode == "beside the split-precision MLP":
        with torch.cuda.stream(sa):
            for _ in range(40): sc["mlp"].forward(xin, L.NRF_PREC_F16_SPLIT)
    if mode.startswith("beside the exact"):
        K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
        L.check(L.lib().nrf_set_render_lanes(1))
        rp = S.lego_render_params(sc["bbox"], chunk=131072, precision=L.NRF_PREC_F16_SPLIT)
        with torch.cuda.stream(sa):
            for _ in range(3): sc["renderer"].Render(800, 800, K, rp, c2w=c2w)
    with torch.cuda.stream(sb):
        for _ in range(12):
            x = torch.empty((16, n, 2), device="cuda", dtype=torch.float16)
            L.check(L.lib().nrf_hash_encode_lm_f16(e._h, _ptr(pts), C.c_int64(n), _ptr(x), _ptr(k), C.c_void_p(sb.cuda_stream)))
            outs.append(x)
    torch.cuda.synchronize()
    print(mode, "-> differing features per repetition:", [int((o != ref).sum()) for o in outs])
