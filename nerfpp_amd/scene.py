"""Synthetic Blender-Lego-shaped scenes for the bench, the smoke test and the parity tests.

No dataset, checkpoint or network exists in the build environment, so cameras follow the reference's Blender loader
arithmetic and every weight / hash-table entry comes from the closed form in include/nrf_synth.h (nerfpp_amd/synth.py).
"""
import math

import numpy as np
import torch

from . import _lib as L
from .modules import CuHashEmbedder, CuSHEncoder, Embedder, HashEmbedder, NeRF, NeRFSmall, SHEncoder
from .renderer import NeRFRenderer, NeRFRenderParams
from .synth import synth_sym

LEGO_CAMERA_ANGLE_X = 0.6911112
LEGO_BBOX = np.array([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], np.float32)

# 96 primes in [2^28, 2^30): the per-level hash primes of the CuHashEmbedder mode.  The reference draws them from
# torch::randint with trial-division rejection (CuHashEmbedder.cpp:28-49); the bench pins this fixed list instead
# (first 3*L are used), generated once from nrf_synth_u32(424242, i) % (2^30 - 2^28) + 2^28 with the same rejection.
CU_PRIMES = [
    386165369, 328816889, 363285193, 664599349, 426166501, 1065949217, 529205959, 791427041, 599257903, 715819693, 293172031, 591017617,
    565386769, 276368591, 537961741, 638270141, 577357117, 473811929, 392382953, 948774581, 852084049, 396726817, 981405811, 327115433,
    349186699, 586339771, 939500743, 732051877, 500085371, 698724029, 788736313, 765365063, 946438387, 405400997, 414515113, 455409239,
    512980241, 993398677, 583548209, 299861017, 819272401, 486247219, 352939841, 711418867, 553615919, 838921697, 579932123, 868889447,
    826009297, 865765763, 334549937, 1070983609, 403035407, 712660387, 764017031, 818273411, 537802541, 368437877, 314756753, 739725739,
    836358977, 418021693, 697933267, 641827211, 1062956077, 560260577, 561752813, 530036851, 605910061, 592304023, 276082627, 752779273,
    845547881, 294434671, 780297709, 752175139, 794346613, 524138519, 808184921, 839281607, 381116261, 472437887, 657056797, 315261917,
    904019279, 418034443, 977216651, 922945631, 909171539, 1058047787, 776331091, 906552223, 301687271, 889392611, 354966947, 834522797,
]


def lego_K(h, w):
    """load_blender.h:161,190-192: focal = .5*W/tan(.5*camera_angle_x); K = [[f,0,W/2],[0,f,H/2],[0,0,1]]."""
    focal = np.float32(0.5 * w / math.tan(0.5 * LEGO_CAMERA_ANGLE_X))
    return np.array([[focal, 0, 0.5 * w], [0, focal, 0.5 * h], [0, 0, 1]], np.float32)


def pose_spherical(theta, phi, radius):
    """load_blender.h:43-57 (fp32 cosf/sinf, fp32 4x4 products) -> c2w [3,4]."""
    f = np.float32
    pi = f(math.acos(-1.0))
    th, ph = f(theta) / f(180.0) * pi, f(phi) / f(180.0) * pi
    t = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], f)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], f)
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], f)
    fl = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], f)
    c2w = (fl @ (rt @ (rp @ t))).astype(f)
    return np.ascontiguousarray(c2w[:3, :4])


def lego_render_params(bbox=LEGO_BBOX, n_samples=64, n_importance=128, chunk=32768, precision=L.NRF_PREC_F32, white_bkgr=True, **kw):
    """The deterministic test-time parameter set (FillRenderParams, NeRFExecutor.h:379-415, + ThinRay)."""
    args = dict(NSamples=n_samples, NImportance=n_importance, Chunk=chunk, ReturnRaw=False, LinDisp=False, Perturb=0.0,
                WhiteBkgr=white_bkgr, RawNoiseStd=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=False, ThinRay=True,
                BoundingBox=np.asarray(bbox, np.float32), Precision=precision)
    args.update(kw)
    return NeRFRenderParams(**args)


def _amp(gain, fan_in, fan_out):
    return np.float32(gain) * np.sqrt(np.float32(6.0) / np.float32(fan_in + fan_out))


def synth_linear_stack(shapes, base_seed, gain, bias_amp=0.0, scales=None):
    """[(name, out, in, has_bias)] -> [(name.weight, W), (name.bias, b)...] with the reference driver's fill rule:
    tensor k gets seed base_seed + 1000*k, amplitude gain*sqrt(6/(in+out)) (weights) or bias_amp (biases)."""
    out, k = [], 0
    scales = scales or {}
    for name, o, i, has_bias in shapes:
        amp = _amp(gain, i, o)
        for key, sc in scales.items():
            if key in name + ".weight":
                amp = np.float32(amp * np.float32(sc))
        out.append((name + ".weight", synth_sym(base_seed + 1000 * k, (o, i), amp))); k += 1
        if has_bias:
            out.append((name + ".bias", synth_sym(base_seed + 1000 * k, (o,), np.float32(bias_amp)))); k += 1
    return out


def small_shapes(input_ch=32, input_ch_views=16, num_layers=3, hidden=64, geo=15, num_layers_color=4, hidden_color=64):
    sh = []
    for l in range(num_layers):
        sh.append((f"model_sigma_net_{l}", (1 + geo) if l == num_layers - 1 else hidden, input_ch if l == 0 else hidden, False))
    for l in range(num_layers_color):
        sh.append((f"model_color_net_{l}", 3 if l == num_layers_color - 1 else hidden_color, (input_ch_views + geo) if l == 0 else hidden_color, False))
    return sh


def nerf_shapes(d=8, w=256, input_ch=63, input_ch_views=27, skip=4):
    sh = [("model_pts_linears_0", w, input_ch, True)]
    for i in range(d - 1):
        sh.append((f"model_pts_linears_{i + 1}", w, (w + input_ch) if i == skip else w, True))
    sh += [("model_views_linears_0", w // 2, input_ch_views + w, True), ("model_feature_linear", w, w, True),
           ("model_alpha_linear", 1, w, True), ("model_rgb_linear", 3, w // 2, True)]
    return sh


def synth_hash_table(n_levels, log2_t, n_feat, base_seed=5000, amp=0.5):
    """Level l gets seed base_seed + 1000*l (the reference driver's order: embeddings_0..L-1)."""
    return np.concatenate([synth_sym(base_seed + 1000 * l, ((1 << log2_t) * n_feat,), np.float32(amp)) for l in range(n_levels)])


def make_hash_scene(mode="cu", n_levels=16, n_feat=2, log2_t=19, base=16, finest=512, sh_degree=4, num_layers_color=4, seed=5000,
                    table_amp=0.5, sigma_scale=30.0, bbox=LEGO_BBOX, num_layers=3):
    """HashNeRF (BASELINE config 3/4): hash grid + SH + NeRFSmall.  mode 'cu' = CuHashEmbedder + CuSHEncoder (the named
    plugin), 'ngp' = HashEmbedder + SHEncoder (the LibTorch CPU twin the oracle/_ref pins)."""
    table = synth_hash_table(n_levels, log2_t, n_feat, seed, table_amp)
    if mode == "cu":
        emb = CuHashEmbedder("embedder", bbox, n_levels, n_feat, log2_t, base, finest)
        primes = np.array(CU_PRIMES[:3 * n_levels], np.int32)
        emb.set_primes(primes)
        dirs = CuSHEncoder("embeddirs", 3, sh_degree)
    else:
        emb = HashEmbedder("embedder", bbox, n_levels, n_feat, log2_t, base, finest)
        primes = None
        dirs = SHEncoder("embeddirs", 3, sh_degree)
    emb.set_table(table)
    in_ch, in_views = n_levels * n_feat, sh_degree * sh_degree
    params = synth_linear_stack(small_shapes(in_ch, in_views, num_layers, 64, 15, num_layers_color, 64), seed + 1000, 1.6, 0.0,
                                {f"sigma_net_{num_layers - 1}": sigma_scale})
    blob = np.concatenate([a.reshape(-1) for _, a in params])
    mlp = NeRFSmall(num_layers, 64, 15, num_layers_color, 64, False, 3, 64, in_ch, in_views, "model", params=blob)
    return dict(renderer=NeRFRenderer(emb, dirs, mlp), embedder=emb, embeddirs=dirs, mlp=mlp, table=table, mlp_blob=blob, primes=primes,
                bbox=np.asarray(bbox, np.float32), mode=mode, cfg=dict(n_levels=n_levels, n_feat=n_feat, log2_t=log2_t, base=base, finest=finest,
                                                                      sh_degree=sh_degree, num_layers_color=num_layers_color))


def make_classic_scene(multires=10, multires_views=4, seed=7000, alpha_scale=40.0, bbox=LEGO_BBOX):
    """Classic NeRF (BASELINE config 1/2): PE(10) + PE(4) + NeRF 8x256 with view directions."""
    emb = Embedder("embedder", multires); dirs = Embedder("embeddirs", multires_views)
    in_ch, in_views = emb.GetOutputDims(), dirs.GetOutputDims()
    params = synth_linear_stack(nerf_shapes(8, 256, in_ch, in_views, 4), seed, 1.4, 0.1, {"alpha_linear.weight": alpha_scale})
    blob = np.concatenate([a.reshape(-1) for _, a in params])
    mlp = NeRF(8, 256, in_ch, in_views, 5, (4,), True, "model", params=blob)
    return dict(renderer=NeRFRenderer(emb, dirs, mlp), embedder=emb, embeddirs=dirs, mlp=mlp, mlp_blob=blob, bbox=np.asarray(bbox, np.float32))


def make_lerf_scene(n_levels=16, n_feat=8, log2_t=19, base=16, finest=1024, num_layers=2, hidden=256, geo=32, embed=768, seed=311, table_amp=0.5,
                    sigma_scale=20.0, bbox=LEGO_BBOX):
    """LeRF render pass (BASELINE config 5, the dimensions of main.cpp:203-213): CuHashEmbedder L16 F8 T2^19 16..1024 + LeRF 2x256 -> 768."""
    from .modules import LeRF
    from .renderer import LeRFRenderer
    from . import synth
    emb = CuHashEmbedder("lang_embedder", bbox, n_levels, n_feat, log2_t, base, finest)
    primes = np.array(CU_PRIMES[:3 * n_levels], np.int32)
    emb.set_primes(primes)
    table = synth.synth_sym(seed, (n_levels * (1 << log2_t) * n_feat,), np.float32(table_amp))
    emb.set_table(table)
    in_ch = n_levels * n_feat
    shapes = []
    for l in range(num_layers):
        shapes.append((f"sigma_le_net_{l}", (1 + geo) if l == num_layers - 1 else hidden, in_ch if l == 0 else hidden, False))
    for l in range(num_layers):
        shapes.append((f"le_net_{l}", embed if l == num_layers - 1 else hidden, (geo + in_ch) if l == 0 else hidden, False))
    params = synth_linear_stack(shapes, seed + 1000, 1.6, 0.0, {f"sigma_le_net_{num_layers - 1}": sigma_scale})
    blob = np.concatenate([a.reshape(-1) for _, a in params])
    lerf = LeRF(geo, num_layers, hidden, embed, in_ch, "lang_model", params=blob)
    return dict(renderer=LeRFRenderer(emb, lerf), embedder=emb, lerf=lerf, table=table, blob=blob, primes=primes, bbox=np.asarray(bbox, np.float32))


def psnr(a, b):
    """-10*log10(mse) (NeRFExecutor.h:893)."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)
