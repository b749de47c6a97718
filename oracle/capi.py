"""ctypes binding of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (nerfpp_amd) must never import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "nerf_oracle.c")):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_mlp_small_param_count.restype = C.c_int64
        _LIB.orc_mlp_nerf_param_count.restype = C.c_int64
        _LIB.orc_aten_row_sum.restype = C.c_float
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


ATEN_VEC = 8  # fp32 lanes of ATen's sum kernel on any AVX2-or-newer x86 host (see nerf_oracle.c)


def linspace(start, end, steps):
    out = np.empty(steps, np.float32)
    lib().orc_linspace(C.c_float(start), C.c_float(end), C.c_int(steps), _p(out))
    return out


def get_rays(h, w, K, c2w, row0=0, rows=None):
    rows = h - row0 if rows is None else rows
    K = _f(K); c2w = _f(c2w)
    o = np.empty((rows, w, 3), np.float32); d = np.empty((rows, w, 3), np.float32)
    cone = C.c_float(0)
    lib().orc_get_rays(C.c_int(h), C.c_int(w), _p(K), _p(c2w), C.c_int(row0), C.c_int(rows), _p(o), _p(d), C.byref(cone))
    return o, d, np.float32(cone.value)


def ndc_rays(h, w, focal, near, o, d):
    o = _f(o).reshape(-1, 3); d = _f(d).reshape(-1, 3)
    oo = np.empty_like(o); od = np.empty_like(d)
    lib().orc_ndc_rays(C.c_int(h), C.c_int(w), C.c_float(focal), C.c_float(near), _p(o), _p(d), C.c_int64(o.shape[0]), _p(oo), _p(od))
    return oo, od


def aabb(o, d, bbox, near_plane=0.0):
    o = _f(o).reshape(-1, 3); d = _f(d).reshape(-1, 3); bbox = _f(bbox)
    n = o.shape[0]
    nr = np.empty(n, np.float32); fr = np.empty(n, np.float32)
    lib().orc_aabb(_p(o), _p(d), _p(bbox), C.c_int64(n), C.c_float(near_plane), _p(nr), _p(fr))
    return nr, fr


def z_vals(nears, fars, t, lindisp=False):
    nears = _f(nears).reshape(-1); fars = _f(fars).reshape(-1); t = _f(t)
    z = np.empty((nears.shape[0], t.shape[0]), np.float32)
    lib().orc_z_vals(_p(nears), _p(fars), _p(t), C.c_int64(nears.shape[0]), C.c_int(t.shape[0]), C.c_int(int(lindisp)), _p(z))
    return z


def points(o, d, z):
    o = _f(o); d = _f(d); z = _f(z)
    n, s = z.shape
    pts = np.empty((n, s, 3), np.float32)
    lib().orc_points(_p(o), _p(d), _p(z), C.c_int64(n), C.c_int(s), _p(pts))
    return pts


def pe(x, nfreq):
    x = _f(x).reshape(-1, 3)
    out = np.empty((x.shape[0], 3 + 6 * nfreq), np.float32)
    lib().orc_pe(_p(x), C.c_int64(x.shape[0]), C.c_int(nfreq), _p(out))
    return out


def sh_libtorch(dirs, degree):
    dirs = _f(dirs).reshape(-1, 3)
    out = np.empty((dirs.shape[0], degree * degree), np.float32)
    lib().orc_sh_libtorch(_p(dirs), C.c_int64(dirs.shape[0]), C.c_int(degree), _p(out))
    return out


def sh_cu(dirs, degree):
    dirs = _f(dirs).reshape(-1, 3)
    out = np.empty((dirs.shape[0], degree * degree), np.float32)
    lib().orc_sh_cu(_p(dirs), C.c_int64(dirs.shape[0]), C.c_int(degree), _p(out))
    return out


def hash_ngp_resolutions(L, base, finest):
    out = np.empty(L, np.float32)
    lib().orc_hash_ngp_resolutions(C.c_int(L), C.c_int(base), C.c_int(finest), _p(out))
    return out


def hash_ngp(x, table, bbox, L, F, log2_t, base, finest):
    x = _f(x).reshape(-1, 3); table = _f(table); bbox = _f(bbox)
    assert table.size == L * (1 << log2_t) * F
    out = np.empty((x.shape[0], L * F), np.float32); mask = np.empty(x.shape[0], np.uint8)
    lib().orc_hash_ngp(_p(x), C.c_int64(x.shape[0]), _p(table), _p(bbox), C.c_int(L), C.c_int(F), C.c_int(log2_t),
                       C.c_int(base), C.c_int(finest), _p(out), _p(mask))
    return out, mask.astype(bool)


def hash_cu_scales(L, base, finest):
    out = np.empty(L, np.float32)
    lib().orc_hash_cu_scales(C.c_int(L), C.c_int(base), C.c_int(finest), _p(out))
    return out


def f32_to_f16(a):
    a = _f(a)
    out = np.empty(a.shape, np.uint16)
    lib().orc_f32_to_f16(_p(a), _p(out), C.c_int64(a.size))
    return out


def set_cuda_fma_model(flags):
    """0 (default): CuHashEmbedder.cu's expressions as written; bit 0 / bit 1: model nvcc's FMA contraction of the blend / of scale + bias (sensitivity study only)."""
    lib().orc_set_cuda_fma_model(C.c_int(int(flags)))


def hash_cu(x, table_f16, primes, local_idx, local_size, bias, bbox, mul, L, F):
    x = _f(x).reshape(-1, 3)
    table_f16 = np.ascontiguousarray(table_f16, dtype=np.uint16)
    primes = np.ascontiguousarray(primes, dtype=np.int32); local_idx = np.ascontiguousarray(local_idx, dtype=np.int32)
    local_size = np.ascontiguousarray(local_size, dtype=np.int32)
    bias = _f(bias); bbox = _f(bbox); mul = _f(mul)
    out = np.empty((x.shape[0], L * F), np.float32); mask = np.empty(x.shape[0], np.uint8)
    lib().orc_hash_cu(_p(x), C.c_int64(x.shape[0]), _p(table_f16), _p(primes), _p(local_idx), _p(local_size), _p(bias), _p(bbox),
                      _p(mul), C.c_int(L), C.c_int(F), _p(out), _p(mask))
    return out, mask.astype(bool)


def mlp_small(params, x, in_ch, in_views, n_layers=3, hidden=64, geo=15, n_layers_c=4, hidden_c=64):
    params = _f(params); x = _f(x)
    n = lib().orc_mlp_small_param_count(C.c_int(in_ch), C.c_int(in_views), C.c_int(n_layers), C.c_int(hidden), C.c_int(geo), C.c_int(n_layers_c), C.c_int(hidden_c))
    assert params.size == n, (params.size, n)
    out = np.empty((x.shape[0], 4), np.float32)
    lib().orc_mlp_small(_p(params), _p(x), C.c_int64(x.shape[0]), C.c_int(in_ch), C.c_int(in_views), C.c_int(n_layers), C.c_int(hidden),
                        C.c_int(geo), C.c_int(n_layers_c), C.c_int(hidden_c), _p(out))
    return out


def mlp_small_pred_normal(params, x, in_ch, in_views, n_layers=3, hidden=64, geo=15, n_layers_c=4, hidden_c=64, n_layers_n=3, hidden_n=64):
    """NeRFSmallImpl::forward with the predicted-normals head (orc_mlp_small_pred_normal) -> [p, 7] = (rgb, sigma, normal)."""
    params = _f(params); x = _f(x)
    out = np.empty((x.shape[0], 7), np.float32)
    lib().orc_mlp_small_pred_normal(_p(params), _p(x), C.c_int64(x.shape[0]), C.c_int(in_ch), C.c_int(in_views), C.c_int(n_layers), C.c_int(hidden), C.c_int(geo), C.c_int(n_layers_c),
                                    C.c_int(hidden_c), C.c_int(n_layers_n), C.c_int(hidden_n), _p(out))
    return out


def mlp_nerf(params, x, d=8, w=256, in_ch=63, in_views=27, out_ch=4, skip=4, use_viewdirs=True):
    params = _f(params); x = _f(x)
    n = lib().orc_mlp_nerf_param_count(C.c_int(d), C.c_int(w), C.c_int(in_ch), C.c_int(in_views), C.c_int(out_ch), C.c_int(skip), C.c_int(int(use_viewdirs)))
    assert params.size == n, (params.size, n)
    od = 4 if use_viewdirs else out_ch
    out = np.empty((x.shape[0], od), np.float32)
    lib().orc_mlp_nerf(_p(params), _p(x), C.c_int64(x.shape[0]), C.c_int(d), C.c_int(w), C.c_int(in_ch), C.c_int(in_views), C.c_int(out_ch),
                       C.c_int(skip), C.c_int(int(use_viewdirs)), _p(out))
    return out


def mlp_nerf_backward(params, x, g_out, d=8, w=256, in_ch=63, in_views=27, out_ch=4, skip=4, use_viewdirs=True):
    """Backward of NeRFImpl::forward (orc_mlp_nerf_backward) -> (g_params [blob], g_x [p, in_ch])."""
    params = _f(params); x = _f(x); g_out = _f(g_out)
    g_params = np.zeros_like(params); g_x = np.empty((x.shape[0], in_ch), np.float32)
    lib().orc_mlp_nerf_backward(_p(params), _p(x), _p(g_out), C.c_int64(x.shape[0]), C.c_int(d), C.c_int(w), C.c_int(in_ch), C.c_int(in_views), C.c_int(out_ch), C.c_int(skip),
                                C.c_int(int(use_viewdirs)), _p(g_params), _p(g_x))
    return g_params, g_x


def lerf(params, x, in_ch=128, n_layers=2, hidden=256, geo=32, embed=768):
    params = _f(params); x = _f(x)
    out = np.empty((x.shape[0], embed + 1), np.float32)
    lib().orc_lerf(_p(params), _p(x), C.c_int64(x.shape[0]), C.c_int(in_ch), C.c_int(n_layers), C.c_int(hidden), C.c_int(geo), C.c_int(embed), _p(out))
    return out


def lerf_sigma_net(params, x, in_ch=128, n_layers=2, hidden=256, geo=32):
    """SigmaLENet(x) of LeRFImpl::forward (LeRF.cpp:86-95) -> [p, 1 + geo]: column 0 = sigma_le, columns 1.. = geo_feat_le."""
    params = _f(params); x = _f(x)
    out = np.empty((x.shape[0], 1 + geo), np.float32)
    lib().orc_lerf_sigma_net(_p(params), _p(x), C.c_int64(x.shape[0]), C.c_int(in_ch), C.c_int(n_layers), C.c_int(hidden), C.c_int(geo), _p(out))
    return out


def raw2outputs(raw, z, d, white_bkgr=False):
    raw = _f(raw); z = _f(z); d = _f(d)
    n, s, c = raw.shape
    rgb = np.empty((n, 3), np.float32); disp = np.empty(n, np.float32); acc = np.empty(n, np.float32)
    w = np.empty((n, s), np.float32); depth = np.empty(n, np.float32)
    lib().orc_raw2outputs(_p(raw), _p(z), _p(d), C.c_int64(n), C.c_int(s), C.c_int(c), C.c_int(int(white_bkgr)), _p(rgb), _p(disp), _p(acc), _p(w), _p(depth))
    return dict(rgb=rgb, disp=disp, acc=acc, weights=w, depth=depth)


def raw2weights(raw, sigma_ch, z, d):
    raw = _f(raw); z = _f(z); d = _f(d)
    n, s, c = raw.shape
    w = np.empty((n, s), np.float32); depth = np.empty(n, np.float32); disp = np.empty(n, np.float32); acc = np.empty(n, np.float32)
    lib().orc_raw2weights(_p(raw), C.c_int(c), C.c_int(sigma_ch), _p(z), _p(d), C.c_int64(n), C.c_int(s), _p(w), _p(depth), _p(disp), _p(acc))
    return dict(weights=w, depth=depth, disp=disp, acc=acc)


def render_clip_embedding(embeds, dim, w):
    embeds = _f(embeds); w = _f(w)
    n, s, stride = embeds.shape
    out = np.empty((n, dim), np.float32)
    lib().orc_render_clip_embedding(_p(embeds), C.c_int(stride), C.c_int(dim), _p(w), C.c_int64(n), C.c_int(s), _p(out))
    return out


def sample_pdf(bins, weights, u, sum_vec=ATEN_VEC):
    bins = _f(bins); weights = _f(weights); u = _f(u)
    n, nb = bins.shape
    assert weights.shape == (n, nb - 1)
    ns = u.shape[0]
    samples = np.empty((n, ns), np.float32); inds = np.empty((n, ns), np.int64); cdf = np.empty((n, nb), np.float32)
    lib().orc_sample_pdf(_p(bins), _p(weights), C.c_int64(n), C.c_int(nb), _p(u), C.c_int(ns), C.c_int(sum_vec), _p(samples), _p(inds), _p(cdf))
    return samples, inds, cdf


def z_mid(z):
    z = _f(z)
    n, s = z.shape
    out = np.empty((n, s - 1), np.float32)
    lib().orc_z_mid(_p(z), C.c_int64(n), C.c_int(s), _p(out))
    return out


def merge_sorted(z, zs):
    z = _f(z); zs = _f(zs)
    n, s = z.shape
    ns = zs.shape[1]
    out = np.empty((n, s + ns), np.float32)
    lib().orc_merge_sorted(_p(z), C.c_int(s), _p(zs), C.c_int(ns), C.c_int64(n), _p(out))
    return out


def pack_rays(o, d, bbox):
    o = _f(o).reshape(-1, 3); d = _f(d).reshape(-1, 3); bbox = _f(bbox)
    rays = np.empty((o.shape[0], 11), np.float32)
    lib().orc_pack_rays(_p(o), _p(d), _p(bbox), C.c_int64(o.shape[0]), _p(rays))
    return rays


class OrcModel(C.Structure):
    _fields_ = [
        ("family", C.c_int),
        ("bbox", C.c_void_p),
        ("n_levels", C.c_int), ("n_feat", C.c_int), ("log2_t", C.c_int), ("base", C.c_int), ("finest", C.c_int),
        ("table_f32", C.c_void_p), ("table_f16", C.c_void_p),
        ("primes", C.c_void_p), ("local_idx", C.c_void_p), ("local_size", C.c_void_p),
        ("bias", C.c_void_p), ("mul", C.c_void_p),
        ("sh_degree", C.c_int), ("pe_freqs", C.c_int), ("pe_freqs_views", C.c_int),
        ("params", C.c_void_p),
        ("n_layers", C.c_int), ("hidden", C.c_int), ("geo", C.c_int), ("n_layers_c", C.c_int), ("hidden_c", C.c_int),
        ("depth", C.c_int), ("width", C.c_int), ("skip", C.c_int),
    ]


class Model:
    """Keeps the numpy buffers alive next to the C struct."""

    def __init__(self, family, params, bbox=None, table_f32=None, table_f16=None, L=16, F=2, log2_t=19, base=16, finest=512,
                 primes=None, local_idx=None, local_size=None, bias=None, mul=None, sh_degree=4, pe_freqs=10, pe_freqs_views=4,
                 n_layers=3, hidden=64, geo=15, n_layers_c=4, hidden_c=64, depth=8, width=256, skip=4):
        self.keep = dict(
            params=_f(params), bbox=_f(bbox) if bbox is not None else None,
            table_f32=_f(table_f32) if table_f32 is not None else None,
            table_f16=np.ascontiguousarray(table_f16, np.uint16) if table_f16 is not None else None,
            primes=np.ascontiguousarray(primes, np.int32) if primes is not None else None,
            local_idx=np.ascontiguousarray(local_idx, np.int32) if local_idx is not None else None,
            local_size=np.ascontiguousarray(local_size, np.int32) if local_size is not None else None,
            bias=_f(bias) if bias is not None else None, mul=_f(mul) if mul is not None else None)
        k = self.keep
        self.c = OrcModel(family, _p(k["bbox"]), L, F, log2_t, base, finest, _p(k["table_f32"]), _p(k["table_f16"]), _p(k["primes"]),
                          _p(k["local_idx"]), _p(k["local_size"]), _p(k["bias"]), _p(k["mul"]), sh_degree, pe_freqs, pe_freqs_views,
                          _p(k["params"]), n_layers, hidden, geo, n_layers_c, hidden_c, depth, width, skip)


class OrcStoch(C.Structure):
    _fields_ = [("perturb", C.c_float), ("has_cone", C.c_int), ("cone_angle", C.c_float), ("raw_noise_std", C.c_float), ("precond_alpha", C.c_float),
                ("seed", C.c_uint64), ("ray_base", C.c_int64)] + [(k, C.c_void_p) for k in
                ("t_rand", "u_r1", "u_theta1", "noise1", "u_pdf", "precond", "u_r2", "u_theta2", "noise2")]


def render_rays(model: Model, rays, n_samples, n_importance, t_coarse, u_fine, lindisp=False, white_bkgr=True, want_intermediates=False,
                sum_vec=ATEN_VEC, stoch=None):
    """stoch: None (deterministic render path) or a dict with perturb / cone_angle (None = thin rays) / raw_noise_std / precond_alpha /
    seed / ray_base and, optionally, explicit draw arrays t_rand, u_r1, u_theta1, noise1, u_pdf, precond, u_r2, u_theta2, noise2."""
    rays = _f(rays)
    n = rays.shape[0]
    s, sf = n_samples, n_samples + n_importance
    t_coarse = _f(t_coarse); u_fine = _f(u_fine) if u_fine is not None else None
    out = dict(rgb=np.empty((n, 3), np.float32), disp=np.empty(n, np.float32), acc=np.empty(n, np.float32), depth=np.empty(n, np.float32),
               weights=np.empty((n, sf if n_importance > 0 else s), np.float32))
    inter = {}
    if want_intermediates:
        inter = dict(z_coarse=np.empty((n, s), np.float32), z_fine=np.empty((n, sf), np.float32), raw_coarse=np.empty((n, s, 4), np.float32),
                     raw_fine=np.empty((n, sf, 4), np.float32), weights_coarse=np.empty((n, s), np.float32),
                     pts_coarse=np.empty((n, s, 3), np.float32), pts_fine=np.empty((n, sf, 3), np.float32))
    st, keep = None, []
    if stoch is not None:
        cone = stoch.get("cone_angle")
        st = OrcStoch(float(stoch.get("perturb", 0.0)), int(cone is not None), float(cone or 0.0), float(stoch.get("raw_noise_std", 0.0)),
                      float(stoch.get("precond_alpha", 0.0)), int(stoch.get("seed", 0)), int(stoch.get("ray_base", 0)))
        for k in ("t_rand", "u_r1", "u_theta1", "noise1", "u_pdf", "precond", "u_r2", "u_theta2", "noise2"):
            if stoch.get(k) is not None:
                a = _f(stoch[k]); keep.append(a)
                setattr(st, k, a.ctypes.data)
    lib().orc_render_rays_stoch(C.byref(model.c), _p(rays), C.c_int64(n), C.c_int(n_samples), C.c_int(n_importance), _p(t_coarse), _p(u_fine),
                                C.c_int(int(lindisp)), C.c_int(int(white_bkgr)), C.c_int(sum_vec), C.byref(st) if st is not None else None,
                                _p(out["rgb"]), _p(out["disp"]), _p(out["acc"]), _p(out["depth"]),
                                _p(out["weights"]), _p(inter.get("z_coarse")), _p(inter.get("z_fine")), _p(inter.get("raw_coarse")),
                                _p(inter.get("raw_fine")), _p(inter.get("weights_coarse")), _p(inter.get("pts_coarse")), _p(inter.get("pts_fine")))
    out.update(inter)
    return out


def rng_uniform(seed, stream, idx0, count):
    out = np.empty(count, np.float32)
    lib().orc_rng_uniform(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(idx0), C.c_int64(count), _p(out))
    return out


def rng_normal(seed, stream, idx0, count):
    out = np.empty(count, np.float32)
    lib().orc_rng_normal(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(idx0), C.c_int64(count), _p(out))
    return out


def jitter_z(z, t_rand):
    z = _f(z); t_rand = _f(t_rand)
    out = np.empty_like(z)
    lib().orc_jitter_z(_p(z), _p(t_rand), C.c_int64(z.shape[0]), C.c_int(z.shape[1]), _p(out))
    return out


def tangent_scatter(pts, z, cone_angle, rays_d, u_r, u_theta, bbox=None):
    pts = _f(pts); z = _f(z); rays_d = _f(rays_d); u_r = _f(u_r); u_theta = _f(u_theta); bb = _f(bbox) if bbox is not None else None
    out = np.empty_like(pts)
    lib().orc_tangent_scatter(_p(pts), _p(z), C.c_float(cone_angle), _p(rays_d), _p(u_r), _p(u_theta), _p(bb), C.c_int64(z.shape[0]), C.c_int(z.shape[1]), _p(out))
    return out


def precondition(pts, noise, alpha, bbox):
    pts = _f(pts); noise = _f(noise); bb = _f(bbox)
    out = np.empty_like(pts)
    lib().orc_precondition(_p(pts), _p(noise), C.c_float(alpha), _p(bb), C.c_int64(pts.size // 3), _p(out))
    return out


def sample_pdf_rand(bins, weights, u, sum_vec=ATEN_VEC):
    bins = _f(bins); weights = _f(weights); u = _f(u)
    n, nb = bins.shape
    ns = u.shape[1]
    samples = np.empty((n, ns), np.float32); inds = np.empty((n, ns), np.int64)
    lib().orc_sample_pdf_rand(_p(bins), _p(weights), C.c_int64(n), C.c_int(nb), _p(u), C.c_int(ns), C.c_int(sum_vec), _p(samples), _p(inds))
    return samples, inds


def raw2outputs_noise(raw, z, d, noise, noise_std, white_bkgr=False):
    raw = _f(raw); z = _f(z); d = _f(d); noise = _f(noise)
    n, s, c = raw.shape
    out = dict(rgb=np.empty((n, 3), np.float32), disp=np.empty(n, np.float32), acc=np.empty(n, np.float32), weights=np.empty((n, s), np.float32),
               depth=np.empty(n, np.float32))
    lib().orc_raw2outputs_noise(_p(raw), _p(z), _p(d), C.c_int64(n), C.c_int(s), C.c_int(c), C.c_int(int(white_bkgr)), _p(noise), C.c_float(noise_std),
                                _p(out["rgb"]), _p(out["disp"]), _p(out["acc"]), _p(out["weights"]), _p(out["depth"]))
    return out


def normalize_depth(depth, near, far):
    d = _f(depth); out = np.empty_like(d)
    lib().orc_normalize_depth(_p(d), C.c_int64(d.size), C.c_float(near), C.c_float(far), _p(out))
    return out


def to_u8(x):
    x = _f(x); out = np.empty(x.shape, np.uint8)
    lib().orc_to_u8(_p(x), C.c_int64(x.size), out.ctypes.data_as(C.c_void_p))
    return out


def relevancy(embeds, positives, negatives, positive_id=0):
    """PARITY UNPINNED restatement of RuCLIP's Relevancy (LeRFRenderer.cpp:79): see nerf_oracle.c."""
    e = _f(embeds); pos = _f(positives); neg = _f(negatives)
    out = np.empty((e.shape[0], 2), np.float32)
    lib().orc_relevancy(_p(e), C.c_int64(e.shape[0]), C.c_int(e.shape[1]), _p(pos), C.c_int(pos.shape[0]), _p(neg), C.c_int(neg.shape[0]), C.c_int(positive_id), _p(out))
    return out


def colormap_jet_lut():
    lut = np.empty((256, 3), np.uint8)
    lib().orc_colormap_jet_lut(lut.ctypes.data_as(C.c_void_p))
    return lut


def relevancy_image(rel):
    rel = _f(rel)
    out = np.empty((rel.shape[0], 3), np.uint8)
    lib().orc_relevancy_image(_p(rel), C.c_int64(rel.shape[0]), C.c_int(rel.shape[1] if rel.ndim > 1 else 1), out.ctypes.data_as(C.c_void_p))
    return out


def huber_loss(pred, target):
    pred = _f(pred); target = _f(target)
    loss = C.c_float(0); mse = C.c_float(0); grad = np.empty_like(pred)
    lib().orc_huber_loss(_p(pred), _p(target), C.c_int64(pred.size), C.byref(loss), C.byref(mse), _p(grad))
    return loss.value, mse.value, grad


def huber_rows_nanmean(pred, target, delta=1.25):
    """huber_loss(pred, target, reduction none, delta).sum(-1).nanmean() (NeRFExecutor.h:970-974) -> (loss, d loss / d pred)."""
    pred = _f(pred); target = _f(target)
    n, e = pred.shape
    loss = C.c_float(0); grad = np.empty_like(pred)
    lib().orc_huber_rows_nanmean(_p(pred), _p(target), C.c_int64(n), C.c_int(e), C.c_float(delta), C.byref(loss), _p(grad))
    return loss.value, grad


def lerf_param_count(in_ch, n_layers, hidden, geo, embed):
    n, cd = 0, in_ch
    for l in range(n_layers):
        od = (1 + geo) if l == n_layers - 1 else hidden
        n += cd * od; cd = od
    cd = geo + in_ch
    for l in range(n_layers):
        od = embed if l == n_layers - 1 else hidden
        n += cd * od; cd = od
    return n


def lerf_head_backward(params, emb, keep, z, d, g_rendered, in_ch=128, n_layers=2, hidden=256, geo=32, embed=768, noise=None, noise_std=0.0):
    """Backward of the fine pass of LeRFRenderer::RenderRays downstream of the language grid (orc_lerf_head_backward) ->
    dict(g_params [blob], g_emb [n*s, in], rendered [n, E], weights [n, s])."""
    params = _f(params); emb = _f(emb); z = _f(z); d = _f(d); g_rendered = _f(g_rendered)
    n, s = z.shape
    k8 = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    g_params = np.zeros(lerf_param_count(in_ch, n_layers, hidden, geo, embed), np.float32)
    assert params.size == g_params.size, (params.size, g_params.size)
    g_emb = np.empty_like(emb); rendered = np.empty((n, embed), np.float32); weights = np.empty((n, s), np.float32)
    lib().orc_lerf_head_backward(_p(params), _p(emb), _p(k8), _p(z), _p(d), C.c_int64(n), C.c_int(s), C.c_int(in_ch), C.c_int(n_layers), C.c_int(hidden), C.c_int(geo),
                                 C.c_int(embed), _p(None if noise is None else _f(noise)), C.c_float(noise_std), _p(g_rendered), _p(g_params), _p(g_emb), _p(rendered), _p(weights))
    return dict(g_params=g_params, g_emb=g_emb, rendered=rendered, weights=weights)


def raw2outputs_backward(raw, z, d, g_rgb, white_bkgr=False):
    raw = _f(raw); z = _f(z); d = _f(d); g_rgb = _f(g_rgb)
    n, s, c = raw.shape
    out = np.empty_like(raw)
    lib().orc_raw2outputs_backward(_p(raw), _p(z), _p(d), C.c_int64(n), C.c_int(s), C.c_int(c), C.c_int(int(white_bkgr)), _p(g_rgb), _p(out))
    return out


def raw2outputs_backward_noise(raw, z, d, g_rgb, noise, noise_std, white_bkgr=False):
    raw = _f(raw); z = _f(z); d = _f(d); g_rgb = _f(g_rgb); noise = _f(noise)
    n, s, c = raw.shape
    out = np.empty_like(raw)
    lib().orc_raw2outputs_backward_noise(_p(raw), _p(z), _p(d), C.c_int64(n), C.c_int(s), C.c_int(c), C.c_int(int(white_bkgr)), _p(noise), C.c_float(noise_std),
                                         _p(g_rgb), _p(out))
    return out


def mlp_small_backward(params, x, g_out, in_ch, in_views, n_layers=3, hidden=64, geo=15, n_layers_c=4, hidden_c=64):
    params = _f(params); x = _f(x); g_out = _f(g_out)
    g_params = np.zeros_like(params); g_x = np.empty((x.shape[0], in_ch), np.float32)
    lib().orc_mlp_small_backward(_p(params), _p(x), _p(g_out), C.c_int64(x.shape[0]), C.c_int(in_ch), C.c_int(in_views), C.c_int(n_layers), C.c_int(hidden),
                                 C.c_int(geo), C.c_int(n_layers_c), C.c_int(hidden_c), _p(g_params), _p(g_x))
    return g_params, g_x


def hash_ngp_backward(x, bbox, L, F, log2_t, base, finest, g_emb):
    x = _f(x); g_emb = _f(g_emb); bb = _f(bbox)
    g_table = np.zeros((L, 1 << log2_t, F), np.float32)
    lib().orc_hash_ngp_backward(_p(x), C.c_int64(x.shape[0]), _p(bb), C.c_int(L), C.c_int(F), C.c_int(log2_t), C.c_int(base), C.c_int(finest), _p(g_emb), _p(g_table))
    return g_table


def hash_cu_backward(x, primes, local_idx, local_size, bias, bbox, mul, L, F, pool_elems, g_emb):
    x = _f(x); g_emb = _f(g_emb); bb = _f(bbox); bias = _f(bias); mul = _f(mul)
    pr = np.ascontiguousarray(primes, np.int32); li = np.ascontiguousarray(local_idx, np.int32); ls = np.ascontiguousarray(local_size, np.int32)
    out = np.zeros(pool_elems, np.float32)
    lib().orc_hash_cu_backward(_p(x), C.c_int64(x.shape[0]), pr.ctypes.data_as(C.c_void_p), li.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p),
                               _p(bias), _p(bb), _p(mul), C.c_int(L), C.c_int(F), C.c_int64(pool_elems), _p(g_emb), _p(out))
    return out


def tv_loss(table_level, log2_t, min_vertex, cube, weight=1.0, want_grad=True):
    t = _f(table_level); F = t.shape[-1]
    mv = np.ascontiguousarray(min_vertex, np.int32)
    loss = C.c_float(0); grad = np.zeros_like(t) if want_grad else None
    lib().orc_tv_loss(_p(t), C.c_int(log2_t), C.c_int(F), mv.ctypes.data_as(C.c_void_p), C.c_int(cube), C.c_float(weight), C.byref(loss), _p(grad))
    return loss.value, grad


def adam_step(p, g, m, v, lr, t, b1=0.9, b2=0.99, eps=1e-15):
    """in place on p, m, v (float32 arrays)"""
    assert p.dtype == np.float32 and m.dtype == np.float32 and v.dtype == np.float32
    g = _f(g)
    lib().orc_adam_step(_p(p), _p(g), _p(m), _p(v), C.c_int64(p.size), C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int(t))


def ray_batch(K, c2w, rand_h, rand_w):
    K = _f(K).reshape(-1); M = _f(np.asarray(c2w)[:3, :4]).reshape(-1)
    rh = np.ascontiguousarray(rand_h, np.int64); rw = np.ascontiguousarray(rand_w, np.int64)
    n = rh.size
    o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32); cone = C.c_float(0)
    lib().orc_ray_batch(_p(K), _p(M), rh.ctypes.data_as(C.c_void_p), rw.ctypes.data_as(C.c_void_p), C.c_int64(n), _p(o), _p(d), C.byref(cone))
    return o, d, cone.value


def gather_pixels(image, rand_h, rand_w):
    img = _f(image); h, w, c = img.shape
    rh = np.ascontiguousarray(rand_h, np.int64); rw = np.ascontiguousarray(rand_w, np.int64)
    out = np.empty((rh.size, c), np.float32)
    lib().orc_gather_pixels(_p(img), C.c_int(h), C.c_int(w), C.c_int(c), rh.ctypes.data_as(C.c_void_p), rw.ctypes.data_as(C.c_void_p), C.c_int64(rh.size), _p(out))
    return out


def precrop_bounds(h, w, it, precrop_iters, precrop_frac):
    out = (C.c_int * 4)()
    lib().orc_precrop_bounds(C.c_int(h), C.c_int(w), C.c_int(it), C.c_int(precrop_iters), C.c_float(precrop_frac), out)
    return tuple(out)


def rand_pixels(seed, it, bounds, n):
    rh = np.empty(n, np.int64); rw = np.empty(n, np.int64)
    lib().orc_rand_pixels(C.c_uint64(seed), C.c_int64(it), C.c_int(bounds[0]), C.c_int(bounds[1]), C.c_int(bounds[2]), C.c_int(bounds[3]), C.c_int64(n),
                          rh.ctypes.data_as(C.c_void_p), rw.ctypes.data_as(C.c_void_p))
    return rh, rw


def num_threads():
    return lib().orc_num_threads()
