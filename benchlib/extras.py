"""Secondary measurements of bench.py at N = 1 (not part of `value`) and the parity checks outside the timed region."""
import argparse
import json
import os
import subprocess
import time

import numpy as np

from .costs import *      # noqa: F401,F403
from .roofline import mfma_roofline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed_frames(L, render, frames, warm=2, lanes_hook=None):
    """warm untimed + `frames` timed calls of render() -> (seconds per frame, per-kernel HIP-event ms per frame and launches per frame)."""
    import ctypes as C
    import torch
    assert not L.lib().nrf_profile_is_enabled(), "frame times are taken with the per-kernel event bracketing off"
    for _ in range(warm):
        render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    # the per-kernel times come from a second, single-lane pass (nrf_set_render_lanes(1): a kernel has the GPU to itself, its launch time is its own); the frame time
    # above is the default two-lane Chunk loop's
    n = len(L.NRF_PROF_NAMES)
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
    prev = L.lib().nrf_get_render_lanes()        # what bench.py measured to be the faster lane count on this box (the frame time above ran with it)
    single = prev != 1
    if single:
        L.lib().nrf_set_render_lanes(1)
        if lanes_hook:
            lanes_hook(1)                        # a host whose own Chunk loop has lanes (LeRFRenderer)
    frames = max(2, min(frames, 4))
    render()
    torch.cuda.synchronize()
    L.lib().nrf_profile_enable(1)
    L.lib().nrf_profile_read(ms, cnt, 1)
    try:
        for _ in range(frames):
            out = render()
        torch.cuda.synchronize()
        L.lib().nrf_profile_read(ms, cnt, 1)
    finally:
        L.lib().nrf_profile_enable(0)
        if single:
            L.lib().nrf_set_render_lanes(prev)
            if lanes_hook:
                lanes_hook(prev)
    return dt, {nm: dict(ms_per_frame=ms[i] / frames, launches_per_frame=cnt[i] / frames) for i, nm in enumerate(L.NRF_PROF_NAMES)}, out


def secondary_measurements(args, scene, L, K, c2w, sc_main, steps=10):
    """Extra timings at N = 1 (not part of `value`): the other matrix-core precision of this workload and the other BASELINE workloads -- each with its own
    per-kernel HIP-event times, the roofline of its dominant kernel (algorithmic flops of the network as written x the evaluations the kernel executed) and its
    render-vs-oracle quality on a 256-ray sample."""
    import torch
    out = []
    # (workload, precision, coarse pass); classic split precision is timed in both coarse modes: "exact" (NRF_COARSE_AUTO: density branch in exact fp32 on the
    # matrix cores, the fp32 path's sample set bit for bit) and "full" (NRF_COARSE_FULL: whole network in split arithmetic, outputs reused: 99.9 % of pixels within 1e-4)
    todo = [("hash", "f16" if args.precision != "f16" else "f16x3", None), ("classic", "f16x3", "exact"), ("classic", "f16x3", "full"), ("classic", "f16", None)] if args.workload == "hash" else \
           [("classic", "f16x3", "full"), ("classic", "f16" if args.precision != "f16" else "f16x3", None), ("hash", "f16x3", None)]
    scenes = {args.workload: sc_main}
    for wl, pname, coarse in todo:
        try:
            if wl not in scenes:
                scenes[wl] = scene.make_hash_scene(mode=args.hash_mode) if wl == "hash" else scene.make_classic_scene()
            sc = scenes[wl]
            prec = {"f16": L.NRF_PREC_F16_MFMA, "f16x3": L.NRF_PREC_F16_SPLIT, "f32": L.NRF_PREC_F32}[pname]
            rp = scene.lego_render_params(sc["bbox"], NS, NI, 65536 if wl == "hash" else 8192, prec)
            if coarse == "full":
                rp.CoarseMode = L.NRF_COARSE_FULL
            n_fr = steps if not (wl == "classic" and pname == "f16x3") else max(3, steps // 2)
            dt, kms, _ = timed_frames(L, lambda: sc["renderer"].Render(H, W, K, rp, c2w=c2w), n_fr)
            a2 = argparse.Namespace(**{**vars(args), "workload": wl, "precision": pname, "coarse_full": coarse == "full"})
            ex_hash, ex_mlp, ex_sigma = executed_per_ray(wl, pname, args.hash_mode, coarse_full=(coarse != "exact") if wl == "classic" else False)
            rec = dict(workload="hashnerf_lego800_64+128" if wl == "hash" else "classic_nerf_lego800_64+128", baseline_config=3 if wl == "hash" else 2,
                       precision=pname, value=H * W * UNITS_PER_RAY / dt, unit="ray-samples/s", ms_per_step=dt * 1e3, steps=n_fr, kernel_ms=kms,
                       **({"coarse_pass": "density branch in exact fp32 on the matrix cores + colour branch on the exact h8 (sigma_nerf_f32.hip): the fp32 path's sample set, outputs reused by the fine pass" if coarse == "exact"
                           else "whole network in the timed arithmetic, outputs reused by the fine pass (NRF_COARSE_FULL)"} if coarse else {}),
                       executed_evaluations_per_ray=dict(hash_encode=ex_hash, fused_mlp=ex_mlp, sigma_only=ex_sigma, colour_net_only=colour_only_per_ray(wl, pname)))
            mk = kms["mlp"]
            if wl == "classic":
                rec["roofline"] = mfma_roofline("mlp_nerf" + ("_split" if pname == "f16x3" else ""), H * W * ex_mlp, NERF_FLOP_PER_UNIT, mk["ms_per_frame"] * 1e-3,
                                                mk["launches_per_frame"], issued_flop_per_unit=(3.0 if pname == "f16x3" else 1.0) * 1058 * 32768 / 32,
                                                note="algorithmic 1 186 816 flop of NeRFImpl::forward as written x the network evaluations the kernel executed "
                                                     "(the fine pass's 64 coarse depths take the coarse pass's outputs); issued: 1 058 matrix instructions per 32 points"
                                                     + (" x 3 products (hi + lo operand pairs)" if pname == "f16x3" else ""))
                sk = kms["sigma"]
                if sk["launches_per_frame"]:
                    rec["roofline"]["sigma_exact"] = mfma_roofline("sigma_nerf_f32 (coarse pass: density branch in exact fp32, v_mfma_f32_32x32x2_f32; + the colour branch in split fp16, 2 % of its matrix time)", H * W * ex_sigma, NERF_SIGMA_FLOP_PER_UNIT,
                                                                   sk["ms_per_frame"] * 1e-3, sk["launches_per_frame"], peak=F32_PEAK)
            else:
                rec["roofline"] = mfma_roofline("mlp_small", H * W * ex_mlp, SMALL_FLOP_PER_UNIT, mk["ms_per_frame"] * 1e-3, mk["launches_per_frame"],
                                                issued_flop_per_unit=SMALL_MFMA_FLOP_PER_UNIT.get(pname))
                ck = kms["mlp_colour"]
                if ck["launches_per_frame"]:
                    rec["roofline"]["colour_only"] = mfma_roofline("mlp_small, colour net alone", H * W * colour_only_per_ray(wl, pname), SMALL_COLOUR_FLOP_PER_UNIT, ck["ms_per_frame"] * 1e-3,
                                                                   ck["launches_per_frame"], issued_flop_per_unit=SMALL_COLOUR_MFMA_FLOP_PER_UNIT)
                hk = kms["hash"]
                rec["roofline"]["hash"] = dict(bound="hbm", kernel="hash_encode", unit="GB/s", peak=HBM_PEAK / 1e9,
                                               achieved=H * W * ex_hash * HASH_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12) / 1e9,
                                               frac=None,           # no counter pass for this secondary line: the fraction of the HBM peak is a counter figure (benchlib/roofline.py)
                                               algorithmic_over_hbm_peak=H * W * ex_hash * HASH_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12) / HBM_PEAK,
                                               gather_frac_of_cache_ceiling=H * W * ex_hash * HASH_GATHER_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12) / GATHER_PEAK)
            rec["psnr_vs_oracle_db"] = quality_check(sc, sc["renderer"], rp, K, c2w, a2)
            out.append(rec)
        except Exception as e:
            out.append(dict(workload=wl, precision=pname, error=str(e)))
    if args.workload == "hash" and args.hash_mode == "cu":
        # the same configuration on the LibTorch HashEmbedder + SHEncoder (the encoders whose reference implementation runs on the CPU and
        # pins the oracle) -- fp32 table, hi + lo fp16 feature planes
        try:
            sc = scene.make_hash_scene(mode="ngp")
            rp = scene.lego_render_params(sc["bbox"], NS, NI, 65536, L.NRF_PREC_F16_SPLIT)
            dt, kms, _ = timed_frames(L, lambda: sc["renderer"].Render(H, W, K, rp, c2w=c2w), steps)
            a2 = argparse.Namespace(**{**vars(args), "workload": "hash", "precision": "f16x3", "hash_mode": "ngp"})
            out.append(dict(workload="hashnerf_lego800_64+128", baseline_config=3, encoder="HashEmbedder + SHEncoder (LibTorch twin)", precision="f16x3",
                            value=H * W * UNITS_PER_RAY / dt, unit="ray-samples/s", ms_per_step=dt * 1e3, steps=steps, kernel_ms=kms,
                            psnr_vs_oracle_db=quality_check(sc, sc["renderer"], rp, K, c2w, a2)))
            del sc
            torch.cuda.empty_cache()
        except Exception as e:
            out.append(dict(workload="hashnerf (HashEmbedder twin)", error=str(e)))
    try:
        out.append(train_step_measurement(args, scene, L))
    except Exception as e:
        out.append(dict(workload="hashnerf_train_step", error=str(e)))
    try:
        cpu_ref = next((r.get("cpu_reference") for r in out if isinstance(r, dict) and r.get("workload") == "hashnerf_train_step"), None)
        out.append(train_run_measurement(scene, L, cpu_reference=cpu_ref))
    except Exception as e:
        out.append(dict(workload="hashnerf_train_run", error=str(e)))
    try:
        out.append(_with_train_gemm(L, lambda: lerf_train_step_measurement(scene, L)))
    except Exception as e:
        out.append(dict(workload="lerf_train_step", error=str(e)))
    try:
        out.append(_with_train_gemm(L, lambda: classic_train_step_measurement(scene, L)))
    except Exception as e:
        out.append(dict(workload="classic_train_step", error=str(e)))
    for lp in (L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA):
        try:
            out.append(lerf_measurement(scene, L, K, c2w, lp))
        except Exception as e:
            out.append(dict(workload="lerf_lego800_64+128", error=str(e)))
    out.extend(dropin_measurements())
    return out


DROPIN_RUNS = (("frame_hash", ()), ("frame_classic", ()), ("frame_lerf", ()), ("train_hash", ()), ("train_hash", ("0", "0", "hipadam")), ("train_classic", ()), ("train_lerf", ()))


def dropin_measurements(runs=DROPIN_RUNS, timeout=300):
    """What a C++ host of the reference gets: include/nerfpp_torch.h's adapters behind the reference's own virtual Render() and inside the statements of NeRFExecutor::Train's
    loop body (NeRFExecutor.h:862-995: zero_grad, Render on the ray batch, huber_loss, backward, Adam::step), timed on that host's clock by the prebuilt harness
    oracle/_ref/adapter_check bench (oracle/ref/adapter_bench.cpp) -- a child process, after and outside every timed region of this file.  Same scenes as the Python-mirror
    lines above (frame: 800x800, 64+128; steps: 16 384 rays for hash / LeRF, 4 096 for the classic model), so `dropin_*` sits next to its mirror twin."""
    exe = os.path.join(ROOT, "oracle", "_ref", "adapter_check")
    out = []
    if not os.path.exists(exe):
        return [dict(workload="dropin", error="oracle/_ref/adapter_check not built (needs /root/reference at build time)")]
    for what, extra in runs:
        name = "dropin_" + what + ("_" + extra[2] if len(extra) > 2 else "")
        try:
            r = subprocess.run([exe, "bench", what, *extra], capture_output=True, text=True, timeout=timeout)
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            if "error" in rec:
                raise RuntimeError(rec["error"])
            ms = rec.get("ms_per_frame", rec.get("ms_per_step"))
            out.append(dict(workload=name, precision="f16x3", value=rec["value"], unit=rec["unit"], ms_per_step=ms, host="C++ / LibTorch (adapter_check bench)", detail=rec))
        except Exception as e:
            out.append(dict(workload=name, error=str(e)[:200]))
    return out


def lerf_oracle_check(sc, res, nrays=256):
    """`nrays` rays of the LeRF frame end to end through the CPU oracle's fp32 stage path (CuHash F = 8 encode -> LeRFImpl::forward -> RawToLEOutputs weights ->
    SamplePDF -> fine pass -> RenderCLIPEmbedding): sample set, weights and rendered embedding of the GPU pass against it.  Checker use of oracle/ only."""
    import torch
    from oracle import capi as O
    from nerfpp_amd import scene
    acc = res.Outputs.AccMapLE.cpu().numpy()
    hit = np.nonzero(acc > 1e-2)[0]
    idx = hit[::max(1, hit.size // nrays)][:nrays]
    rays = res.Extras["rays_flat"].cpu().numpy()[idx]
    Lv, F, T = 16, 8, 19
    ls = ((1 << T) >> 4) << 4
    tab16 = O.f32_to_f16(sc["table"])
    mul = O.hash_cu_scales(Lv, 16, 1024)

    def net(pts):
        e_, keep = O.hash_cu(pts.reshape(-1, 3), tab16, sc["primes"], np.arange(Lv, dtype=np.int32) * ls, np.full(Lv, ls, np.int32), np.zeros((Lv, 3), np.float32), sc["bbox"], mul, Lv, F)
        o = O.lerf(sc["blob"], e_)
        o[~keep, -1] = 0
        return o.reshape(pts.shape[0], pts.shape[1], -1)
    zc = O.z_vals(rays[:, 6], rays[:, 7], O.linspace(0, 1, NS))
    wc = O.raw2weights(net(O.points(rays[:, :3], rays[:, 3:6], zc)), 768, zc, rays[:, 3:6])["weights"]
    samples, _, _ = O.sample_pdf(O.z_mid(zc), wc[:, 1:-1], O.linspace(0, 1, NI))
    zf = O.merge_sorted(zc, samples)
    rawf = net(O.points(rays[:, :3], rays[:, 3:6], zf))
    fin = O.raw2weights(rawf, 768, zf, rays[:, 3:6])
    ref = O.render_clip_embedding(rawf, 768, fin["weights"])
    zg = res.Extras["z_fine"].cpu().numpy()[idx]
    wg = res.Outputs.WeightsLE.cpu().numpy()[idx]
    eg = res.Outputs.RenderedLangEmbedding.cpu().numpy()[idx]
    cos = (eg * ref).sum(1)
    same = (zg == zf).all(1)
    rel_err = None
    if res.Outputs.Relevancy is not None and sc["renderer"].LerfPositives is not None:
        rel_err = float(np.abs(res.Outputs.Relevancy.cpu().numpy()[idx] - O.relevancy(eg, sc["renderer"].LerfPositives, sc["renderer"].LerfNegatives)).max())
    return dict(relevancy_max_abs_err_vs_oracle_on_gpu_embeddings=rel_err, rays=int(idx.size), embedding_max_abs_err=float(np.abs(eg - ref).max()), embedding_rms_err=float(np.sqrt(((eg - ref).astype(np.float64) ** 2).mean())),
                fine_sample_set_bit_identical_rays=float(same.mean()), weights_max_abs_err_over_max=float(np.abs(wg - fin["weights"])[same].max() / fin["weights"].max()) if same.any() else None,
                embedding_cos_min=float(cos.min()), embedding_cos_min_same_samples=float(cos[same].min()) if same.any() else None, embedding_cos_median=float(np.median(cos)),
                against="CPU oracle, fp32 stage path end to end (its own coarse pass and fine sample set)")


def lerf_measurement(scene, L, K, c2w, precision, repeats=10):
    """BASELINE config 5: the LeRF language-embedding render pass (CuHashEmbedder L16 F8 T2^19 16..1024 + LeRF 2x256 -> 768, main.cpp:203-213)
    on the WHOLE 800x800 frame, 64+128 samples: warm-up, then `repeats` timed frames with per-kernel HIP-event times, rooflines and an oracle check of 256 rays."""
    import torch
    from nerfpp_amd import renderer as R
    sc = scene.make_lerf_scene()
    p = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True,
                           BoundingBox=sc["bbox"])
    r = sc["renderer"]
    r.set_precision(precision)
    # the pass ends where its users look: with prompts set, the frame includes Relevancy (LeRFRenderer.cpp:79).  Synthetic unit phrase embeddings: one positive, three canonical negatives
    rng = np.random.RandomState(79)
    pos = rng.randn(1, 768).astype(np.float32); pos /= np.linalg.norm(pos)
    neg = rng.randn(3, 768).astype(np.float32); neg /= np.linalg.norm(neg, axis=1, keepdims=True)
    r.SetLeRFPrompts(pos, neg)
    r.keep_intermediates = False           # the timed frames write what LeRFRenderer::Render returns; the oracle check below renders once more with the depth sets kept
    lanes0 = r.lanes                       # LeRF's own default is ONE lane: the frame time and the per-kernel pass below both run that way
    dt, kms, _ = timed_frames(L, lambda: r.Render(H, W, K, p, c2w=c2w), repeats, warm=1, lanes_hook=lambda k: setattr(r, "lanes", min(k, lanes0)))
    r.keep_intermediates = True
    res = r.Render(H, W, K, p, c2w=c2w)
    n = H * W
    emb = res.Outputs.RenderedLangEmbedding
    hit = res.Outputs.AccMapLE > 1e-2
    nrm = emb[hit].norm(dim=1)
    split = r.precision_name == "f16x3"
    exact = bool(split and r._exact_coarse_on())
    rec = dict(workload="lerf_lego800_64+128", baseline_config=5, rays=n, value=n * UNITS_PER_RAY / dt, unit="ray-samples/s", s_per_frame=dt, frames_timed=repeats, kernel_ms=kms,
               lanes=int(r.lanes), fused_matrix_core_path=bool(r.fused), finite=bool(torch.isfinite(emb).all()), single_library_call=bool(r._single_call_ok(p)),
               relevancy="rendered in the pass (1 positive, 3 negatives)" if res.Outputs.Relevancy is not None else None,
               rays_with_language_density=int(hit.sum()), embedding_norm_min_max=[float(nrm.min()), float(nrm.max())] if int(hit.sum()) else None,
               level_major_features=bool(getattr(r, "level_major", False)), precision=getattr(r, "precision_name", "f16"),
               coarse_pass="sigma_le in exact fp32 on the matrix cores (sigma_lerf_f32.hip): the fp32 path's fine sample set" if exact else "the timed arithmetic",
               executed_evaluations_per_ray=dict(hash_encode=NS + NI, density_net=NS + NI, embedding_net=NS + NI,
                                                 note="every sample point is encoded once and its density net evaluated once (the fine pass's 64 coarse depths reuse the coarse pass's columns)"),
               arithmetic=("split-f16 MFMA (hi + lo operand pairs, three products, fp32 accumulate: fp32-grade)" if split else "fp16 MFMA (fp32 accumulate)") +
                          " LeRF head fused with the render pass; CuHash F=8 features level-major fp16 (256 B per sample point), "
                          "read by the kernels as operand fragments; embedding norm via the Gram matrix of the bias-free output layer, which is applied once per ray to the weighted sum of its inputs")
    # rooflines: algorithmic flops of LeRFImpl::forward as written (557 568 per sample, the 256 -> 768 layer per SAMPLE) x the 192 evaluations per ray the passes execute
    mk, sk, hk = kms["mlp"], kms["sigma"], kms["hash"]
    units = n * (NS + NI)
    t_mlp = (mk["ms_per_frame"] + sk["ms_per_frame"]) * 1e-3
    issued = None
    if split:
        new_pts, all_pts = n * NI, n * (NS + NI)
        issued_f16 = ((new_pts if exact else all_pts) * LERF_SPLIT_MFMA_SIGMA + all_pts * LERF_SPLIT_MFMA_EMBED) * 32768 / 32 + n * 24 * 16 * 3 * 32768 / 32
        issued = issued_f16 / units
    rec["roofline"] = mfma_roofline("lerf passes (density net + embedding net + per-ray output layer" + ("; coarse density net: exact-fp32 kernel, in `sigma_exact`" if exact else "") + ")",
                                    units, LERF_FLOP_PER_UNIT, t_mlp, mk["launches_per_frame"] + sk["launches_per_frame"], issued_flop_per_unit=issued,
                                    note="achieved / frac price the ALGORITHMIC 557 568 flop per sample of LeRFImpl::forward as written over the time of ALL LeRF network kernels; the kernels "
                                         "execute far fewer (Gram-matrix norm, output layer once per ray): mfma_issued_frac is the fp16 matrix pipe's own share")
    if exact and sk["launches_per_frame"]:
        rec["roofline"]["sigma_exact"] = mfma_roofline("lerf_sigma_f32 (coarse pass: density net in exact fp32, v_mfma_f32_32x32x2_f32)", n * NS, LERF_SIGMA_FLOP_PER_UNIT, sk["ms_per_frame"] * 1e-3,
                                                       sk["launches_per_frame"], peak=F32_PEAK)
    if hk["launches_per_frame"]:
        b = units * LERF_HASH_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12)
        rec["roofline"]["hash"] = dict(bound="hbm", kernel="hash_encode F=8 (k_hash_cu, level-major fp16 out)", unit="GB/s", achieved=b / 1e9, peak=HBM_PEAK / 1e9, frac=None, algorithmic_over_hbm_peak=b / HBM_PEAK,
                                       frac_of_infinity_cache_gather_rate=b / GATHER_PEAK, bytes_per_unit=LERF_HASH_BYTES_PER_UNIT, units_per_frame=units,
                                       note="the 134 MB hashed table is Infinity-Cache resident: the gather path's ceiling for such tables is 8.6 TB/s (MI355X_MICROARCH.md)")
    try:
        rec["oracle_check"] = lerf_oracle_check(sc, res)
    except Exception as e:
        rec["oracle_check"] = f"unavailable: {e}"
    return rec


def _with_train_gemm(L, fn):
    """fn() under the library's default arithmetic of the training paths' layer products (by family: f16x3 split-precision matrix-core products for the classic and the
    LeRF networks: the record's headline) and with fp32 products (rocBLAS sgemm) for comparison."""
    prev = L.lib().nrf_get_train_gemm()
    try:
        L.check(L.lib().nrf_set_train_gemm(-1))
        rec = fn()
        rec["train_gemm"] = ("f16x3: hi + lo fp16 of power-of-two scaled operands, three matrix-core products per accumulator, bias / ReLU / ReLU-mask fused "
                             "(gemm_bf16x3.hip; the library's default for this family); weight-gradient products: see `weight_gradients`")
        L.check(L.lib().nrf_set_train_gemm(0))
        r32 = fn()
        rec["fp32_products"] = dict(ms_per_step=r32["ms_per_step"], loss_first_last=r32["loss_first_last"], what="rocBLAS sgemm + separate bias / ReLU / mask passes (nrf_set_train_gemm(0))")
    finally:
        L.check(L.lib().nrf_set_train_gemm(prev))
    return rec


def _layer_products_text(L):
    """what the training paths' layer products run as under the library's current nrf_get_train_gemm() setting"""
    mode = int(L.lib().nrf_get_train_gemm())
    if mode in (-1, 2):
        return ("f16x3 split-precision matrix-core products (hi + lo fp16 of power-of-two scaled rows; forward and back-propagation), bf16x3 weight gradients "
                "(gemm_bf16x3.hip)" + (": the library's default for this family" if mode == -1 else ""))
    if mode == 1:
        return "bf16x3 split-precision matrix-core products (gemm_bf16x3.hip)"
    return "rocBLAS sgemm (fp32 matrix cores)" if L.lib().nrf_fp32_gemm_available() else "hand-written FMA kernels"


def lerf_train_step_measurement(scene, L, n_rand=16384, steps=2):
    """SURVEY section 8f row N1, LeRF branch (NeRFExecutor.h:955-985): LeRFRenderer->Render on the ray batch (the fused pass) -> lang_loss -> backward into the LeRF head and
    the F = 8 language grid (ONE library call, fp32 layer kernels on the recomputed forward) -> Adam, at main.cpp:203-213 sizes and N_rand = 32*32*16 rays (main.cpp:232)."""
    import torch
    from nerfpp_amd import renderer as R
    from nerfpp_amd.train import LeRFTrainer
    sc = scene.make_lerf_scene()
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    tgt = torch.nn.functional.normalize(torch.randn((n_rand, 768), device="cuda"), dim=-1)
    p = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=False, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
    tr = LeRFTrainer(sc["renderer"], sc["table"], sc["blob"], learning_rate=5e-4)
    try:
        l0, _ = tr.step(o, d, tgt, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = [float(l0.item())]
        for _ in range(steps):
            l, _ = tr.step(o, d, tgt, p)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        losses.append(float(l.item()))
    finally:
        tr.close()
    return dict(workload="lerf_train_step", baseline_config=5, rays_per_step=n_rand, samples="64+128", ms_per_step=dt * 1e3, rays_per_s=n_rand / dt, value=n_rand * UNITS_PER_RAY / dt,
                unit="ray-samples/s", steps=steps, loss_first_last=losses, layer_products=_layer_products_text(L),
                backward_workspace_bytes=int(tr._ws.numel()) if getattr(tr, "_ws", None) is not None else None,
                arithmetic="render: split-f16 MFMA fused pass; backward: the head's forward recomputed and differentiated with fp32-grade layer products (see layer_products; "
                           "the 256 -> 768 layer, normalize and RenderCLIPEmbedding in their Gram form: nothing 768-wide per sample), language-grid gradient by float atomics "
                           "after the ray-coherent pre-sum; Adam fp32; the head's weight images rebuilt on the device at the parameter upload")


def classic_train_step_measurement(scene, L, n_rand=4096, steps=3):
    """N1 on the classic configuration (Embedder(10) / Embedder(4) / NeRFImpl 8x256, a legal TNeRF of NeRFExecutor::Train): render, huber, backward through RawToOutputs and
    NeRFImpl (fp32 layer products: rocBLAS on the fp32 matrix cores where present), Adam -- 4 096 rays per step (a quarter of main.cpp:232's batch: the step is GEMM-bound and linear)."""
    import torch
    from nerfpp_amd import renderer as R
    from nerfpp_amd.train import Trainer
    sc = scene.make_classic_scene()
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    tgt = torch.rand((n_rand, 3), device="cuda")
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], None, sc["mlp_blob"], learning_rate=5e-4)
    rp = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX,
                            Precision=L.NRF_PREC_F16_SPLIT)
    l0, _ = tr.step(o, d, tgt, rp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        l, _ = tr.step(o, d, tgt, rp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dict(workload="classic_train_step", baseline_config=2, rays_per_step=n_rand, samples="64+128", ms_per_step=dt * 1e3, rays_per_s=n_rand / dt, value=n_rand * UNITS_PER_RAY / dt,
                unit="ray-samples/s", steps=steps, loss_first_last=[float(l0[0]), float(l[0])], layer_products=_layer_products_text(L),
                backward_workspace_bytes=int(tr._ws.numel()) if getattr(tr, "_ws", None) is not None else None)


def train_step_measurement(args, scene, L, n_rand=16384, steps=5, mlp_backward="f16"):
    """one optimisation step of NeRFExecutor::Train (render the ray batch, huber loss, backward of the fine pass,
    Adam) on the HashNeRF configuration, N_rand = 32*32*16 rays per step as in the reference's main.cpp:232, next to the reference's own
    LibTorch CPU step (oracle/_ref/ref_driver bench_train) on a bounded ray batch."""
    import torch
    from nerfpp_amd import renderer as R
    from nerfpp_amd.train import Trainer
    sc = scene.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    tgt = torch.rand((n_rand, 3), device="cuda")
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward=mlp_backward, hash_backward="binned" if mlp_backward == "f16" else "f32")
    rp = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True,
                            BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    for _ in range(2):
        tr.step(o, d, tgt, rp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for _ in range(steps):
        lm, _ = tr.step(o, d, tgt, rp)
        losses.append(lm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rec = dict(workload="hashnerf_train_step", rays_per_step=n_rand, samples="64+128", ms_per_step=dt * 1e3, rays_per_s=n_rand / dt,
               value=n_rand * UNITS_PER_RAY / dt, unit="ray-samples/s", steps=steps, loss_first_last=[float(losses[0][0]), float(losses[-1][0])],
               arithmetic="render: split-f16 MFMA; NeRFSmall backward: " + ("one fused matrix-core kernel, fp16 operands / fp32 accumulation / device-side loss scale" if mlp_backward == "f16"
                                                                            else "fp32 layer-wise kernels") +
                          "; hash backward: ray-coherent fp32 pre-sum, then " + ("fixed-point records binned by table range and summed in LDS (no atomics to memory; equals the packed-atomic path bit for bit)" if mlp_backward == "f16" else "one float atomic per feature") +
                          "; Adam fp32")
    if mlp_backward == "f16":
        try:
            r32 = train_step_measurement(argparse.Namespace(**{**vars(args), "no_cpu_baseline": True}), scene, L, n_rand, steps, "f32")
            rec["fp32_backward"] = dict(ms_per_step=r32["ms_per_step"], rays_per_s=r32["rays_per_s"], loss_first_last=r32["loss_first_last"])
        except Exception as e:
            rec["fp32_backward"] = f"unavailable: {e}"
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv) and not args.no_cpu_baseline:
        try:
            outp = subprocess.run([drv, "bench_train", "1024", str(NS), str(NI), "4096", "1"], capture_output=True, text=True, timeout=600)
            r = json.loads(outp.stdout.strip().splitlines()[-1])
            rec["cpu_reference"] = dict(rays_per_s=r["rays_per_s"], value=r["units_per_s"], unit="ray-samples/s", cores=os.cpu_count() or r["threads"], threads=r["threads"], kind="reference",
                                        sample=f"{r['rays']} rays per step, LibTorch CPU HashEmbedder+SHEncoder+NeRFSmall forward+backward+Adam, {r['seconds']:.1f} s per step")
        except Exception as e:
            rec["cpu_reference"] = f"unavailable: {e}"
    return rec


def train_run_measurement(scene, L, iters=400, n_rand=16384, views=8, hw=400, cpu_reference=None):
    """An END-TO-END training run as a measured workload: main.cpp's schedule -- 16 384 rays per iteration, 64 + 192 samples (main.cpp:189, 231-232), Adam
    lr 1e-2 with the learning rate the reference's loop really has in force (constant: its decay statement updates a copy, golden train_curve), the density-noise schedule of
    FillRenderParams (RawNoiseStd = max(0, 1 - i / (NIters / 8)), NeRFExecutor.h:411) -- of a freshly initialised student (table U 1e-4, CuHashEmbedder.cpp:24) on `views`
    teacher-rendered hw x hw views of the bench scene.  Every iteration runs the ray-batch producer (NeRFDataset::get_batch: random pixels, rays, target gather; dataset.py)
    and one Trainer.step (split-precision render, fused fp16 backward, binned scatter, Adam); the clock runs over all of it.  Quality: PSNR of a held-out view against the
    teacher's render, before and after."""
    import torch
    from nerfpp_amd import renderer as R
    from nerfpp_amd import dataset as D
    from nerfpp_amd.train import Trainer
    teacher = scene.make_hash_scene(mode="cu")
    student = scene.make_hash_scene(mode="cu", seed=777, table_amp=1e-4, sigma_scale=1.0)
    K = scene.lego_K(hw, hw)
    rp_t = scene.lego_render_params(teacher["bbox"], NS, NI, 65536, L.NRF_PREC_F16_SPLIT, white_bkgr=False)
    vs = []
    for v in range(views + 1):                                   # the last one is held out
        c2w = scene.pose_spherical(-180.0 + 360.0 * v / (views + 1), -30.0, 4.0)
        img = teacher["renderer"].Render(hw, hw, K, rp_t, c2w=c2w).Outputs.RGBMap.reshape(hw, hw, 3).clone()
        vs.append(D.View(hw, hw, K, c2w, img))
    held = vs.pop()
    ds = D.NeRFDataset(vs, n_rand, precorp_iters=0, seed=11)

    def psnr_held():
        rp_e = scene.lego_render_params(student["bbox"], NS, 192, 65536, L.NRF_PREC_F16_SPLIT, white_bkgr=False)
        x = student["renderer"].Render(hw, hw, K, rp_e, c2w=held.Pose).Outputs.RGBMap.reshape(hw, hw, 3)
        mse = float(((x - held.Image).double() ** 2).mean())
        return -10.0 * float(np.log10(max(mse, 1e-30)))
    rp = R.NeRFRenderParams(NSamples=64, NImportance=192, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX,
                            Precision=L.NRF_PREC_F16_SPLIT)
    with Trainer(student["embedder"], student["embeddirs"], student["mlp"], student["table"], student["mlp_blob"], learning_rate=1e-2, mlp_backward="f16", hash_backward="binned") as tr:
        psnr0 = psnr_held()
        losses = []

        def one(i):
            ds.SetCurrentIter(i)
            b = ds.get_batch()
            rp.RawNoiseStd = max(0.0, 1.0 - float(i) / (float(iters) / 8.0))                     # FillRenderParams, NeRFExecutor.h:411
            lm, _ = tr.step(b["rays_o"], b["rays_d"], b["target_s"], rp)
            return lm
        for i in range(3):                                        # untimed: first-use allocations
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + iters):
            losses.append(one(i))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        skipped = int(getattr(tr, "skipped_steps", 0))
    # (leaving the `with` gives the student its full baked image back: the evaluation render below is the render fast path)
    psnr1 = psnr_held()
    l = [float(x[0]) for x in (losses[0], losses[len(losses) // 2], losses[-1])]
    rec = dict(workload="hashnerf_train_run", iterations=iters, rays_per_iteration=n_rand, samples="64+192", views=views, view_size=[hw, hw], seconds=dt, ms_per_step=dt / iters * 1e3,
               value=n_rand * 256 * iters / dt, unit="ray-samples/s (64 + 192 = 256 per ray)", rays_per_s=n_rand * iters / dt, iterations_per_s=iters / dt,
               loss_first_mid_last=l, psnr_held_out_view_before_after_db=[psnr0, psnr1], skipped_steps=skipped,
               includes="ray-batch producer (random pixels + rays + target gather) + render + huber + backward + Adam, every iteration",
               schedule="main.cpp:189,231-232 (16 384 rays, 64 + 192), lr 1e-2 constant (the reference's decay loop updates a copy: golden train_curve), RawNoiseStd per FillRenderParams")
    if isinstance(cpu_reference, dict) and cpu_reference.get("rays_per_s"):
        rec["cpu_reference"] = dict(iterations_per_s=cpu_reference["rays_per_s"] / n_rand, rays_per_s=cpu_reference["rays_per_s"], kind="reference", cores=cpu_reference.get("cores"),
                                    sample=cpu_reference.get("sample"))
    return rec


def dp_train_step_measurement(L, scene, comm, rank, world, sync, agree_max, n_rand=16384, steps=5):
    """Data-parallel training at N > 1 (every rank calls this): one optimisation step of NeRFExecutor::Train per rank on ITS n_rand rays, the gradients averaged by
    nrf_allreduce_grads behind the C ABI (CabiGradSync over the render path's communicator; torch.distributed's GradSync when that communicator does not exist), Adam on
    every replica.  Weak scaling: value = world * n_rand rays per step.  Returns the record (rank 0 reports it)."""
    import torch
    from nerfpp_amd import renderer as R
    from nerfpp_amd.dist import CabiGradSync, GradSync
    from nerfpp_amd.train import Trainer
    sc = scene.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0 + 9.0 * rank, -30.0, 4.0)          # every rank its own view: its own ray batch
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    torch.manual_seed(100 + rank)
    tgt = torch.rand((n_rand, 3), device="cuda")
    gs = CabiGradSync(comm) if comm is not None else GradSync(world)
    rp = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True,
                            BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    with Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned", grad_sync=gs) as tr:
        for _ in range(2):
            tr.step(o, d, tgt, rp)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            lm, _ = tr.step(o, d, tgt, rp)
        sync()
        dt = agree_max((time.perf_counter() - t0) / steps)
        # replicas must agree: the checksum of the parameters, max - min over the ranks
        chk = float(tr.blob.double().sum().item() + tr.table.double().sum().item())
        spread = agree_max(chk) - (-agree_max(-chk))
    return dict(workload="hashnerf_train_step_dp", n_gpus=world, rays_per_step=n_rand * world, samples="64+128", ms_per_step=dt * 1e3, value=n_rand * world * UNITS_PER_RAY / dt,
                unit="ray-samples/s", steps=steps, batch="weak scaling: 16 384 rays per GPU per step", replicas_checksum_spread=spread,
                gradient_exchange="nrf_allreduce_grads (C ABI: bucketed ncclAllReduce + scale, overflow agreement first)" if comm is not None else "torch.distributed all_reduce (GradSync)",
                skipped_steps=int(getattr(tr, "skipped_steps", 0)))


def full_frame_parity(sc, renderer, rp, K, c2w, args, scene, L):
    """The whole frame just timed against this library's own NRF_PREC_F32 mode (which equals the CPU oracle bit for bit -- tests/ and the 256-ray sample
    below) on identical weights and pose: every pixel value of the 800x800 frame (a 100-row band for the classic 8x256 network, whose fp32 path takes seconds per frame)."""
    import copy
    import torch
    if args.precision == "f32":
        return "the timed mode IS the parity mode"
    rows = H if args.workload == "hash" else 100
    row0 = (H - rows) // 2
    a = renderer.Render(H, W, K, rp, c2w=c2w, row0=row0, rows=rows)
    rp32 = copy.copy(rp); rp32.Precision = L.NRF_PREC_F32; rp32.Chunk = 32768 if args.workload == "hash" else 8192
    b = renderer.Render(H, W, K, rp32, c2w=c2w, row0=row0, rows=rows)
    d = (a.Outputs.RGBMap - b.Outputs.RGBMap).abs()
    mse = float((d.double() ** 2).mean())
    return dict(pixels=int(rows * W), max_abs_err=float(d.max()), median_abs_err=float(d.median()), frac_within_1e4=float((d < 1e-4).float().mean()),
                psnr=(float("inf") if mse == 0 else -10.0 * float(np.log10(mse))), against="NRF_PREC_F32 (bit-exact with the CPU oracle) on the same frame")


def quality_check(sc, renderer, rp, K, c2w, args, nrays=256):
    """Second half of the cpu_baseline leg (outside every timed region): the CPU oracle renders 256 rays of the frame with the same weights
    and the GPU pixels are compared with it -- oracle/ is used as the checker only, never as part of what is measured or shipped."""
    import torch
    from oracle import capi as O
    from nerfpp_amd import scene
    res = renderer.Render(H, W, K, rp, c2w=c2w, row0=H // 2, rows=1)
    rays = res.Extras["rays_flat"].cpu().numpy()[::W // nrays][:nrays]
    rgb = res.Outputs.RGBMap.cpu().numpy().reshape(-1, 3)[::W // nrays][:nrays]
    if args.workload == "hash":
        cfg = sc["cfg"]
        if sc["mode"] == "cu":
            ls = ((1 << cfg["log2_t"]) >> 4) << 4
            Lv = cfg["n_levels"]
            model = O.Model(2, sc["mlp_blob"], bbox=sc["bbox"], table_f16=O.f32_to_f16(sc["table"]), primes=sc["primes"],
                            local_idx=np.arange(Lv, dtype=np.int32) * ls, local_size=np.full(Lv, ls, np.int32), bias=np.zeros((Lv, 3), np.float32),
                            mul=O.hash_cu_scales(Lv, cfg["base"], cfg["finest"]))
        else:
            model = O.Model(0, sc["mlp_blob"], bbox=sc["bbox"], table_f32=sc["table"])
    else:
        model = O.Model(1, sc["mlp_blob"], bbox=sc["bbox"])
    ref = O.render_rays(model, rays, NS, NI, O.linspace(0, 1, NS), O.linspace(0, 1, NI), white_bkgr=True)
    d = np.abs(rgb - ref["rgb"])
    return dict(psnr=scene.psnr(rgb, ref["rgb"]), max_abs_err=float(d.max()), median_abs_err=float(np.median(d)), frac_within_1e4=float((d < 1e-4).mean()),
                rays=int(rays.shape[0]))
