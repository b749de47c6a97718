"""Rooflines of the timed kernels from HIP-event totals, and the committed PMC summary (profiles/pmc_latest.json)."""
import json
import os

from .costs import *      # noqa: F401,F403

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mfma_roofline(kernel, units, flop_per_unit, seconds, launches, issued_flop_per_unit=None, peak=MFMA_F16_PEAK, note=None):
    r = dict(bound="mfma", kernel=kernel, achieved=units * flop_per_unit / max(seconds, 1e-12) / 1e12, peak=peak / 1e12, unit="TFLOP/s",
             frac=units * flop_per_unit / max(seconds, 1e-12) / peak, launches=launches, avg_launch_ms=seconds * 1e3 / max(launches, 1),
             units_per_launch=units / max(launches, 1), flop_per_unit=flop_per_unit)
    if issued_flop_per_unit:
        r["mfma_issued_frac"] = units * issued_flop_per_unit / max(seconds, 1e-12) / peak
        r["mfma_issued_vs_sustained_gemm"] = units * issued_flop_per_unit / max(seconds, 1e-12) / MFMA_F16_SUSTAINED_GEMM
    if note:
        r["note"] = note
    return r


# source files whose contents decide a kernel's HBM traffic: the PMC summary records their hashes, and a summary taken from other sources is not reported
PMC_KERNEL_SOURCES = {
    "hash_encode (k_hash_cu_lm)": ["hash_fast.hip", "hash_fast.h", "encode.h"],
    "mlp_small (k_mlp_small_mfma)": ["mlp_small_mfma.hip"],
    "sigma_small_f32 (k_sigma_small_f32)": ["sigma_small_f32.hip"],
    "mlp_nerf_split (k_mlp_nerf_split)": ["mlp_nerf_split_mfma.hip", "mlp_nerf_net.h"],
    "mlp_nerf (k_mlp_nerf_mfma)": ["mlp_nerf_mfma.hip", "mlp_nerf_net.h"],
}


def kernel_source_hash(kernel):
    import hashlib
    h = hashlib.sha256()
    for f in PMC_KERNEL_SOURCES.get(kernel, []):
        with open(os.path.join(ROOT, "nerfpp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, units_per_launch, meta_key=None):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/pmc_latest.json, written from separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of this same bench: bytes per point the kernel processed), times this run's points per launch.  The summary is stamped with the commit it was taken at
    and with a hash of each kernel's sources: if the kernel's source differs from what was profiled, NO traffic is reported (None) and the reason is given."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
        meta = d["_meta"]
        stamp = "PMC pass at commit " + str(meta.get("commit", "unrecorded (round 2)"))
        want = (meta.get("kernel_source_sha256_16") or {}).get(kernel)
        if want is None or want != kernel_source_hash(kernel):
            return None, f"not reported: kernel sources changed since the {stamp}; re-run tools/gpu_pmc_round.sh"
        # (2*FETCH_SIZE + WRITE_SIZE) KB per dispatch (gfx950 x2 read correction), separate --pmc passes of this bench, divided by the points processed
        return d[kernel]["hbm_bytes_per_point"] * units_per_launch, "profiles/pmc_latest.json@" + str(meta.get("commit", "?")) + " (kernel sources unchanged since)"
    except Exception:
        return None, None


def pmc_value(kernel, counter, per_point=False):
    """A raw counter of the committed PMC summary (summed over the profiled run's dispatches of `kernel`), optionally per point the kernel processed in that run."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        v = float(d[kernel][counter]) * float(d[kernel].get("launches", 1))          # the summary holds averages per logical launch
        return v / float(d[kernel]["points_in_run"]) if per_point else v
    except Exception:
        return None


def pmc_mfma_busy(kernel, precision):
    """Share of GPU-active cycles in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD), from the committed PMC
    passes; with the issued fraction at the nominal 2.4 GHz it gives the clock the chip held: clock = 2.4 GHz * issued_frac / busy_frac."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        return float(d[kernel]["mfma_busy_frac_of_active_cycles"][precision])
    except Exception:
        return None


def kernel_rooflines(prof, workload, precision, hash_mode, units_total):
    """Per-kernel rooflines from HIP-event totals.  `prof` = {slot: {ms, launches}} over the measured region (nrf_profile_read, events on the launch stream),
    `units_total` = this rank's ray-samples over that region (256 per ray).  Each entry prices the ALGORITHMIC bytes / flops of SURVEY.md section 8(d)
    times the points the kernel really processed (executed_per_ray) over the kernel's own launch time.  Returns {name: roofline}, dominant first."""
    out = []
    if workload == "hash":
        ex_hash, ex_mlp, ex_sigma = executed_per_ray(workload, precision, hash_mode)
        k = prof["hash"]
        upl = units_total * ex_hash / UNITS_PER_RAY / max(k["launches"], 1)
        dur = k["ms"] * 1e-3 / max(k["launches"], 1)
        achieved = upl * HASH_BYTES_PER_UNIT / max(dur, 1e-12)
        traffic, traffic_src = pmc_traffic("hash_encode (k_hash_cu_lm)", upl)
        # The gathers are cache-served (32 x 16-B loads per point out of the baked pyramid, ray-coherent: by counters a third of the requested bytes reach HBM), so the
        # ALGORITHMIC bytes over the kernel time exceed the 8 TB/s HBM peak -- that quotient is not a roofline fraction.  What is reported:
        #   frac = achieved / peak with achieved = the HBM bytes the counters saw per launch / kernel time (null when the kernel's sources changed since the counter passes);
        #   gather_frac_of_cache_ceiling = the 512 B of table reads per point / kernel time over the guide's 8.6 TB/s random-gather rate of an Infinity-Cache-resident table;
        #   valu_issue_frac = SQ_INSTS_VALU x 2 cycles (a wave64 instruction on a SIMD-32) over 1 024 SIMDs x kernel time x 2.4 GHz: the kernel's other bound;
        #   algorithmic_over_hbm_peak = the old figure (588 B per point / time / 8 TB/s), > 1, kept for continuity with rounds 1-4.
        alg = achieved
        hbm_rate = (traffic / max(dur, 1e-12)) if traffic else None
        valu_pp = pmc_value("hash_encode (k_hash_cu_lm)", "SQ_INSTS_VALU", per_point=True) if traffic else None
        hroof = dict(bound="hbm", kernel="hash_encode (k_hash_cu_lm)", achieved=(hbm_rate / 1e9) if hbm_rate else None, peak=HBM_PEAK / 1e9, unit="GB/s",
                     frac=(hbm_rate / HBM_PEAK) if hbm_rate else None, hbm_frac=(hbm_rate / HBM_PEAK) if hbm_rate else None,
                     gather_frac_of_cache_ceiling=upl * HASH_GATHER_BYTES_PER_UNIT / max(dur, 1e-12) / GATHER_PEAK,
                     valu_issue_frac=(valu_pp * upl * 2.0 / (1024 * max(dur, 1e-12) * 2.4e9)) if valu_pp else None,
                     valu_wave_instructions_per_point=valu_pp,
                     algorithmic_over_hbm_peak=alg / HBM_PEAK, algorithmic_gb_s=alg / 1e9, frac_of_l2_gather_ceiling=alg / GATHER_PEAK_L2,
                     traffic=traffic, traffic_source=traffic_src,
                     launches=k["launches"], avg_launch_ms=dur * 1e3, units_per_launch=upl, bytes_per_unit=HASH_BYTES_PER_UNIT,
                     l1_tag_lookup_floor_ms_at_2p1_ghz=upl * 32 / (256 * 2.1e9) * 1e3)
        out.append((k["ms"], "hash", hroof))
        mk, ck, sk = prof["mlp"], prof["mlp_colour"], prof["sigma"]
        mlp_peak = MFMA_F16_PEAK if precision != "f32" else F32_PEAK
        mlp_units = units_total * ex_mlp / UNITS_PER_RAY                  # whole-network launches (k_mlp_small_mfma<..., GEOIN = false>)
        mupl = mlp_units / max(mk["launches"], 1)
        mtraffic, mtraffic_src = pmc_traffic("mlp_small (k_mlp_small_mfma)", mupl)
        mroof = mfma_roofline("mlp_small (k_mlp_small_mfma)", mlp_units, SMALL_FLOP_PER_UNIT, mk["ms"] * 1e-3, mk["launches"],
                              issued_flop_per_unit=SMALL_MFMA_FLOP_PER_UNIT.get(precision), peak=mlp_peak)
        mroof.update(traffic=mtraffic, traffic_source=mtraffic_src)
        busy = pmc_mfma_busy("mlp_small (k_mlp_small_mfma)", precision)
        if busy:
            mroof["mfma_busy_frac_of_active_cycles"] = busy
        if ck["launches"]:
            # the colour-net-only launches of the fine pass (its S coarse depths; sigma and geo_feat come from the exact coarse kernel): instance GEOIN = true
            col_units = units_total * colour_only_per_ray(workload, precision) / UNITS_PER_RAY
            mroof["colour_only"] = mfma_roofline("mlp_small, colour net alone (GEOIN)", col_units, SMALL_COLOUR_FLOP_PER_UNIT, ck["ms"] * 1e-3, ck["launches"],
                                                 issued_flop_per_unit=SMALL_COLOUR_MFMA_FLOP_PER_UNIT, peak=mlp_peak)
        out.append((mk["ms"] + ck["ms"], "mlp", mroof))
        if sk["launches"]:
            s_units = units_total * ex_sigma / UNITS_PER_RAY
            sroof = mfma_roofline("sigma_small_f32 (k_sigma_small_f32; coarse pass, exact fp32 on v_mfma_f32_32x32x2_f32)", s_units, SIGMA_FLOP_PER_UNIT,
                                  sk["ms"] * 1e-3, sk["launches"], peak=F32_PEAK)
            straffic, straffic_src = pmc_traffic("sigma_small_f32 (k_sigma_small_f32)", s_units / sk["launches"])
            sroof.update(traffic=straffic, traffic_source=straffic_src, mfma_busy_frac_of_active_cycles=pmc_mfma_busy("sigma_small_f32 (k_sigma_small_f32)", "f16x3"))
            out.append((sk["ms"], "sigma", sroof))
    else:
        k = prof["mlp"]
        ex_mlp = executed_per_ray(workload, precision, hash_mode, coarse_full=False)[1]
        exec_units = units_total * ex_mlp / UNITS_PER_RAY
        peak = MFMA_F16_PEAK if precision != "f32" else F32_PEAK
        name = "mlp_nerf_split (k_mlp_nerf_split)" if precision == "f16x3" else "mlp_nerf (k_mlp_nerf_mfma)"
        # 1 058 matrix instructions per 32 points (feature_linear and views_linears_0 pre-multiplied into one affine layer), x3 products in split precision
        roof = mfma_roofline(name, exec_units, NERF_FLOP_PER_UNIT, k["ms"] * 1e-3, k["launches"],
                             issued_flop_per_unit=(3.0 if precision == "f16x3" else 1.0) * 1058 * 32768 / 32 if precision != "f32" else None, peak=peak)
        traffic, traffic_src = pmc_traffic(name, exec_units / max(k["launches"], 1))
        roof.update(traffic=traffic, traffic_source=traffic_src)
        busy = pmc_mfma_busy(name, precision)
        if busy:
            roof["mfma_busy_frac_of_active_cycles"] = busy
        out.append((k["ms"], "mlp", roof))
        sk = prof["sigma"]
        if sk["launches"]:
            s_units = units_total * executed_per_ray(workload, precision, hash_mode)[2] / UNITS_PER_RAY
            out.append((sk["ms"], "sigma", mfma_roofline("sigma_nerf_f32 (coarse pass: density branch in exact fp32)", s_units, NERF_SIGMA_FLOP_PER_UNIT,
                                                         sk["ms"] * 1e-3, sk["launches"], peak=F32_PEAK)))
    out.sort(key=lambda c: -c[0])
    return {name: r for _, name, r in out}


def prof_table(ms, cnt, names):
    return {n: dict(ms=ms[i], launches=int(cnt[i])) for i, n in enumerate(names)}
