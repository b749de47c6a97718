"""Workload shape and per-unit algorithmic costs (SURVEY.md section 8d / BASELINE.md section 2) behind bench.py's rooflines."""
H = W = 800
NS, NI = 64, 128
UNITS_PER_RAY = NS + NS + NI          # one shared network, fine pass re-evaluates all depths (NeRFRenderer.h:422,447): the metric's unit count per ray
# What the kernels actually execute per ray on the matrix-core paths: the S coarse depths of the fine set are the coarse pass's own sample points, so their hash
# features (default split mode) or their whole network outputs (coarse pass = whole network in the same arithmetic) are reused -- results unchanged.  `value` keeps
# counting the reference's 256 evaluations per ray (the work the frame stands for); the rooflines below price what each kernel really processed.


def executed_per_ray(workload, precision, hash_mode, coarse_full=False):
    """(hash-encode points, fused-MLP points, sigma-only points) per ray."""
    if precision == "f32":
        return UNITS_PER_RAY, UNITS_PER_RAY, 0
    if workload == "classic":
        if precision == "f16x3" and coarse_full is False:
            return 0, NI, NS                                      # coarse pass: density branch in exact fp32 + colour branch on the exact h8 (sigma_nerf_f32.hip); the fine pass evaluates the 128 new depths
        return 0, NS + NI, 0                                      # whole network on the coarse pass, its outputs reused by the fine pass: 64 + 128 evaluations
    if precision == "f16x3":                                  # coarse pass: sigma net alone (exact fp32), which hands (sigma, geo_feat) to the fine pass
        return NS + NI, NI, NS                                # both encoders: the fine pass keeps the coarse pass's feature columns; whole network on the new samples only
    return NS + NI, NS + NI, 0                                # plain fp16: coarse outputs reused by the fine pass


def colour_only_per_ray(workload, precision):
    """Points per ray at which the fused-MLP kernel runs the colour net alone (HashNeRF default mode: the fine pass's S coarse depths, whose sigma-net output comes
    from the exact coarse kernel)."""
    return NS if (workload == "hash" and precision == "f16x3") else 0

# algorithmic cost per ray-sample (SURVEY.md section 8d / BASELINE.md section 2)
HASH_BYTES_PER_UNIT = 16 * 8 * 2 * 2 + 12 + 64      # table gathers + point in + fp16 features out (standalone encode kernel)
HASH_GATHER_BYTES_PER_UNIT = 16 * 8 * 2 * 2         # the table gathers alone: 8 corners x F = 2 fp16 on each of 16 levels
SMALL_FLOP_PER_UNIT = 35072
SMALL_COLOUR_FLOP_PER_UNIT = 2 * ((16 + 15) * 64 + 64 * 64 + 64 * 64 + 64 * 3)      # the colour net alone (NeRF.cpp:383-406): 20 736
SMALL_COLOUR_MFMA_FLOP_PER_UNIT = 72 * 32768 // 32                                   # its 72 of the split kernel's 116 matrix instructions per 32 points
# matrix-core work the NeRFSmall kernel actually issues per point (32-row / 16-k padded tiles; x3 products in split mode, x2 on layer 0)
SMALL_MFMA_FLOP_PER_UNIT = {"f16": 40 * 32768 // 32, "f16x3": 116 * 32768 // 32}
NERF_FLOP_PER_UNIT = 1186816
# the density branch alone (NeRF.cpp:92-108: pts_linears 0..7 with the skip-concat, alpha_linear): 491 264 MAC
NERF_SIGMA_FLOP_PER_UNIT = 2 * (63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256)
# the coarse pass of the default mode needs sigma only (NeRFRenderer.h:422-428): in 32 -> 64 -> 64 -> 1
SIGMA_FLOP_PER_UNIT = 2 * (32 * 64 + 64 * 64 + 64)
# LeRF head at main.cpp:203-213 sizes: 128 -> 256 -> 33 ; cat[geo32, in128] -> 256 -> 768, bias-free
LERF_FLOP_PER_UNIT = 557568
LERF_SIGMA_FLOP_PER_UNIT = 2 * (128 * 256 + 256 * 33)                        # the density net alone (the coarse pass's exact-fp32 kernel, geo rows included)
# matrix instructions (32x32x16 fp16 = 32 768 flop) the split-precision LeRF passes issue per 32 points: the sigma pass on the new samples (layer 0 on exact-fp16
# features: 2 products, layer 1: 3) and the embedding pass from LE0 on (LE0: 8 tiles x (8 x 2 + 4 x 3), Gram: 8 x 16 x 1); the 256 -> 768 layer runs once per RAY
LERF_SPLIT_MFMA_SIGMA = 8 * 8 * 2 + 2 * 16 * 3
LERF_SPLIT_MFMA_EMBED = 8 * (8 * 2 + 4 * 3) + 8 * 16 * 1                       # LE0 in split precision; the Gram product (a scalar norm per sample) on the hi parts only
LERF_HASH_BYTES_PER_UNIT = 16 * 8 * 8 * 2 + 12 + 16 * 8 * 2                    # CuHash F = 8: 2 048 B of table reads + the point + 256 B of level-major fp16 features
HBM_PEAK = 8.0e12
MFMA_F16_PEAK = 2.5e15
# MI355X_MICROARCH.md, "DVFS give-back" item 1: the chip lowers its clock under matrix load; a tuned bf16 GEMM on random data holds 1.90-1.95 GHz and
# delivers 1 247 TFLOP/s (1 483 on all-zero operands at 2.30 GHz).  The rate the matrix pipes SUSTAIN on real data, as measured by the guide.
MFMA_F16_SUSTAINED_GEMM = 1.247e15
F32_PEAK = 157.3e12
# MI355X_MICROARCH.md, "Indexed rows: gather": uniformly random rows of a table served from the Infinity Cache read at 8.6 TB/s chip-wide
# (16.8-18.8 TB/s when every row is L2-resident, 6.0 TB/s swept from HBM) -- the ceiling of the vector-memory gather path the hash encode runs on
GATHER_PEAK = 8.6e12
GATHER_PEAK_L2 = 16.8e12
