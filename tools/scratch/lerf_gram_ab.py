"""LeRF split-precision frame: time of the passes and the oracle check (bench.lerf_measurement) -- run once per library build (NRF_LIB_PATH)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import benchlib.extras as bench
from nerfpp_amd import scene as S, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(-180.0, -30.0, 4.0)
r = bench.lerf_measurement(S, L, K, c2w, L.NRF_PREC_F16_SPLIT, repeats=5)
print(json.dumps(dict(lib=os.environ.get("NRF_LIB_PATH", "default"), s_per_frame=r["s_per_frame"], kernel_ms={k: round(v["ms_per_frame"], 1) for k, v in r["kernel_ms"].items()}, oracle=r["oracle_check"])))
