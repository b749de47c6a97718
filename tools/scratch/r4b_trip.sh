mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "relevancy or lerf" 2>&1 | tail -30 > gpurun_out/r4b_lerf_tests.log
cat gpurun_out/r4b_lerf_tests.log | tail -12
timeout -k 10 600 python - <<'PY' > gpurun_out/r4b_lerf_time.log 2>&1
import sys, json; sys.path.insert(0,'.')
import torch
from nerfpp_amd import _lib as L, scene
from benchlib import extras
from benchlib.costs import H, W
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
for prec in (L.NRF_PREC_F16_SPLIT,):
    rec = extras.lerf_measurement(scene, L, K, c2w, prec, repeats=6)
    print(json.dumps({k: rec[k] for k in ("s_per_frame","value","single_library_call","oracle_check","kernel_ms")}))
PY
tail -3 gpurun_out/r4b_lerf_time.log
