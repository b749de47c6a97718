import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene
from benchlib import extras
w = sys.argv[1]
r = extras.classic_train_step_measurement(scene, L) if w == "classic" else extras.lerf_train_step_measurement(scene, L)
print(json.dumps(dict(workload=r["workload"], ms_per_step=r["ms_per_step"], train_gemm=L.lib().nrf_get_train_gemm())))
