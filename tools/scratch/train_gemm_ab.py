"""A/B of the training GEMM arithmetic (nrf_set_train_gemm): classic and LeRF training steps of bench.py's `also` under fp32 products (0), bf16x3 (1), f16x3 (2)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene
from benchlib import extras
which = sys.argv[1:] or ["classic", "lerf"]
NAMES = {0: "f32", 1: "bf16x3", 2: "f16x3"}
for mode in (2, 1, 0, 2):
    L.check(L.lib().nrf_set_train_gemm(mode))
    for w in which:
        r = extras.classic_train_step_measurement(scene, L) if w == "classic" else extras.lerf_train_step_measurement(scene, L)
        print(json.dumps(dict(train_gemm=NAMES[mode], workload=r["workload"], ms_per_step=round(r["ms_per_step"], 3), loss=r["loss_first_last"])), flush=True)
