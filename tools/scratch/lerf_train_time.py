"""LeRF training step time (bench.py's `lerf_train_step` also-line alone) and whether the fp32 layer products run on rocBLAS.  usage: python tools/scratch/lerf_train_time.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene
from benchlib import extras
r = extras.lerf_train_step_measurement(scene, L)
print(json.dumps(dict(fp32_gemm=int(L.lib().nrf_fp32_gemm_available()), ms_per_step=r["ms_per_step"], loss=r["loss_first_last"])))
