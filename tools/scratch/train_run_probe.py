import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene
from benchlib import extras
print(json.dumps(extras.train_run_measurement(scene, L, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 400)))
