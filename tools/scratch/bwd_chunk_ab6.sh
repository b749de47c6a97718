#!/bin/bash
# classic training step against the point count of a backward pass (NRF_BWD_CHUNK_LOG2 builds under tune/): step time and the backward workspace
for v in c18 c19 default c18 default; do
  if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
  echo "== $v"; timeout -k 10 200 python - <<'PY' 2>/dev/null | grep '^{'
import json, sys
sys.path.insert(0, ".")
from nerfpp_amd import _lib as L, scene
from benchlib import extras
r = extras.classic_train_step_measurement(scene, L)
print(json.dumps(dict(ms_per_step=round(r["ms_per_step"], 2), backward_workspace_GB=round((r.get("backward_workspace_bytes") or 0) / 2**30, 2))))
PY
done
