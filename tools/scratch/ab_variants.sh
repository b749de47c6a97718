# same-call alternating A/B of the default bench frame over library builds: default (working tree) and tune/<name>/libnerfpp_hip.so for each name given
# prints the two-lane frame time, the single-lane pass's frame time and its per-kernel ms per frame, the frame's sha256
for i in 1 2 3; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --no-parity --steps 20 --warmup 3 >/dev/null 2>&1
    python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json")); r = d["roofline"]; n = r["isolated_steps"]
print(sys.argv[1], "two lanes %.3f ms" % d["ms_per_step"], "one lane %.3f" % r["isolated_ms_per_step"], {k: round(v["ms"] / n, 3) for k, v in r["isolated_kernel_ms"].items() if v["ms"] > 0}, d["frame_sha256"][:8], flush=True)
PY
  done
done
