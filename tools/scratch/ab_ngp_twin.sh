# same-call alternating A/B of the LibTorch-twin frame (HashEmbedder mode) over library builds
for i in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    timeout -k 10 300 python bench.py --hash-mode ngp --no-cpu-baseline --no-also --no-parity --no-isolated --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k:round(v['ms']/10,3) for k,v in d['kernel_ms'].items()}, d['frame_sha256'][:8])"
  done
done
