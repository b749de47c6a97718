# same-call alternating A/B over library builds, two lanes and one: default (working tree) and tune/<name>/libnerfpp_hip.so for each name given
for i in 1 2; do
  for lanes in 2 1; do
    for v in default "$@"; do
      if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
      export NRF_RENDER_LANES=$lanes
      timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --no-parity --no-isolated --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v lanes $lanes', round(d['ms_per_step'],3), {k:round(v['ms']/20,3) for k,v in d['kernel_ms'].items()}, d['frame_sha256'][:8])"
    done
  done
done
