# classic NeRF frame (default split mode) per library build, alternating
for i in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    timeout -k 10 300 python bench.py --workload classic --no-cpu-baseline --no-also --no-parity --no-isolated --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), {k:round(v['ms']/3,2) for k,v in d['kernel_ms'].items() if v['ms']>0}, d['frame_sha256'][:8])"
  done
done
