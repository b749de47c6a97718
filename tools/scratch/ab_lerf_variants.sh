# same-call alternating A/B of the LeRF split-precision frame over library builds: default (working tree) and tune/<name>/libnerfpp_hip.so for each name given
for i in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    timeout -k 10 400 python tools/scratch/lerf_gram_ab.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['s_per_frame']*1e3,2), d['kernel_ms'], d['oracle']['embedding_max_abs_err'], d['oracle']['fine_sample_set_bit_identical_rays'])"
  done
done
