# same-call alternating A/B of the LeRF split-precision frame: working-tree library vs tune/old/libnerfpp_hip.so (built from HEAD)
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export NRF_LIB_PATH=$PWD/tune/old/libnerfpp_hip.so; else unset NRF_LIB_PATH; fi
    timeout -k 10 400 python tools/scratch/lerf_gram_ab.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['s_per_frame']*1e3,2), d['kernel_ms'], d['oracle']['embedding_max_abs_err'], d['oracle']['fine_sample_set_bit_identical_rays'])"
  done
done
