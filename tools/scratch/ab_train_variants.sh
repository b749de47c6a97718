# same-call alternating A/B of the training step (tools/scratch/train_prof.py 16384 f16 binned --fast-only) over library builds: default and tune/<name>/libnerfpp_hip.so
for i in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    echo "$v $(timeout -k 10 300 python tools/scratch/train_prof.py 16384 f16 binned --fast-only 2>/dev/null | grep -m2 'step ms\|render' | tr '\n' ' ')"
  done
done
