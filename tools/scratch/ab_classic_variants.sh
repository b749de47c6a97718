# same-call alternating A/B of the classic-NeRF frame (tools/scratch/classic_time.py f16x3) over library builds: default and tune/<name>/libnerfpp_hip.so
for i in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    echo "$v $(timeout -k 10 300 python tools/scratch/classic_time.py f16x3 2>/dev/null | tail -1)"
  done
done
