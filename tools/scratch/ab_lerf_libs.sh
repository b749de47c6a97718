for i in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/tune/$v/libnerfpp_hip.so; fi
    LERF_OUT=/tmp/lerf_$v.pt timeout -k 10 300 python tools/scratch/lerf_ab_lib.py 2>/dev/null | tail -1
  done
done
python - <<'P'
import torch,sys,glob
fs=sorted(glob.glob('/tmp/lerf_*.pt')); ref=torch.load(fs[0])
for f in fs[1:]:
    e=torch.load(f); cos=(e*ref).sum(1); print(f, 'vs', fs[0], 'cos min %.9f'%float(cos.min()), 'max abs diff %.3e'%float((e-ref).abs().max()))
P
