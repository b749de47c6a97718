#!/bin/bash
for mb in 0 3 8 24 80 160 300 1200; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dense-mb $mb 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dense_mb=$mb value %.3e ms/step %.1f hash_ms/frame %.2f psnr %s'%(d['value'],d['ms_per_step'],d['kernel_ms']['hash']['ms']/d['steps'],d['psnr_vs_oracle_db']))"
done
