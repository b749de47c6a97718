for i in 1 2; do
python bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2rank gloo', d['ms_per_step'], d['host_ms_per_tile'])"
done
NRF_RENDER_LANES=1 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1rank lanes1', d['ms_per_step'], d['host_ms_per_tile'])"
python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-also --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1rank lanes2', d['ms_per_step'], d['host_ms_per_tile'])"
