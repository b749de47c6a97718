for rep in 1 2 3; do for cfg in "2 131072" "3 131072" "4 131072" "3 98304"; do set -- $cfg
NRF_RENDER_LANES=$1 python bench.py --steps 10 --warmup 3 --chunk $2 --no-cpu-baseline --no-parity --no-also --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $1 chunk $2 ms', d['ms_per_step'], 'host', d['host_ms_per_tile'], {k:v for k,v in d['roofline'].get('timed_two_lane_avg_launch_ms',{}).items()} if d.get('roofline') else '', d['roofline']['avg_launch_ms'], d['roofline'].get('hash',{}).get('avg_launch_ms'))"
done; done
