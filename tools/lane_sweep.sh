# item 4(a): the Chunk loop on 1 / 2 / 3 / 4 lanes at several chunk sizes -- frame time of the default bench frame, same call
for lanes in 1 2 3 4; do for chunk in 32768 65536 98304 131072 163840; do
NRF_RENDER_LANES=$lanes python bench.py --steps 10 --warmup 3 --chunk $chunk --no-cpu-baseline --no-parity --no-also --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $lanes chunk $chunk ms', d['ms_per_step'], d['frame_sha256'][:8])"
done; done
