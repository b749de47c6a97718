# training step with the forward render's Chunk loop on two lanes vs one (16 384-ray batch), alternating
for i in 1 2 3; do
  for lanes in 2 1; do
    export NRF_RENDER_LANES=$lanes
    echo -n "lanes $lanes: "; timeout -k 10 300 python tools/scratch/train_prof.py 16384 f16 binned --fast-only 2>/dev/null | grep "step ms"
  done
done
