# builds and runs tools/scratch/wstream_probe.hip over chunk lengths (k-steps per chunk) x prefetch depths (chunks ahead); on the GPU box: bash tools/scratch/wstream_probe.sh
for cfg in "16 2" "8 2" "8 3" "8 4" "4 2" "4 3" "4 4" "4 6"; do
  set -- $cfg
  hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -DKSTEPS=$1 -DAHEAD=$2 tools/scratch/wstream_probe.hip -o /tmp/wsp_$1_$2 2>/dev/null && timeout -k 10 60 /tmp/wsp_$1_$2 | grep -E "^mode [01]" | tail -2
done
