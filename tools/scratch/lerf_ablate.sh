# Timing-only ablation builds of the LeRF split kernels (NRF_LERF_ABLATE bit mask, mlp_lerf_split_mfma.hip) into tune/abl_<mask>/ -- run here (no GPU needed):
#   bash tools/scratch/lerf_ablate.sh build 1 2 4 8 16 32 63 ...     then on the GPU box:   bash tools/scratch/lerf_ab_libs.sh default abl_1 abl_2 ...
# Results of an ablated build are garbage by construction; only its frame time means something.
CS=nerfpp_amd/csrc
FLAGS="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Iinclude -I$CS -Wall -Wno-unused-function -fno-honor-nans"
shift
for m in "$@"; do
  ( d=tune/abl_$m; mkdir -p $d
    /opt/rocm/bin/hipcc $FLAGS -DNRF_LERF_ABLATE=$m -c $CS/mlp_lerf_split_mfma.hip -o $d/x.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/libnerfpp_hip.so $d/x.o $(ls nerfpp_amd/lib/obj/*.o | grep -v mlp_lerf_split_mfma.o) && rm $d/x.o && echo built $d ) &
done
wait
