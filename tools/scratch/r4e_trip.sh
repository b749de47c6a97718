mkdir -p gpurun_out
python - <<'PY' > gpurun_out/r4e_chunk_kernels.log 2>&1
import sys, ctypes as C; sys.path.insert(0,'.')
import torch, time
from nerfpp_amd import _lib as L, scene
H=W=800
sc = scene.make_hash_scene(mode="cu"); K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0); r = sc["renderer"]
L.lib().nrf_set_render_lanes(1)
n=len(L.NRF_PROF_NAMES); ms=(C.c_double*n)(); cnt=(C.c_int64*n)()
for chunk in (131072, 147456, 163840, 196608, 320000):
    rp = scene.lego_render_params(sc["bbox"], 64, 128, chunk, L.NRF_PREC_F16_SPLIT)
    for _ in range(2): r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    L.lib().nrf_profile_enable(1); L.lib().nrf_profile_read(ms,cnt,1)
    t0=time.perf_counter()
    for _ in range(5): r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5*1e3
    L.lib().nrf_profile_read(ms,cnt,1); L.lib().nrf_profile_enable(0)
    print(chunk, round(dt,2), {nm: (round(ms[i]/5,2), cnt[i]//5) for i,nm in enumerate(L.NRF_PROF_NAMES)}, flush=True)
PY
cat gpurun_out/r4e_chunk_kernels.log
timeout -k 10 600 bash tools/scratch/r4_lane_sweep.sh > gpurun_out/r4e_lane_sweep.log 2>&1
cat gpurun_out/r4e_lane_sweep.log
