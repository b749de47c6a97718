"""Worker of tests/test_gpu_parity.py::test_bench_step_with_the_c_abi_collective_at_world_two_threads_as_ranks.

bench.py's own step function (benchlib/steps.py FrameStepper: render this rank's row tile with one nrf_render_rows call, all-gather through TileComm = nrf_allgather_tiles
behind the C ABI, overlapped as in the bench) at world size 2 on ONE GPU: the ranks are two THREADS of this process and RCCL is tests/helpers/mock_rccl.cpp, named to the
library through NRF_RCCL_LIBRARY (torch maps the real RCCL, which refuses two ranks on one device).  Strong scaling (one frame per step, 400-row tiles... here a small
frame) and weak (two frames per step); every rank's gathered frames must equal the single-rank render bit for bit.  Prints one JSON line; exit code 0 iff all equal."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NRF_RCCL_LIBRARY"] = os.path.join(ROOT, "tests", "helpers", "_build", "librccl.so.1")

import torch                                     # noqa: E402
from nerfpp_amd import _lib as L, scene          # noqa: E402
from nerfpp_amd.dist import TileShard, TileComm  # noqa: E402
from benchlib.steps import FrameStepper          # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.cuda.set_device(0)
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (24, 32)
    K = scene.lego_K(h, w)
    uid = TileComm.unique_id()
    results, errors = {}, []
    # one model per rank (a replica, as on two GPUs): a renderer serves one caller at a time
    scs = [scene.make_hash_scene(mode="cu", log2_t=12 if world > 2 else 14, seed=5000) for _ in range(world)]
    full = {}
    modes = ("strong", "weak") if world <= 2 else ("strong",)          # weak scaling renders `world` frames per step: kept to the small world
    for scaling in modes:
        fs1 = FrameStepper(scs[0]["renderer"], scene.lego_render_params(scs[0]["bbox"], chunk=256 if h < 100 else 4096, precision=L.NRF_PREC_F16_SPLIT), K, h, w, TileShard(h, w, 0, 1), scaling, world)
        full[scaling] = fs1.step().clone()        # [frames, h, w, 3]: the single-rank render of the same poses (TileShard world 1: no collective)
        fs1.drain()
    torch.cuda.synchronize()

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                comm = TileComm(rank, world, timeout_s=60.0, unique_id=uid)
                shard = TileShard(h, w, rank, world)
                rp = scene.lego_render_params(scs[rank]["bbox"], chunk=256 if h < 100 else 4096, precision=L.NRF_PREC_F16_SPLIT)
                out = {"tile_rows": shard.rows, "row0": shard.row0}
                for scaling in modes:
                    fs = FrameStepper(scs[rank]["renderer"], rp, K, h, w, shard, scaling, world, comm=comm, overlap=True)
                    for _ in range(3):            # several steps: the overlapped gather of step k completes at step k + 1
                        frames = fs.step()
                    fs.drain()
                    st.synchronize()
                    out[scaling] = bool(torch.equal(frames, full[scaling]))
                    out["host_ms_per_tile"] = 1e3 * sorted(fs.host)[len(fs.host) // 2]
                out["ranks_seen_by_rccl"] = int(L.lib().nrf_comm_world(comm._c))
                results[rank] = out
        except Exception as e:                    # noqa: BLE001
            errors.append(f"rank {rank}: {e!r}")

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240.0)
    hung = any(t.is_alive() for t in th)
    ok = not hung and not errors and len(results) == world and all(all(r[m] for m in modes) and r["ranks_seen_by_rccl"] == world for r in results.values())
    ok = ok and sum(r["tile_rows"] for r in results.values()) == h
    print(json.dumps(dict(ok=ok, hung=hung, errors=errors, world=world, h=h, w=w, modes=list(modes), ranks={str(k): v for k, v in results.items()},
                          collective="nrf_allgather_tiles (C ABI) over tests/helpers/mock_rccl.cpp")), flush=True)
    os._exit(0 if ok else 1)


if __name__ == "__main__":
    main()
