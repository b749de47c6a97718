"""Worker of tests/test_gpu_parity.py::test_two_real_rccl_ranks_*: one process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE), REAL RCCL.

Each rank renders its row tile of a small frame whose height does NOT divide by the world size (the ncclBroadcast-group path of nrf_allgather_tiles) and of one that
does (ncclAllGather), gathers through the C ABI (TileComm: nrf_comm_create with the bounded rendezvous, a blocking communicator) -- synchronously and in the overlapped
form bench.py uses -- and through torch.distributed, and compares both with the full frame it renders itself.  Prints one JSON line per rank; exit code 0 iff all equal."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from nerfpp_amd import _lib as L, scene
    from nerfpp_amd.dist import TileShard, TileComm
    sc = scene.make_hash_scene(mode="ngp", log2_t=14, seed=5000)
    comm = TileComm(rank, world, timeout_s=120.0)
    ok, report = True, {}
    for h, w in ((2 * world + 1, 24), (4 * world, 24)):            # uneven tiles, then even ones
        K = scene.lego_K(h, w); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
        rp = scene.lego_render_params(sc["bbox"], chunk=64, precision=L.NRF_PREC_F32)
        shard = TileShard(h, w, rank, world)
        full = sc["renderer"].Render(h, w, K, rp, c2w=c2w).Outputs.RGBMap.reshape(1, h, w, 3)
        tile = sc["renderer"].Render(h, w, K, rp, c2w=c2w, row0=shard.row0, rows=shard.rows).Outputs.RGBMap.reshape(1, shard.rows, w, 3)
        a = comm.all_gather_frames(tile, h)                          # nrf_allgather_tiles on the current stream
        b, pending = comm.all_gather_frames(tile, h, overlap=True)   # ... on a side stream, completed by stream order
        pending.wait()
        c = shard.all_gather_frames([tile[0]])                       # torch.distributed's collective on the same tiles
        torch.cuda.synchronize()
        same = bool(torch.equal(a, full) and torch.equal(b, full) and torch.equal(c.reshape(full.shape), full))
        report[f"{h}x{w}"] = dict(rows=shard.rows, row0=shard.row0, gathered_equals_full_frame=same)
        ok = ok and same
    seen = int(L.lib().nrf_comm_world(comm._c))
    ok = ok and seen == world
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    print(json.dumps(dict(rank=rank, ok=bool(flag.item() == 1.0), ranks_seen_by_rccl=seen, frames=report)), flush=True)
    sys.stdout.flush()
    os._exit(0 if flag.item() == 1.0 else 1)          # no tear-down collective (see bench.py)


if __name__ == "__main__":
    main()
