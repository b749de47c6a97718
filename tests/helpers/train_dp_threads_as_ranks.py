"""Worker of tests/test_gpu_parity.py::test_data_parallel_training_over_the_c_abi_all_reduce_threads_as_ranks.

Data-parallel training at world size 2 on ONE GPU: the ranks are two THREADS of this process, each with its own replica (hash grid + NeRFSmall) and its own half of every
ray batch, Trainer(grad_sync=CabiGradSync(TileComm)) = nrf_allreduce_grads behind the C ABI over tests/helpers/mock_rccl.cpp (NRF_RCCL_LIBRARY).  After N steps
  * the two replicas hold the SAME parameters bit for bit (same averaged gradient, same Adam step), and
  * the averaged gradient of the first step equals the gradient of ONE process that took the whole batch (the mean over 2n rays is the mean of the two ranks' means)
    to rounding, and the parameters after the steps stay near that process's.
Prints one JSON line; exit code 0 iff both hold."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NRF_RCCL_LIBRARY"] = os.path.join(ROOT, "tests", "helpers", "_build", "librccl.so.1")

import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from nerfpp_amd import _lib as L, scene           # noqa: E402
from nerfpp_amd import renderer as R              # noqa: E402
from nerfpp_amd.dist import TileComm, CabiGradSync          # noqa: E402
from nerfpp_amd.train import Trainer              # noqa: E402


def make(seed=5000):
    return scene.make_hash_scene(mode="cu", log2_t=14, seed=seed, table_amp=1e-2, sigma_scale=4.0)


def main():
    world, steps, n = 2, 4, 512
    mlp_backward = sys.argv[1] if len(sys.argv) > 1 else "f32"
    torch.cuda.set_device(0)
    K = scene.lego_K(64, 64); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(64, 64, K, c2w)
    o = o.reshape(-1, 3)[: world * n].contiguous(); d = d.reshape(-1, 3)[: world * n].contiguous()
    torch.manual_seed(3)
    tgt = torch.rand((world * n, 3), device="cuda")
    rp = R.NeRFRenderParams(NSamples=32, NImportance=32, Chunk=n * world, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX,
                            Precision=L.NRF_PREC_F16_SPLIT if mlp_backward == "f16" else L.NRF_PREC_F32)
    hb = "binned" if mlp_backward == "f16" else "f32"
    # one process, the whole batch
    sc = make()
    with Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=1e-2, mlp_backward=mlp_backward, hash_backward=hb) as tr:
        g1 = None
        for _ in range(steps):
            tr.step(o, d, tgt, rp)
            if g1 is None:
                g1 = (tr.g_table.clone(), tr.g_blob.clone())          # the first step's gradient (same initial parameters everywhere)
        torch.cuda.synchronize()
        whole = (tr.table.clone(), tr.blob.clone())
    uid = TileComm.unique_id()
    scs = [make() for _ in range(world)]
    out, errors = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                comm = TileComm(rank, world, timeout_s=60.0, unique_id=uid)
                s_ = scs[rank]
                with Trainer(s_["embedder"], s_["embeddirs"], s_["mlp"], s_["table"], s_["mlp_blob"], learning_rate=1e-2, mlp_backward=mlp_backward, hash_backward=hb,
                             grad_sync=CabiGradSync(comm, bucket_bytes=1 << 16)) as tr:
                    sl = slice(rank * n, (rank + 1) * n)
                    gr1 = None
                    for _ in range(steps):
                        tr.step(o[sl].contiguous(), d[sl].contiguous(), tgt[sl].contiguous(), rp)
                        if gr1 is None:
                            gr1 = (tr.g_table.clone(), tr.g_blob.clone())      # already the mean over the ranks
                    st.synchronize()
                    out[rank] = (tr.table.clone(), tr.blob.clone(), int(L.lib().nrf_comm_world(comm._c)), int(getattr(tr, "skipped_steps", 0)), gr1)
        except Exception as e:                    # noqa: BLE001
            errors.append(f"rank {rank}: {e!r}")

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240.0)
    hung = any(t.is_alive() for t in th)
    rec = dict(hung=hung, errors=errors, mlp_backward=mlp_backward)
    ok = not hung and not errors and len(out) == world
    if ok:
        same = bool(torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]))
        moved = float((out[0][1] - torch.as_tensor(scs[0]["mlp_blob"], device="cuda")).abs().max())
        rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
        # against the single process: the DIFFERENCE from the initial parameters (what training did), norm-wise
        t0 = torch.as_tensor(np.asarray(scs[0]["table"], np.float32), device="cuda"); b0 = torch.as_tensor(np.asarray(scs[0]["mlp_blob"], np.float32), device="cuda")
        rec.update(replicas_bit_identical=same, blob_moved_max=moved, table_update_rel_err_vs_whole_batch=rel(out[0][0] - t0, whole[0] - t0),
                   blob_update_rel_err_vs_whole_batch=rel(out[0][1] - b0, whole[1] - b0), ranks_seen_by_rccl=out[0][2], skipped_steps=[out[0][3], out[1][3]])
        # the averaged gradient of the FIRST step against the whole-batch gradient: the exchange itself (fp32 chain: float atomics' order only; fp16 chain: a loss scale per rank)
        rec.update(step1_table_grad_rel_err=rel(out[0][4][0], g1[0]), step1_blob_grad_rel_err=rel(out[0][4][1], g1[1]))
        gtol = 3e-2 if mlp_backward == "f16" else 1e-4
        # the parameters after four Adam steps only loosely: eps = 1e-15 makes an entry's step lr * sign(g) however small g is, so rounding-level gradients move whole steps
        ok = same and moved > 1e-4 and rec["step1_table_grad_rel_err"] < gtol and rec["step1_blob_grad_rel_err"] < gtol and rec["table_update_rel_err_vs_whole_batch"] < 0.5 and \
            rec["blob_update_rel_err_vs_whole_batch"] < 0.5 and out[0][2] == world
    rec["ok"] = ok
    print(json.dumps(rec), flush=True)
    os._exit(0 if ok else 1)


if __name__ == "__main__":
    main()
