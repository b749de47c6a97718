"""Multi-GPU sharding of the render path: whole-image batches are partitioned by ray (contiguous row tiles), the model
(hash table 32-64 MiB + MLP <= 1.2 MB) is replicated read-only on every GPU, and one all-gather per step returns the
per-tile pixels to every rank (RCCL over xGMI on the GPU box; the same code runs on gloo for the CPU tests).

The reference is single-process / single-GPU; this is new work with no reference counterpart.
Row-contiguous tiles keep the reference's row-major pixel <-> ray index mapping (RayUtils.h:5-21) trivially intact: ray
r of tile t is pixel (row0_t + r // W, r % W).  The payload is tiny (800x800x3 fp32 = 7.7 MB per frame across all ranks),
so the collective is latency-bound; it is issued once per step for all frames of the step, not per chunk.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


def tile_partition(h, world, rank):
    """(row0, rows) of rank's contiguous row tile: nrf_tile_partition, the one definition both hosts (C++ adapter, this mirror) use."""
    r0, rr = C.c_int(0), C.c_int(0)
    L.check(L.lib().nrf_tile_partition(int(h), int(world), int(rank), C.byref(r0), C.byref(rr)))
    return r0.value, rr.value


class TileShard:
    def __init__(self, h, w, rank=0, world=1, force_collective=False):
        self.h, self.w, self.rank, self.world = h, w, rank, world
        self.force_collective = force_collective
        parts = [tile_partition(h, world, r) for r in range(world)]
        self.row0_of = [p[0] for p in parts]
        self.rows_of = [p[1] for p in parts]
        self.rows, self.row0 = self.rows_of[rank], self.row0_of[rank]
        self.max_rows = max(self.rows_of)

    def all_gather_frames(self, tiles, overlap=False):
        """tiles: list (one per frame of the step) of this rank's [rows, W, C] pixel tiles.
        Returns [frames, H, W, C] on every rank.  world == 1: a stack, no collective.
        overlap = True returns (frames, work): the collective is issued asynchronously (its own RCCL stream, ordered behind the kernels that produce the tiles), so
        the caller's next render is not held behind it; `frames` may be read -- on the current stream -- after work.wait() (work is None where nothing is pending).
        That is what a renderer of consecutive frames does: a rank's tile at 8 GPUs is ~3 ms of kernels, the latency-bound all-gather a tenth of that."""
        local = torch.stack([t.reshape(self.rows, self.w, -1) for t in tiles], 0)       # [F, rows, W, C]
        if self.world == 1 and not self.force_collective:
            return (local, None) if overlap else local
        f, _, w, c = local.shape
        if self.rows != self.max_rows:                                                    # uneven split: pad to the tallest tile
            pad = torch.zeros((f, self.max_rows - self.rows, w, c), device=local.device, dtype=local.dtype)
            local = torch.cat([local, pad], 1)
        local = local.contiguous()
        if f == 1 and self.rows == self.max_rows and self.h == self.world * self.rows:
            # one frame, equal tiles (800 rows over 2 / 4 / 8 ranks): the gathered buffer IS the frame -- no copy after the collective
            out = torch.empty((1, self.h, w, c), device=local.device, dtype=local.dtype)
            work = dist.all_gather_into_tensor(out.view(self.h, w, c), local.view(self.rows, w, c), async_op=overlap)   # dim-0 concatenation of the tiles = the frame's rows
            return (out, work) if overlap else out
        out = torch.empty((self.world * f,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local)                                           # ncclAllGather over xGMI (dim-0 concatenation)
        out = out.view((self.world, f) + tuple(local.shape[1:]))
        parts = [out[r, :, :self.rows_of[r]] for r in range(self.world)]
        res = torch.cat(parts, 1)                                                         # [F, H, W, C]
        return (res, None) if overlap else res                                            # uneven tiles / several frames: re-assembled behind the collective, nothing left pending


class TileComm:
    """The collective behind the C ABI (nrf_comm_* / nrf_allgather_tiles, include/nerfpp_hip.h): what the C++ / LibTorch host calls.  The RCCL unique id is
    created on rank 0 by the library and handed to the other ranks through torch.distributed (any backend; a file or a socket serves a host without it)."""

    @staticmethod
    def unique_id():
        """A fresh RCCL unique id (nrf_comm_unique_id) as bytes: created by ONE rank and handed to the others by whatever channel the host has."""
        buf = (C.c_ubyte * L.NRF_COMM_ID_BYTES)()
        L.check(L.lib().nrf_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, rank=0, world=1, group=None, timeout_s=300.0, unique_id=None):
        """timeout_s bounds the communicator's own rendezvous (nrf_comm_create_timeout): a peer that never arrives raises instead of parking this rank for ever.
        unique_id: the id every rank was handed (TileComm.unique_id() on one of them); None: rank 0 creates it and torch.distributed broadcasts it."""
        self.rank, self.world = int(rank), int(world)
        self._c = None
        buf = (C.c_ubyte * L.NRF_COMM_ID_BYTES)()
        if unique_id is not None:
            buf = (C.c_ubyte * L.NRF_COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        elif self.rank == 0:
            L.check(L.lib().nrf_comm_unique_id(buf))
        if self.world > 1 and unique_id is None:
            box = [bytes(buf)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = (C.c_ubyte * L.NRF_COMM_ID_BYTES).from_buffer_copy(box[0])
        self._c = C.c_void_p()
        L.check(L.lib().nrf_comm_create_timeout(buf, self.world, self.rank, C.c_double(float(timeout_s)), C.byref(self._c)))
        self._side = None

    class _Pending:
        """What all_gather_frames(overlap = True) hands back: wait() orders the CURRENT stream behind the collective (the host does not block)."""
        def __init__(self, event):
            self._e = event

        def wait(self):
            torch.cuda.current_stream().wait_event(self._e)

    def all_gather_frames(self, tiles, h, overlap=False):
        """tiles: [F, rows_rank, W, C] fp32 (contiguous) -> [F, h, W, C] on every rank; one fused RCCL launch on the current stream.
        overlap = True: the launch goes to a side stream ordered behind the current one and (frames, pending) is returned -- see TileShard.all_gather_frames."""
        tiles = tiles.contiguous()
        f, _, w, c = tiles.shape
        out = torch.empty((f, h, w, c), device=tiles.device, dtype=torch.float32)
        cur = torch.cuda.current_stream()
        st = cur
        if overlap:
            if self._side is None:
                self._side = torch.cuda.Stream()
            st = self._side
            st.wait_stream(cur)                      # the tiles' producers
            tiles.record_stream(st); out.record_stream(st)
        L.check(L.lib().nrf_allgather_tiles(self._c, C.c_void_p(tiles.data_ptr()), int(f), int(h), int(w), int(c), C.c_void_p(out.data_ptr()),
                                            C.c_void_p(st.cuda_stream)))
        if not overlap:
            return out
        ev = torch.cuda.Event()
        ev.record(st)
        return out, TileComm._Pending(ev)

    def all_reduce_grads(self, grads, overflow=None, bucket_bytes=32 << 20):
        """nrf_allreduce_grads: the fp32 gradient tensors become their mean over the ranks IN PLACE (bucketed ncclAllReduce + one scale, on the current stream).
        overflow: None -> no agreement, nothing synchronises, returns False; a bool -> the ranks first agree on it (one word, one read-back) and True comes back on EVERY
        rank iff any rank passed True -- no gradient is exchanged then."""
        gs = [g for g in grads if g is not None and g.numel()]
        for g in gs:
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise L.NrfError("all_reduce_grads: gradients must be contiguous fp32 tensors on the GPU")
        n = len(gs)
        ptrs = (C.c_void_p * max(n, 1))(*[g.data_ptr() for g in gs])
        counts = (C.c_int64 * max(n, 1))(*[g.numel() for g in gs])
        skip = C.c_int(0)
        L.check(L.lib().nrf_allreduce_grads(self._c, ptrs, counts, n, C.c_int64(int(bucket_bytes)), -1 if overflow is None else int(bool(overflow)), C.byref(skip),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return bool(skip.value)

    def close(self):
        if getattr(self, "_c", None):
            L.lib().nrf_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GradSync:
    """Data-parallel training step (SURVEY section 8f row N1 across GPUs): every rank renders and back-propagates its own ray batch against the replicated model,
    the gradients (hash table: 67 MB of fp32 at L16 T2^19 F2; MLP blob: 70 KB) are summed with ONE all-reduce per bucket and divided by the world size before Adam, so
    all replicas take the same step.  xGMI is point-to-point (7 links per GPU), so a ring all-reduce moves 2 (N-1)/N of the payload per link: the table gradient is
    sent in `bucket_bytes` slices to keep several links busy while the previous slice is being reduced; the blob rides in the last one.  No reference counterpart
    (the reference trains on one device)."""

    def __init__(self, world=None, bucket_bytes=32 << 20):
        self.world = (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1) if world is None else int(world)
        self.bucket_elems = max(int(bucket_bytes) // 4, 1)

    def any_overflow(self, flag, device=None):
        """True on EVERY rank iff any rank passes True: the fp16 backward's overflow report is per rank, the decision to skip the optimizer step must not be --
        a replica that steps while another skips (or that steps on the sum of a peer's inf) leaves the replicas with different parameters, moments and step counts."""
        if self.world == 1:
            return bool(flag)
        dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def reduce_or_skip(self, overflow, *grads):
        """The data-parallel step's exchange: agree on the overflow flag FIRST, then average the gradients only if nobody overflowed (a non-finite gradient is never
        summed into the peers').  Returns True when the step must be skipped -- on all ranks alike."""
        if self.any_overflow(overflow):
            return True
        self(*grads)
        return False

    def __call__(self, *grads):
        """grads: fp32 gradient tensors, reduced IN PLACE to their mean over the ranks."""
        if self.world == 1:
            return grads
        handles = []
        for g in grads:
            flat = g.view(-1)
            for i in range(0, flat.numel(), self.bucket_elems):
                handles.append(dist.all_reduce(flat[i:i + self.bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
        for h in handles:
            h.wait()
        inv = 1.0 / self.world
        for g in grads:
            g.mul_(inv)
        return grads


class CabiGradSync:
    """GradSync's interface over the C ABI (nrf_allreduce_grads: what a C++ host calls through nrfpp::TileComm::AllReduceGrads): Trainer(grad_sync=CabiGradSync(comm)).
    `comm` is a TileComm (one RCCL communicator per rank, shared with the render path's all-gather)."""

    def __init__(self, comm, bucket_bytes=32 << 20):
        self.comm, self.world, self.bucket_bytes = comm, int(comm.world), int(bucket_bytes)

    def reduce_or_skip(self, overflow, *grads):
        return self.comm.all_reduce_grads(grads, overflow=bool(overflow), bucket_bytes=self.bucket_bytes)

    def __call__(self, *grads):
        self.comm.all_reduce_grads(grads, overflow=None, bucket_bytes=self.bucket_bytes)
        return grads
