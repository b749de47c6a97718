"""bench.py's step: render this rank's row tile of every frame of the step (one nrf_render_rows call each) and all-gather the tiles into whole frames.

Lives here (not as closures inside bench.py's main) so that tests can drive the SAME step function with threads as ranks over the mock RCCL
(tests/test_gpu_parity.py::test_bench_step_with_the_c_abi_collective_at_world_two_threads_as_ranks)."""
import time


class FrameStepper:
    """One rank's step of `scaling` ("strong": ONE frame per step cut into `world` row tiles; "weak": `world` frames per step, this rank renders its tile of each).

    collective: `comm` (nerfpp_amd.dist.TileComm: nrf_allgather_tiles, RCCL behind the C ABI -- what a C++ host calls) when given, else `shard.all_gather_frames`
    (torch.distributed).  With overlap the all-gather of step k is issued asynchronously and completed (stream order, no host wait) at step k + 1: the next frame's
    kernels are not held behind a latency-bound collective -- what a renderer of consecutive frames does."""

    def __init__(self, renderer, rp, K, h, w, shard, scaling, world, comm=None, overlap=True, poses=None):
        import torch
        from nerfpp_amd import scene
        self.torch = torch
        self.renderer, self.rp, self.K, self.h, self.w, self.shard, self.comm, self.overlap = renderer, rp, K, h, w, shard, comm, overlap
        nfr = 1 if scaling == "strong" else world
        # frames of one step: poses on the reference's test orbit (pose_spherical(theta, -30, 4), theta step 9 degrees); strong scaling: one frame
        self.poses = poses if poses is not None else [scene.pose_spherical(-180.0 + 9.0 * k, -30.0, 4.0) for k in range(nfr)]
        self.host = []                       # seconds inside each Render call (one nrf_render_rows call: ~110 asynchronous launches)
        self.pending = None

    def render_tiles(self):
        tiles = []
        for c2w in self.poses:
            t_h = time.perf_counter()
            tiles.append(self.renderer.Render(self.h, self.w, self.K, self.rp, c2w=c2w, row0=self.shard.row0, rows=self.shard.rows).Outputs.RGBMap)
            self.host.append(time.perf_counter() - t_h)
        return tiles

    def step(self):
        tiles = self.render_tiles()
        if self.comm is not None:
            stacked = self.torch.stack([t.reshape(self.shard.rows, self.w, 3) for t in tiles], 0)
            if self.overlap:
                out, work = self.comm.all_gather_frames(stacked, self.h, overlap=True)
            else:
                out, work = self.comm.all_gather_frames(stacked, self.h), None
        elif self.overlap:
            out, work = self.shard.all_gather_frames(tiles, overlap=True)     # [frames, H, W, 3] on every rank; identity at N = 1
        else:
            out, work = self.shard.all_gather_frames(tiles), None
        if self.pending is not None:
            self.pending.wait()
        self.pending = work
        return out

    def drain(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
