"""Two ranks on one GPU over gloo: the time of an all_gather of frame-sized GPU tensors (the rehearsal path of bench.py --gpus 2 --backend gloo), by itself."""
import os, sys, time, torch, torch.distributed as dist
def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = torch.rand((400 * 800, 3), device="cuda")
    out = torch.empty((world * 400 * 800, 3), device="cuda")
    for name, fn in (("all_gather_into_tensor (GPU tensors)", lambda: dist.all_gather_into_tensor(out, x)),
                     ("all_gather via CPU copies", lambda: (lambda xc, oc: (dist.all_gather_into_tensor(oc, xc), out.copy_(oc)))(x.cpu(), torch.empty((world * 400 * 800, 3))))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if rank == 0: print("%s: %.1f ms" % (name, dt * 1e3), flush=True)
    dist.destroy_process_group()
main()
