"""`python bench.py --gpus N` without a launcher: start the N ranks as child processes."""
import os
import subprocess
import sys
import time


def spawn_ranks(n, script):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would set), relay rank 0's JSON line, return non-zero if any rank failed.  The parent never
    initialises the GPU and never replaces itself (no exec): it waits for its children."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))      # as torch.distributed.run does: N ranks x all host cores of intra-op threads is oversubscription
    cmd = [sys.executable, script] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = ""
    deadline = time.time() + float(os.environ.get("NRF_BENCH_TIMEOUT", "1500"))
    try:
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()), daemon=True)
        th.start()
        rc = [None] * n
        while any(c is None for c in rc) and time.time() < deadline:
            for i, pr in enumerate(procs):
                if rc[i] is None:
                    rc[i] = pr.poll()
            if any(c not in (None, 0) for c in rc):
                break                                    # a rank died: its peers would wait in a collective for ever
            time.sleep(0.05)
        th.join(timeout=5.0)
        out0 = box.get("out", "") or ""
    finally:
        for pr in procs:                                 # exact PIDs of the children this process started
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except Exception:
                pass
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    codes = [pr.returncode for pr in procs]
    if any(c != 0 for c in codes) or not lines:
        print(f"[bench] rank exit codes {codes}" + ("" if lines else "; rank 0 printed no result line"), file=sys.stderr)
        return 1
    return 0
