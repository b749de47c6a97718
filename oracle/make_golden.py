#!/usr/bin/env python3
"""make_golden.py -- TEST INFRASTRUCTURE.

Runs the compiled reference (oracle/_ref/ref_driver, built by oracle/build_ref.sh from
/root/reference) and packs the vectors it emits into tests/golden/<group>.npz plus
tests/golden/manifest.txt (one line per synthetic parameter tensor: tag, name, seed,
amplitude, shape -- see include/nrf_synth.h).

The fixtures are DATA (inputs + the reference's outputs).  Re-run only in a container that
has /root/reference; the GPU box consumes the committed .npz files.
"""
import os
import subprocess
import sys
import tempfile
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    drv = os.path.join(HERE, "_ref", "ref_driver")
    if not os.path.exists(drv):
        subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([drv, "golden", tmp])
        groups = defaultdict(dict)
        for fn in sorted(os.listdir(tmp)):
            if not fn.endswith(".npy"):
                continue
            group, name = fn[:-4].split(".", 1)
            groups[group][name] = np.load(os.path.join(tmp, fn))
        for g, arrs in groups.items():
            np.savez_compressed(os.path.join(out, g + ".npz"), **arrs)
            print(f"{g}: {len(arrs)} arrays, {sum(a.nbytes for a in arrs.values())} bytes raw")
        with open(os.path.join(tmp, "manifest.txt")) as f, open(os.path.join(out, "manifest.txt"), "w") as g:
            g.write(f.read())
    # checkpoint archives written by the reference's own torch::save calls (SaveCheckpoint, NeRFExecutor.h:1055-1070): format fixtures
    ck = os.path.join(out, "ckpt")
    os.makedirs(ck, exist_ok=True)
    subprocess.check_call([drv, "ckpt_save", ck])
    return 0


if __name__ == "__main__":
    sys.exit(main())
