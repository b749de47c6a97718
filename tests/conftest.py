import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("filterwarnings", "ignore:.*torch.jit.*:DeprecationWarning")        # the reference's checkpoints ARE TorchScript archives


def load_golden(group):
    return dict(np.load(os.path.join(GOLDEN, group + ".npz")))


@pytest.fixture(scope="session")
def manifest():
    from nerfpp_amd.synth import load_manifest
    return load_manifest(os.path.join(GOLDEN, "manifest.txt"))
