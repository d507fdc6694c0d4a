"""pytest configuration: marker registration + import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`      : parity of the HIP path against the oracle / goldens, through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vox-e_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Image-ordered renders below 8192 rays take the scatter backward by default (faster on an empty chip); the parity
# tests use small images, so they lower the threshold to keep exercising the LDS-window backward.  The scatter
# backward is covered by the unordered-ray cases (tests/test_hip_fuzz.py, bench --ray-order random).
os.environ.setdefault("VOXE_TILE_MIN_RAYS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _seed():
    np.random.seed(42)
    try:
        import torch

        torch.manual_seed(42)
    except ImportError:  # pragma: no cover
        pass
