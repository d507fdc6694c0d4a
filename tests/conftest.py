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

# The suite runs under the SHIPPED kernel dispatch (voxe_hip.dispatch.SHIPPED: e.g. image-ordered renders below 8192 rays take
# the line-dense scatter backward).  Tests that are about one particular route ask for it through the C ABI, call by call
# (VoxeRenderCfg::dispatch): the `disp` fixture replaces fields of the dispatch the helpers hand to every render call of ONE
# test; modules whose small images are meant to exercise the LDS-window (tile) backward use the `tile_always` fixture.
class _DispatchPatch:
    def __init__(self):
        from voxe_hip import dispatch

        self._mod = dispatch
        self._saved = dispatch._override.get()

    def set(self, **fields):
        import dataclasses

        self._mod._override.set(dataclasses.replace(self._mod.current(), **fields))

    def restore(self):
        self._mod._override.set(self._saved)


@pytest.fixture
def disp():
    p = _DispatchPatch()
    yield p
    p.restore()


@pytest.fixture
def tile_always(disp):
    """image-ordered renders of any size through the LDS-window backward (VoxeDispatch::tile_min_rays = -1)"""
    disp.set(tile_min_rays=-1)
    yield disp


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
