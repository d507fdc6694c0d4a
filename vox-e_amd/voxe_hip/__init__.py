"""Python binding of libvoxe_hip.so -- the MI355X (gfx950) voxel-grid renderer behind the C ABI of
include/voxe.h.  `abi`/`desc` are torch-free; `runtime`/`ops` plug the library into PyTorch-ROCm
(device memory, streams, autograd).  There is NO CPU fallback: every op raises if the HIP library
or a gfx950 device is missing."""
from . import abi, desc  # noqa: F401

__all__ = ["abi", "desc"]
