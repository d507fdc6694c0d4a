"""Loader + thin call helpers for libvoxe_hip.so (ctypes over the C ABI of include/voxe.h)."""
import ctypes as C
import os
import threading

import torch

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# VOXE_HIP_LIB: load another build of the same library (kernel-variant A/B runs, tools/variants.py)
LIB_PATH = os.environ.get("VOXE_HIP_LIB") or os.path.join(_HERE, "libvoxe_hip.so")

_lib = None
_lock = threading.Lock()


class VoxeError(RuntimeError):
    pass


def lib():
    """The loaded HIP library.  Fails loudly (no fallback) when the extension was not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise VoxeError(
                        f"{LIB_PATH} is missing: build the HIP extension first "
                        f"(python __graft_entry__.py build  or  python vox-e_amd/voxe_hip/build.py)"
                    )
                # torch is imported above, so its bundled libamdhip64 (SONAME libamdhip64.so.7) is
                # already mapped and the dynamic loader binds our NEEDED entry to that same runtime.
                handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
                abi.declare(handle, "voxe_")
                ver = handle.voxe_abi_version()
                if ver != abi.ABI_VERSION:
                    raise VoxeError(f"libvoxe_hip.so ABI {ver} != binding ABI {abi.ABI_VERSION}")
                _lib = handle
    return _lib


_DEBUG_SYNC = bool(os.environ.get("VOXE_DEBUG_SYNC"))   # debugging aid: synchronise after every library call, so that an
                                                        # asynchronous device fault surfaces at the call that caused it


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().voxe_strerror(status).decode()
        raise VoxeError(f"{what}: {msg} (status {status})")
    if _DEBUG_SYNC:
        try:
            torch.cuda.synchronize()
        except RuntimeError as e:
            raise VoxeError(f"{what}: device fault surfaced at the synchronisation after this call: {e}") from e


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise VoxeError(
            f"{what}: tensor is on {t.device}; the voxe HIP path only runs on a ROCm GPU "
            f"(there is no CPU fallback in the product path)"
        )


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def f32c(t: torch.Tensor) -> torch.Tensor:
    """contiguous float32 view/copy (the ABI takes dense float32 only)"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_device_checked = set()


def ensure_gfx950(device) -> None:
    if torch.device(device).type != "cuda":
        raise VoxeError(f"the voxe HIP path only runs on a ROCm GPU, not on {torch.device(device)} (there is no CPU fallback in "
                        f"the product path)")
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx in _device_checked:
        return
    with torch.cuda.device(idx):
        buf = C.create_string_buffer(64)
        st = lib().voxe_device_check(buf, 64)
        if st != 0:
            raise VoxeError(f"device {idx} is '{buf.value.decode()}', libvoxe_hip.so holds gfx950 code only")
    _device_checked.add(idx)
