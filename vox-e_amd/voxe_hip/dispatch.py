"""Which kernels render a call, and with which tuning parameters: the Python face of `VoxeDispatch` (include/voxe.h, ABI v8).

The library itself holds no dispatch state and reads no environment on the render path: every render call carries a pointer
to one of these structs (`VoxeRenderCfg::dispatch`; NULL = the shipped dispatch).  This module

  * resolves the process's VOXE_* environment switches ONCE, on first use, into a `Dispatch` (`from_env()`): the A/B shell
    scripts under tools/ (`ab_env.sh`, `ab_env_bench.sh`: `VOXE_TILE_KL=10 python bench.py ...`) set them before the process
    starts; changing the environment afterwards has no effect and nothing is re-read per launch;
  * lets a caller ask for another dispatch per call (`RenderParams(dispatch=Dispatch(...))`) or for a scope
    (`with dispatch.override(region_min_rays=1): ...`), e.g. parity tests that must reach the LDS-window backward with a
    48x48 image.

Forward and backward of one render have to use the same dispatch; `ops` keys the per-ray states on it.
"""
import contextlib
import contextvars
import dataclasses
import functools
import os
from dataclasses import dataclass
from typing import Optional

from . import abi


@dataclass(frozen=True)
class Dispatch:
    """field for field `VoxeDispatch`; 0 = the shipped default everywhere"""
    bwd_mode: int = 0                # 0 auto | 1 plain global-atomic scatter | 2 line-dense scatter
    tile_map: int = 0                # 0 auto | 1 interleave | 2 band | 3 rows
    tile_min_rays: int = 0           # 0 = 8192 | > 0 | -1: no minimum (small images through the LDS-window backward)
    tile_two_phase: int = 0          # 0 default | -1 single-kernel channel groups
    tile_qsplit: int = 0             # 0 auto | 1 | 4
    tile_kl: int = 0                 # 0 auto | 8 | 10
    tile_fit_m: float = 0.0
    tile_fit_lat: float = 0.0
    fwd_window: int = 0              # 0 on | -1 off
    fwd_fit_lat: float = 0.0
    fwd_fit_m: float = 0.0
    fwd_zdom: float = 0.0            # 0 = 1.0 (SH-0) / -1.0 (SH 1-3) | < 0: z-dominant tiles through the window as well
    fwd_max_adv: float = 0.0
    fwd_segments_per_thread: int = 0
    region_min_rays: int = 0         # 0 = 16384 | > 0 | -1: route off
    region_image_ratio: float = 0.0  # 0 = 1.3 | < 0: every image-ordered launch the route accepts
    tile_lean: int = 0               # 0 lean SH-0 tile backward where it applies | -1 always the general kernel
    precise_grad: int = 0            # 0 float in-segment suffix sums | 1 double (image-ordered SH-0 backward)
    region_lds_ranks: int = 0        # 0 block-local LDS ranks in the space-binned segment pass | -1 global returning atomics
    tile_phases: int = 0             # 0 split tiles run 2 / 4 sample phases per ray (full waves) | -1 one sample per ray, idle lanes

    def struct(self) -> abi.VoxeDispatch:
        return _struct_of(self)


@functools.lru_cache(maxsize=None)
def _struct_of(d: Dispatch) -> abi.VoxeDispatch:
    s = abi.VoxeDispatch()
    for f in dataclasses.fields(d):
        setattr(s, f.name, getattr(d, f.name))
    return s      # cached for the life of the process: render calls hold raw pointers to it


SHIPPED = Dispatch()
# parity tests / tools: image-ordered renders of any size through the LDS-window (tile) backward
TILE_ALWAYS = Dispatch(tile_min_rays=-1)


def _env_int(name):
    v = os.environ.get(name)
    return None if v is None or v == "" else int(v)


def _env_float(name):
    v = os.environ.get(name)
    return None if v is None or v == "" else float(v)


@functools.lru_cache(maxsize=1)
def from_env() -> Dispatch:
    """the environment's VOXE_* dispatch switches as a Dispatch -- evaluated once per process"""
    kw = {}
    mode = os.environ.get("VOXE_BWD_MODE")
    if mode == "scatter":
        kw["bwd_mode"] = 1
    elif mode == "packed":
        kw["bwd_mode"] = 2
    tmap = {"interleave": 1, "band": 2, "rows": 3}.get(os.environ.get("VOXE_TILE_MAP", ""))
    if tmap:
        kw["tile_map"] = tmap
    v = _env_int("VOXE_TILE_MIN_RAYS")
    if v is not None:
        kw["tile_min_rays"] = -1 if v <= 0 else v
    if os.environ.get("VOXE_TILE_TWO_PHASE", "")[:1] == "0":
        kw["tile_two_phase"] = -1
    v = _env_int("VOXE_TILE_QSPLIT")
    if v:
        kw["tile_qsplit"] = 4 if v == 4 else 1
    v = _env_int("VOXE_TILE_KL")
    if v:
        kw["tile_kl"] = v
    for env, field in (("VOXE_TILE_FIT_M", "tile_fit_m"), ("VOXE_TILE_FIT_LAT", "tile_fit_lat"),
                       ("VOXE_FWD_TILE_FIT_LAT", "fwd_fit_lat"), ("VOXE_FWD_TILE_FIT_M", "fwd_fit_m"),
                       ("VOXE_FWD_TILE_ZDOM", "fwd_zdom"), ("VOXE_FWD_TILE_ADV", "fwd_max_adv")):
        f = _env_float(env)
        if f:
            kw[field] = f
    if os.environ.get("VOXE_FWD_TILE", "")[:1] == "0":
        kw["fwd_window"] = -1
    v = _env_int("VOXE_FSEG")
    if v and v > 0:
        kw["fwd_segments_per_thread"] = v
    v = _env_int("VOXE_REGION_MIN_RAYS")
    if v is not None:
        kw["region_min_rays"] = -1 if v < 0 else max(v, 1)
    f = _env_float("VOXE_REGION_IMAGE_RATIO")
    if f is not None:
        kw["region_image_ratio"] = -1.0 if f <= 0.0 else f
    if os.environ.get("VOXE_TILE_LEAN", "")[:1] == "0":
        kw["tile_lean"] = -1
    if os.environ.get("VOXE_PRECISE_GRAD", "")[:1] == "1":
        kw["precise_grad"] = 1
    if os.environ.get("VOXE_REGION_LDS_RANKS", "")[:1] == "0":
        kw["region_lds_ranks"] = -1
    if os.environ.get("VOXE_TILE_PHASES", "")[:1] == "0":
        kw["tile_phases"] = -1
    return Dispatch(**kw)


# the override of the CURRENT context (thread / asyncio task): a render issued from another thread -- a DataLoader or feedback
# thread -- inside someone else's override() block keeps the process dispatch
_override: contextvars.ContextVar = contextvars.ContextVar("voxe_dispatch_override", default=None)


def current() -> Dispatch:
    """the dispatch a render call uses when its RenderParams name none"""
    o = _override.get()
    return o if o is not None else from_env()


@contextlib.contextmanager
def override(base: Optional[Dispatch] = None, **fields):
    """`with override(region_min_rays=1):` -- renders inside the block default to the current dispatch with these fields
    replaced (or to `base`); Python-side state only, handed to the library call by call"""
    new = dataclasses.replace(base if base is not None else current(), **fields)
    token = _override.set(new)
    try:
        yield new
    finally:
        _override.reset(token)
