"""Ray / render-output containers of the volumetric render interface.

Same public types as the reference's thre3d_atom/rendering/volumetric/render_interface.py
(`Rays` :13-44, `RenderOut` :47-83, `RenderOutAttn` :85-121, `SampledPointsOnRays` :123-131).
The reference's three-stage driver `render(rays, bounds, S, sampler_fn, processor_fn, accumulator_fn)`
(:140-171) has no counterpart here on purpose: the three stages are ONE fused HIP kernel
(see thre3d_atom.thre3d_reprs.renderers.render_sh_voxel_grid).
"""
import dataclasses
from typing import Any, Callable, Dict, NamedTuple, Optional, Tuple

import torch
from torch import Tensor

from thre3d_atom.utils.constants import NUM_ATTN_CHANNELS, NUM_COLOUR_CHANNELS, NUM_COORD_DIMENSIONS
from thre3d_atom.utils.imaging_utils import CameraBounds

ExtraInfo = Dict[str, Any]


@dataclasses.dataclass
class Rays:
    origins: Tensor  # [..., 3]
    directions: Tensor  # [..., 3]
    # (height, width) when the rays are the row-major pixels of one image (set by cast_rays, kept by
    # flatten_rays, dropped by slicing).  Lets the HIP kernels walk 2-D pixel tiles.  Not in the reference.
    image_shape: Optional[Tuple[int, int]] = None

    def __post_init__(self):
        if self.origins.shape != self.directions.shape:
            raise AssertionError("ray-origins and ray-directions are incompatible :(")
        if self.origins.shape[-1] != NUM_COORD_DIMENSIONS:
            raise AssertionError("only 3D coordinate spaces are supported: cast your rays in 3 dimensions")

    def __getitem__(self, item) -> "Rays":
        return Rays(origins=self.origins[item, :], directions=self.directions[item, :])

    def __len__(self) -> int:
        return len(self.origins)

    def to(self, device: torch.device) -> "Rays":
        return Rays(self.origins.to(device), self.directions.to(device), self.image_shape)


def _map_extra(extra: ExtraInfo, fn: Callable[[Tensor], Tensor]) -> ExtraInfo:
    return {key: fn(value) for key, value in extra.items()}


class _RenderOutBase:
    """detach()/to() shared by the two output records (main image field named by `_main`)."""

    _main = "colour"

    def _rebuild(self, fn):
        kwargs = {self._main: fn(getattr(self, self._main)), "depth": fn(self.depth), "extra": _map_extra(self.extra, fn)}
        return type(self)(**kwargs)

    def detach(self):
        return self._rebuild(lambda t: t.detach())

    def to(self, device: torch.device):
        return self._rebuild(lambda t: t.to(device))

    def _check(self, channels: int):
        main = getattr(self, self._main)
        if main.shape[:-1] != self.depth.shape[:-1]:
            raise AssertionError("rendered maps and depth maps are shape-incompatible")
        if main.shape[-1] != channels:
            raise AssertionError(f"rendered map must have {channels} channel(s), got {main.shape[-1]}")
        if self.depth.shape[-1] != 1:
            raise AssertionError("depth map must have exactly 1 channel")
        if self.extra is None:
            self.extra = {}


@dataclasses.dataclass
class RenderOut(_RenderOutBase):
    colour: Tensor  # [..., 3]
    depth: Tensor  # [..., 1]
    extra: Optional[ExtraInfo] = None
    _main = "colour"

    def __post_init__(self):
        self._check(NUM_COLOUR_CHANNELS)


@dataclasses.dataclass
class RenderOutAttn(_RenderOutBase):
    attn: Tensor  # [..., 1]
    depth: Tensor  # [..., 1]
    extra: Optional[ExtraInfo] = None
    _main = "attn"

    def __post_init__(self):
        self._check(NUM_ATTN_CHANNELS)


class SampledPointsOnRays(NamedTuple):
    points: Tensor  # [N, num_samples, 3]
    depths: Tensor  # [N, num_samples]


ProcessedPointsOnRays = SampledPointsOnRays

RaySamplerFunction = Callable[[Rays, CameraBounds, int], SampledPointsOnRays]
PointProcessorFunction = Callable[[SampledPointsOnRays, Rays], ProcessedPointsOnRays]
AccumulatorFunction = Callable[[ProcessedPointsOnRays, Rays], RenderOut]
