"""Render procedures of the SH voxel grid -- the drop-in boundary.

`RenderProcedure = Callable[[Module, Rays, RenderConfig, Optional[int]], RenderOut]` and
`SHVoxGridRenderConfig` keep the reference's contract (thre3d_atom/thre3d_reprs/renderers.py:23-47)
so `VolumetricModel` and checkpoints are interchangeable.  Where the reference binds a sampler, a point
processor and an accumulator with functools.partial and runs ~40 ATen ops under autograd
(renderers.py:50-163), these procedures translate the config into one VoxeRenderCfg and call the fused
HIP forward kernel; autograd sees a single node whose backward is the fused HIP backward kernel.
"""
import dataclasses
from typing import Any, Callable, Optional

import torch
from torch import Tensor
from torch.nn import Module

from thre3d_atom.rendering.volumetric.accumulate import density2occupancy_pb
from thre3d_atom.rendering.volumetric.render_interface import Rays, RenderOut, RenderOutAttn
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid
from thre3d_atom.utils.constants import EXTRA_ACCUMULATED_WEIGHTS, EXTRA_DISPARITY, NUM_COLOUR_CHANNELS
from thre3d_atom.utils.imaging_utils import CameraBounds
from voxe_hip import ops as _ops
from voxe_hip.runtime import VoxeError

RenderConfig = Any
RenderProcedure = Callable[[Module, Rays, RenderConfig, Optional[int]], RenderOut]

# Gradient truncation is NOT part of the reference (it differentiates all S samples).  0.0 keeps exact reference semantics.
_TERM_EPS = 0.0


def set_gradient_truncation(eps: float) -> None:
    """gradient truncation of the HIP renderer (VoxeRenderCfg::term_eps; not in the reference, 0 = off): the BACKWARD stops
    marching a ray once its transmittance is below `eps` -- the samples in front keep their exact gradient, the samples behind
    get none (a biased gradient, by at most eps of the ray's weight); forward outputs are never affected.  While it is on,
    unordered batches of >= 16384 rays do not take the space-binned route (its backward has no truncation): they render
    through the line-dense scatter, which is slower for such batches -- switch it on for image-ordered renders (SDS steps)."""
    global _TERM_EPS
    _TERM_EPS = float(eps)


def set_early_termination(eps: float) -> None:
    """deprecated name of `set_gradient_truncation` (until r02 the switch also cut the FORWARD march short; since r03 the
    forward always integrates every sample like the reference and only the gradient is truncated)"""
    import warnings

    warnings.warn("set_early_termination() is now set_gradient_truncation(): the forward is always exact, only the backward "
                  "is truncated", DeprecationWarning, stacklevel=2)
    set_gradient_truncation(eps)


@dataclasses.dataclass
class SHVoxGridRenderConfig:
    # probing
    num_samples_per_ray: int
    camera_bounds: CameraBounds
    perturb_sampled_points: bool = True
    optimized_sampling: bool = False
    linear_disparity_sampling: bool = False
    # accumulation
    density2occupancy: Callable[[Tensor, Tensor], Tensor] = density2occupancy_pb
    radiance_hdr_tone_map: Callable[[Tensor], Tensor] = torch.sigmoid
    stochastic_density_noise_std: float = 0.0
    white_bkgd: bool = False
    # render modes
    render_diffuse: bool = False
    render_num_samples_per_ray: int = 1024
    parallel_rays_chunk_size: int = 32768


def _sh_degree_of(voxel_grid: VoxelGrid) -> int:
    per_channel = voxel_grid.features.shape[-1] / NUM_COLOUR_CHANNELS
    degree = int(round(per_channel ** 0.5)) - 1
    if NUM_COLOUR_CHANNELS * (degree + 1) ** 2 != voxel_grid.features.shape[-1] or not 0 <= degree <= 3:
        raise VoxeError(f"feature channels {voxel_grid.features.shape[-1]} are not 3*(deg+1)^2 for deg in 0..3")
    return degree


def _render_params(voxel_grid: VoxelGrid, rays: Optional[Rays], cfg: SHVoxGridRenderConfig, attn: bool) -> _ops.RenderParams:
    if cfg.density2occupancy is not density2occupancy_pb:
        raise VoxeError("only density2occupancy_pb has a HIP path")
    if cfg.radiance_hdr_tone_map is not torch.sigmoid:
        raise VoxeError("only torch.sigmoid tone mapping has a HIP path")
    if cfg.stochastic_density_noise_std != 0.0:
        # accumulate.py:57-63 adds the noise to every sample, the last one (delta 1e10) included: the reference itself renders
        # NaN for half of all rays of a voxel grid at any std (tools/ref_density_noise_demo.py) and never sets it
        raise VoxeError("stochastic_density_noise_std != 0 is not supported by the HIP renderer (the reference renders NaN "
                        "for every ray whose last sample draws negative noise: profiles/r05_density_noise_reference.txt)")
    num_rays = rays.origins.shape[0] if rays is not None else 0     # (rays None: parameters of an unordered batch)
    # image-ordered rays: one image (R == H * W) or a multi-view batch of K images of that shape, one after the other
    # (collate_rays of flattened cameras that all carry the same image_shape): ONE launch, 2-D pixel tiles per camera
    width = height = 0
    if rays is not None and rays.image_shape is not None and num_rays > 0:
        per_image = int(rays.image_shape[0]) * int(rays.image_shape[1])
        if per_image > 0 and num_rays % per_image == 0:
            width = int(rays.image_shape[1])
            height = int(rays.image_shape[0]) if num_rays != per_image else 0
    return _ops.RenderParams(
        num_samples=int(cfg.num_samples_per_ray),
        near=float(cfg.camera_bounds[0]),
        far=float(cfg.camera_bounds[1]),
        perturb=bool(cfg.perturb_sampled_points),
        # the attention procedure of the reference does not forward linear_disparity (renderers.py:131-134)
        linear_disparity=bool(cfg.linear_disparity_sampling) and not attn,
        aabb_clip=bool(cfg.optimized_sampling),
        white_bkgd=bool(cfg.white_bkgd),
        sh_degree=0 if attn else _sh_degree_of(voxel_grid),
        render_diffuse=bool(cfg.render_diffuse),
        term_eps=_TERM_EPS,
        image_width=width,
        image_height=height,
    )


def attn_render_params(voxel_grid: VoxelGrid, rays: Rays, render_config: SHVoxGridRenderConfig) -> _ops.RenderParams:
    """the kernel parameters `render_sh_voxel_grid_attn` renders (voxel_grid, rays, render_config) with -- for callers that hand a
    whole refinement iteration to the library (FusedGridAdam.attention_refinement_step)"""
    _check_flat(rays)
    return _render_params(voxel_grid, rays, render_config, attn=True)


def _check_flat(rays: Rays) -> None:
    if rays.origins.dim() != 2 or rays.directions.dim() != 2:
        raise AssertionError("Please note that the RENDER interface only works with FLAT RAYS!")


def render_sh_voxel_grid(
    voxel_grid: VoxelGrid,
    rays: Rays,
    render_config: SHVoxGridRenderConfig,
    parallel_points_chunk_size: Optional[int] = None,
) -> RenderOut:
    """Render flat rays through an SH voxel grid.  `parallel_points_chunk_size` is accepted for API
    parity and ignored: the fused kernel has no per-point temporaries to chunk."""
    _check_flat(rays)
    params = _render_params(voxel_grid, rays, render_config, attn=False)
    colour, depth, acc, disparity = _ops.render(
        voxel_grid.voxe_grid_spec(attn=False), params, voxel_grid.densities, voxel_grid.features,
        rays.origins, rays.directions, workspace=voxel_grid.voxe_workspace("sh"),
    )
    return RenderOut(colour=colour, depth=depth, extra={EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc})


def render_sh_voxel_grid_attn(
    voxel_grid: VoxelGrid,
    rays: Rays,
    render_config: SHVoxGridRenderConfig,
    parallel_points_chunk_size: Optional[int] = None,
    orig_densities=False,
) -> RenderOutAttn:
    """Render the 1-channel attention grid (VoxelGrid.attn) with the grid's densities (or the detached
    `orig_densities` snapshot); background contributes 0 (accumulate.py:166)."""
    _check_flat(rays)
    if voxel_grid.attn is None:
        raise VoxeError("this VoxelGrid has no attention grid (attn is None)")
    params = _render_params(voxel_grid, rays, render_config, attn=True)
    densities = voxel_grid.densities
    if orig_densities:
        densities = voxel_grid.orig_densities.detach().to(voxel_grid.attn.device)
    attn, depth, acc, disparity = _ops.render(
        voxel_grid.voxe_grid_spec(attn=True), params, densities, voxel_grid.attn,
        rays.origins, rays.directions, workspace=voxel_grid.voxe_workspace("attn"),
    )
    return RenderOutAttn(attn=attn, depth=depth, extra={EXTRA_DISPARITY: disparity, EXTRA_ACCUMULATED_WEIGHTS: acc})
