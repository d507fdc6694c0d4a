"""Ray casting / batching helpers around the renderer.

API of the reference's thre3d_atom/rendering/volumetric/utils/misc.py; `cast_rays` runs the HIP
ray-generation kernel (voxe_cast_rays), everything else is tensor bookkeeping.
"""
from typing import Any, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from thre3d_atom.rendering.volumetric.render_interface import Rays, RenderOut, RenderOutAttn
from thre3d_atom.utils.constants import NUM_COORD_DIMENSIONS
from thre3d_atom.utils.imaging_utils import CameraIntrinsics, CameraPose
from voxe_hip import ops as _ops


def cast_rays(camera_intrinsics: CameraIntrinsics, pose: CameraPose, device: torch.device = torch.device("cuda")) -> Rays:
    """Pin-hole rays of one camera, shaped [H, W, 3] (pixel centres at +0.5, camera looks down -z,
    directions NOT normalised; misc.py:12-50).  Computed on the GPU by voxe_cast_rays."""
    height, width, focal = camera_intrinsics
    rot, trans = pose.rotation, pose.translation
    if not isinstance(rot, Tensor):
        rot, trans = torch.from_numpy(np.asarray(rot)), torch.from_numpy(np.asarray(trans))
    origins, directions = _ops.cast_rays(height, width, focal, rot, trans, device)
    shape = (int(height), int(width), NUM_COORD_DIMENSIONS)
    return Rays(origins.view(shape), directions.view(shape), image_shape=(int(height), int(width)))


def flatten_rays(rays: Rays) -> Rays:
    return Rays(
        origins=rays.origins.reshape(-1, NUM_COORD_DIMENSIONS),
        directions=rays.directions.reshape(-1, NUM_COORD_DIMENSIONS),
        image_shape=rays.image_shape,
    )


def collate_rays(rays_list: Sequence[Rays]) -> Rays:
    """Concatenation of flat ray batches (misc.py:60-70).  When every batch is one whole image of the same shape the
    result keeps that `image_shape`: the renderer then runs the K cameras as ONE image-ordered launch."""
    shapes = {r.image_shape for r in rays_list}
    shape = shapes.pop() if len(shapes) == 1 else None
    if shape is not None and any(r.origins.dim() != 2 or r.origins.shape[0] != shape[0] * shape[1] for r in rays_list):
        shape = None
    return Rays(
        origins=torch.cat([r.origins for r in rays_list], dim=0),
        directions=torch.cat([r.directions for r in rays_list], dim=0),
        image_shape=shape,
    )


def collate_rays_unflattened(rays_list: Sequence[Rays]) -> Rays:
    return Rays(
        origins=torch.stack([r.origins for r in rays_list], dim=0),
        directions=torch.stack([r.directions for r in rays_list], dim=0),
    )


def compute_expected_density_scale_for_relu_field_grid(grid_world_size: Tuple[float, float, float]) -> float:
    """(sqrt(27) * 100 / |diag|) / 3 : 33.33 for a 3^3 world (misc.py:77-87)."""
    diagonal = float(np.sqrt(np.sum([extent ** 2 for extent in grid_world_size])))
    return ((float(np.sqrt(3.0 ** 3)) * 100.0) / diagonal) / NUM_COORD_DIMENSIONS


def _select(rays: Rays, pixels: Tensor, sample_size: int):
    subset = torch.randperm(pixels.shape[0], dtype=torch.long, device=pixels.device)[:sample_size]
    return subset, Rays(rays.origins[subset, :], rays.directions[subset, :]), pixels[subset, :]


def sample_random_rays_and_pixels_synchronously(rays: Rays, pixels: Tensor, sample_size: int) -> Tuple[Rays, Tensor]:
    """The same random subset of rays and of their target pixels (misc.py:126-138)."""
    _, picked_rays, picked_pixels = _select(rays, pixels, sample_size)
    return picked_rays, picked_pixels


def sample_random_rays_and_pixels_from_cameras(camera_intrinsics: CameraIntrinsics, poses: Tensor, images: Tensor,
                                               sample_size: int, image_ids: Any = None,
                                               memory_order: bool = False, fast_subset: bool = False) -> Tuple[Rays, Tensor]:
    """What the reconstruction loop keeps of `cast_rays` per camera -> `collate_rays` -> pixel concat ->
    `sample_random_rays_and_pixels_synchronously` (modules/trainers.py:290-313), computed for the selected pixels
    only: the same `randperm` draw picks flat (camera, y, x) indices, the HIP kernel casts just those rays and the
    target pixels are gathered straight from `images`.  Bit-identical rays / pixels, no full-image ray buffers, no
    host synchronisation.   poses [K,3,4] and images [N,C,H,W] on the GPU; `image_ids` [K] maps the K cameras to
    rows of `images` (default: the first K).  `memory_order=True` returns the same random subset sorted by (camera,
    row, column) instead of in draw order: the batch (a set; the loss is a mean over it) is unchanged, but rays that
    are neighbours in the batch then walk neighbouring voxels, which the forward gather rewards (-25 % at 32768 rays).
    `fast_subset=True` draws the subset with voxe_random_subset (same distribution: a uniformly random set of distinct
    pixels; 0.02 ms instead of 0.20 ms for `randperm` over 1.28 M pixels) -- a different random stream than torch's."""
    height, width, focal = camera_intrinsics
    K = int(poses.shape[0])
    per = int(height) * int(width)
    if fast_subset:  # a keyed Feistel permutation evaluated at sample_size points instead of a permutation of all pixels
        subset = _ops.random_subset(K * per, min(int(sample_size), K * per), images.device)
    else:
        subset = torch.randperm(K * per, dtype=torch.long, device=images.device)[:sample_size]
    if memory_order:
        subset = torch.sort(subset).values
    origins, directions = _ops.cast_rays_indexed(height, width, focal, poses, subset)
    cam = torch.div(subset, per, rounding_mode="floor")
    rem = subset - cam * per
    rows = cam if image_ids is None else torch.as_tensor(image_ids, device=images.device, dtype=torch.long)[cam]
    pixels = images[rows, :, torch.div(rem, int(width), rounding_mode="floor"), rem % int(width)]
    return Rays(origins, directions), pixels


def sample_rays_and_pixels_synchronously(rays: Rays, pixels: Tensor, indices: Any, sample_size: int):
    """Image-level variant: rays [B,H,W,3], pixels [B,C,H,W]; returns flat rays/pixels of the picked
    images plus their dataset indices (misc.py:140-158)."""
    subset, picked_rays, picked_pixels = _select(rays, pixels, sample_size)
    picked_indices = indices[subset.to("cpu")]
    if sample_size == 1:
        picked_indices = [picked_indices]
    flat_pixels = picked_pixels.permute(0, 2, 3, 1).reshape(-1, pixels.shape[1])
    return flatten_rays(picked_rays), flat_pixels, picked_indices, subset.tolist()


def _collate(chunks, main: str, cls):
    extra_keys = chunks[0].extra.keys() if chunks else ()
    return cls(
        **{main: torch.cat([getattr(c, main) for c in chunks], dim=0)},
        depth=torch.cat([c.depth for c in chunks], dim=0),
        extra={k: torch.cat([c.extra[k] for c in chunks], dim=0) for k in extra_keys},
    )


def _reshape(out, main: str, cls, camera_intrinsics: CameraIntrinsics):
    shape = (camera_intrinsics.height, camera_intrinsics.width, -1)
    return cls(
        **{main: getattr(out, main).reshape(*shape)},
        depth=out.depth.reshape(*shape),
        extra={k: v.reshape(*shape) for k, v in out.extra.items()},
    )


def collate_rendered_output(rendered_chunks: Sequence[RenderOut]) -> RenderOut:
    return _collate(rendered_chunks, "colour", RenderOut)


def collate_rendered_output_attn(rendered_chunks: Sequence[RenderOutAttn]) -> RenderOutAttn:
    return _collate(rendered_chunks, "attn", RenderOutAttn)


def reshape_rendered_output(rendered_output: RenderOut, camera_intrinsics: CameraIntrinsics) -> RenderOut:
    return _reshape(rendered_output, "colour", RenderOut, camera_intrinsics)


def reshape_rendered_output_attn(rendered_output: RenderOutAttn, camera_intrinsics: CameraIntrinsics) -> RenderOutAttn:
    return _reshape(rendered_output, "attn", RenderOutAttn, camera_intrinsics)
