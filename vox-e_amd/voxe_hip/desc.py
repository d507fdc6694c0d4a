"""Builders for the two C-ABI descriptor structs (pure numpy/ctypes, no torch)."""
from typing import Optional, Sequence, Tuple

import numpy as np

from . import abi


def norm_constants(aabb: Sequence[Tuple[float, float]]):
    """scale/bias of VoxelGrid._normalize_points.

    Mirrors thre3d_atom/utils/imaging_utils.py:57-63 (adjust_dynamic_range, slack=True,
    drange_out=(-1, 1)) evaluated per axis as in thre3d_atom/thre3d_reprs/voxels.py:225-234:
    all scalars are np.float32, so the rounding of scale and bias is part of the contract.
    """
    scale, bias = [], []
    for lo, hi in aabb:
        s = (np.float32(1.0) - np.float32(-1.0)) / (np.float32(hi) - np.float32(lo))
        b = np.float32(-1.0) - np.float32(lo) * s
        scale.append(np.float32(s))
        bias.append(np.float32(b))
    return scale, bias


def make_grid_desc(
    densities_ptr: int,
    features_ptr: int,
    dims: Tuple[int, int, int],
    num_features: int,
    aabb: Sequence[Tuple[float, float]],
    density_scale: float,
    density_pre_act: int,
    density_post_act: int,
    feature_kind: int = abi.FEAT_SH,
) -> abi.VoxeGridDesc:
    g = abi.VoxeGridDesc()
    g.densities = densities_ptr
    g.features = features_ptr
    g.X, g.Y, g.Z = (int(d) for d in dims)
    g.F = int(num_features)
    scale, bias = norm_constants(aabb)
    for a in range(3):
        # comparisons `points > aabb.range[0]` are evaluated in float32 (voxels.py:263-285)
        g.aabb_lo[a] = np.float32(aabb[a][0])
        g.aabb_hi[a] = np.float32(aabb[a][1])
        g.norm_scale[a] = scale[a]
        g.norm_bias[a] = bias[a]
    g.density_scale = np.float32(density_scale)
    g.density_pre_act = int(density_pre_act)
    g.density_post_act = int(density_post_act)
    g.feature_kind = int(feature_kind)
    return g


def make_render_cfg(
    num_samples: int,
    near: float,
    far: float,
    perturb: bool = False,
    linear_disparity: bool = False,
    aabb_clip: bool = False,
    white_bkgd: bool = False,
    sh_degree: int = 0,
    render_diffuse: bool = False,
    term_eps: float = 0.0,
    seed: int = 0,
    rng_offset: int = 0,
    reuse_packed_grid: bool = False,
    image_width: Optional[int] = None,
    ray_state_valid: bool = False,
    image_height: Optional[int] = None,
    deterministic: bool = False,
    linear_grad: bool = False,
    dispatch: Optional[abi.VoxeDispatch] = None,
) -> abi.VoxeRenderCfg:
    c = abi.VoxeRenderCfg()
    c.num_samples = int(num_samples)
    c.near = np.float32(near)
    c.far = np.float32(far)
    c.perturb = int(bool(perturb))
    c.linear_disparity = int(bool(linear_disparity))
    c.aabb_clip = int(bool(aabb_clip))
    c.white_bkgd = int(bool(white_bkgd))
    c.sh_degree = int(sh_degree)
    c.render_diffuse = int(bool(render_diffuse))
    c.term_eps = np.float32(term_eps)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.rng_offset = int(rng_offset) & 0xFFFFFFFFFFFFFFFF
    c.reuse_packed_grid = int(bool(reuse_packed_grid))
    c.image_width = int(image_width or 0)
    c.image_height = int(image_height or 0)
    c.deterministic = int(bool(deterministic))
    c.linear_grad = int(bool(linear_grad))
    c.ray_state_valid = int(bool(ray_state_valid))
    if dispatch is not None:
        import ctypes

        c.dispatch = ctypes.pointer(dispatch)
        c._dispatch_keepalive = dispatch   # (the struct must outlive every call that is handed this cfg)
    return c
