"""VoxelGrid: the trainable density/feature volume rendered by the fused HIP kernels.

Public surface of the reference's thre3d_atom/thre3d_reprs/voxels.py (`VoxelSize` :19,
`VoxelGridLocation` :28, `AxisAlignedBoundingBox` :38, `VoxelGrid` :46, scaling :409-488,
(de)serialisation :491-517).  The module only *holds* the tensors and the grid geometry; sampling it
along rays is done by voxe_hip (render_sh_voxel_grid), never by torch ops.
"""
from typing import Any, Callable, Dict, NamedTuple, Optional, Tuple

import torch
from torch import Tensor
from torch.nn import Module

from thre3d_atom.thre3d_reprs.constants import CONFIG_DICT, STATE_DICT, THRE3D_REPR, u_ATTN, u_DENSITIES, u_FEATURES
from voxe_hip import abi
from voxe_hip import ops as _ops
from voxe_hip.runtime import VoxeError


class VoxelSize(NamedTuple):
    """edge lengths of one voxel (anisotropic voxels allowed)"""
    x_size: float = 1.0
    y_size: float = 1.0
    z_size: float = 1.0


class VoxelGridLocation(NamedTuple):
    """world position of the grid centre (the grid is axis aligned)"""
    x_coord: float = 0.0
    y_coord: float = 0.0
    z_coord: float = 0.0


class AxisAlignedBoundingBox(NamedTuple):
    x_range: Tuple[float, float]
    y_range: Tuple[float, float]
    z_range: Tuple[float, float]


def _is_identity(fn) -> bool:
    return isinstance(fn, torch.nn.Identity)


def density_activation_codes(pre: Callable, post: Callable) -> Tuple[int, int]:
    """Map the (pre, post) density activation callables onto the kernel's enum (include/voxe.h VoxeAct).
    Anything the HIP path does not implement raises -- there is no torch fallback."""
    if _is_identity(pre):
        pre_code = abi.ACT_IDENTITY
    elif pre is torch.abs or isinstance(pre, type(torch.abs)) and getattr(pre, "__name__", "") == "abs":
        pre_code = abi.ACT_ABS
    else:
        raise VoxeError(f"density_preactivation {pre!r} is not supported by the HIP renderer (Identity | torch.abs)")
    if _is_identity(post):
        post_code = abi.ACT_IDENTITY
    elif isinstance(post, torch.nn.ReLU) or post is torch.relu or post is torch.nn.functional.relu:
        post_code = abi.ACT_RELU
    elif isinstance(post, torch.nn.Softplus):
        if post.beta != 1 or post.threshold != 20:
            raise VoxeError("only Softplus(beta=1, threshold=20) is supported by the HIP renderer")
        post_code = abi.ACT_SOFTPLUS
    else:
        raise VoxeError(f"density_postactivation {post!r} is not supported by the HIP renderer (Identity | ReLU | Softplus)")
    return pre_code, post_code


class VoxelGrid(Module):
    def __init__(
        self,
        densities: Tensor,  # [X, Y, Z, 1]
        features: Tensor,  # [X, Y, Z, F]
        voxel_size: VoxelSize,
        grid_location: Optional[VoxelGridLocation] = VoxelGridLocation(),
        density_preactivation: Callable[[Tensor], Tensor] = torch.abs,
        density_postactivation: Callable[[Tensor], Tensor] = torch.nn.Identity(),
        feature_preactivation: Callable[[Tensor], Tensor] = torch.nn.Identity(),
        feature_postactivation: Callable[[Tensor], Tensor] = torch.nn.Identity(),
        radiance_transfer_function: Callable[[Tensor, Tensor], Tensor] = None,
        expected_density_scale: float = 1.0,
        tunable: bool = False,
        attn: Optional[Tensor] = None,
    ):
        if densities.dim() != 4 or densities.shape[-1] != 1:
            raise AssertionError(f"densities should be of shape [W x D x H x 1], got {tuple(densities.shape)}")
        if features.dim() != 4:
            raise AssertionError(f"features should be of shape [W x D x H x F], got {tuple(features.shape)}")
        if densities.device != features.device:
            raise AssertionError("densities and features are not on the same device :(")
        super().__init__()

        self._density_preactivation = density_preactivation
        self._density_postactivation = density_postactivation
        self._feature_preactivation = feature_preactivation
        self._feature_postactivation = feature_postactivation
        self._radiance_transfer_function = radiance_transfer_function
        self._grid_location = grid_location
        self._voxel_size = voxel_size
        self._expected_density_scale = expected_density_scale
        self._tunable = tunable
        # snapshot handle used by forward_attn(orig_densities=True); like the reference it aliases the
        # live values until update_orig_densities() is called (voxels.py:103,135-136)
        self.orig_densities = densities

        if tunable:
            self._densities = torch.nn.Parameter(densities)
            self._features = torch.nn.Parameter(features)
            self.attn = torch.nn.Parameter(attn) if attn is not None else None
        else:
            self._densities, self._features, self.attn = densities, features, attn

        self.width_x, self.depth_y, self.height_z = (int(s) for s in features.shape[:3])
        self._aabb = self._setup_bounding_box_planes()
        self._voxe_workspaces: Dict[str, _ops.Workspace] = {}

    # -- tensors ----------------------------------------------------------------------------------
    def add_attn_params(self, attn: Tensor) -> None:
        self.attn = torch.nn.Parameter(attn)

    def update_orig_densities(self) -> None:
        self.orig_densities = self._densities.clone().detach()

    @property
    def densities(self) -> Tensor:
        return self._densities

    @densities.setter
    def densities(self, densities: Tensor) -> None:
        if densities.shape != self._densities.shape:
            raise AssertionError("new densities don't match original densities tensor's dimensions")
        wrap = self._tunable and not isinstance(densities, torch.nn.Parameter)
        self._densities = torch.nn.Parameter(densities) if wrap else densities

    @property
    def features(self) -> Tensor:
        return self._features

    @features.setter
    def features(self, features: Tensor) -> None:
        if features.shape != self._features.shape:
            raise AssertionError("new features don't match original feature tensor's dimensions")
        wrap = self._tunable and not isinstance(features, torch.nn.Parameter)
        self._features = torch.nn.Parameter(features) if wrap else features

    # -- geometry ---------------------------------------------------------------------------------
    @property
    def aabb(self) -> AxisAlignedBoundingBox:
        return self._aabb

    @property
    def grid_dims(self) -> Tuple[int, int, int]:
        return self.width_x, self.depth_y, self.height_z

    @property
    def voxel_size(self) -> VoxelSize:
        return self._voxel_size

    @voxel_size.setter
    def voxel_size(self, voxel_size: VoxelSize) -> None:
        self._voxel_size = voxel_size

    def _setup_bounding_box_planes(self) -> AxisAlignedBoundingBox:
        """centre -/+ (count * voxel edge) / 2 per axis, in python floats (voxels.py:196-223)"""
        ranges = []
        for count, edge, centre in zip(self.grid_dims, self._voxel_size, self._grid_location):
            half = (count * edge) / 2
            ranges.append((centre - half, centre + half))
        return AxisAlignedBoundingBox(*ranges)

    def get_bounding_volume_vertices(self) -> Tensor:
        xs, ys, zs = self._aabb
        return torch.tensor([[x, y, z] for x in xs for y in ys for z in zs], dtype=torch.float32)

    def test_inside_volume(self, points: Tensor) -> Tensor:
        """strict lo < p < hi on all three axes -> bool [N, 1] (voxels.py:263-285).  Host-side helper;
        the kernels apply the same strict test per sample."""
        inside = torch.ones_like(points[..., 0:1], dtype=torch.bool)
        for axis, (lo, hi) in enumerate(self._aabb):
            coord = points[..., axis: axis + 1]
            inside = inside & (coord > lo) & (coord < hi)
        return inside

    # -- configuration ----------------------------------------------------------------------------
    def get_config_dict(self) -> Dict[str, Any]:
        return {
            "grid_location": self._grid_location,
            "density_preactivation": self._density_preactivation,
            "density_postactivation": self._density_postactivation,
            "feature_preactivation": self._feature_preactivation,
            "feature_postactivation": self._feature_postactivation,
            "radiance_transfer_function": self._radiance_transfer_function,
            "expected_density_scale": self._expected_density_scale,
            "tunable": self._tunable,
        }

    def get_save_config_dict(self) -> Dict[str, Any]:
        config = self.get_config_dict()
        config["voxel_size"] = self._voxel_size
        return config

    def extra_repr(self) -> str:
        return (
            f"grid_dims: {self.grid_dims}, feature_dims: {self._features.shape[-1]}, "
            f"voxel_size: {self._voxel_size}, grid_location: {self._grid_location}, tunable: {self._tunable}"
        )

    # -- bridge to the HIP renderer ---------------------------------------------------------------
    def voxe_grid_spec(self, attn: bool = False) -> _ops.GridSpec:
        """Static description handed to the kernels; raises for configurations without a HIP path."""
        if not (_is_identity(self._feature_preactivation) and _is_identity(self._feature_postactivation)):
            raise VoxeError("feature pre/post-activations other than Identity are not supported by the HIP renderer")
        if self._radiance_transfer_function is not None:
            raise VoxeError("radiance_transfer_function is not supported by the HIP renderer")
        pre, post = density_activation_codes(self._density_preactivation, self._density_postactivation)
        return _ops.GridSpec(
            aabb=tuple((float(lo), float(hi)) for lo, hi in self._aabb),
            density_scale=float(self._expected_density_scale),
            density_pre_act=pre,
            density_post_act=post,
            feature_kind=abi.FEAT_ATTN if attn else abi.FEAT_SH,
        )

    def voxe_workspace(self, tag: str) -> _ops.Workspace:
        ws = self._voxe_workspaces.get(tag)
        if ws is None:
            ws = self._voxe_workspaces[tag] = _ops.Workspace()
        return ws

    def __deepcopy__(self, memo):
        import copy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            # scratch buffers are per-instance caches, not state
            new.__dict__[k] = {} if k == "_voxe_workspaces" else copy.deepcopy(v, memo)
        return new

    def _query(self, points: Tensor, features: Tensor, densities: Tensor, attn: bool) -> Tensor:
        out = _ops.query_points(self.voxe_grid_spec(attn=attn), densities, features, points,
                                workspace=self.voxe_workspace("query_attn" if attn else "query"))
        # the reference squeezes its grid_sample outputs (voxels.py:316,331): a single point loses its axis
        return out.reshape(-1) if points.shape[0] == 1 and not attn else out

    def forward(self, points: Tensor, viewdirs: Optional[Tensor] = None) -> Tensor:
        """[N,3] world points -> [N, F+1] = cat(trilinear features, post-activated trilinear density), zero padding
        outside the grid, NOT masked by the AABB (voxels.py:287-342).  Runs the HIP point-query kernel; differentiable
        w.r.t. the grid tensors.  (The renderer does not go through this: sampling is fused into its kernels.)"""
        return self._query(points, self._features, self._densities, attn=False)

    def forward_attn(self, points: Tensor, viewdirs: Optional[Tensor] = None, orig_densities=False) -> Tensor:
        """like forward() with the attention grid as the feature: [N, 2] = (attn, density) (voxels.py:344-406)"""
        if self.attn is None:
            raise VoxeError("this VoxelGrid has no attention grid (attn is None)")
        densities = self.orig_densities.detach().to(self.attn.device) if orig_densities else self._densities
        return self._query(points, self.attn, densities, attn=True)


# ------------------------------------------------------------------------------------------------
def _rescaled_voxel_size(voxel_grid: VoxelGrid, output_size: Tuple[int, int, int]) -> VoxelSize:
    old = voxel_grid.voxel_size
    return VoxelSize(
        (old.x_size * voxel_grid.width_x) / output_size[0],
        (old.y_size * voxel_grid.depth_y) / output_size[1],
        (old.z_size * voxel_grid.height_z) / output_size[2],
    )


def scale_voxel_grid_with_required_output_size(
    voxel_grid: VoxelGrid, output_size: Tuple[int, int, int], mode: str = "trilinear"
) -> VoxelGrid:
    """Resample the grid to `output_size` with trilinear interpolation (align_corners=False semantics of
    F.interpolate, voxels.py:409-447) using the HIP upsampling kernel; the world extent is unchanged."""
    if mode != "trilinear":
        raise VoxeError("only trilinear grid scaling has a HIP path")
    output_size = tuple(int(v) for v in output_size)
    with torch.no_grad():
        new_features = _ops.upsample_trilinear(voxel_grid.features, output_size)
        new_densities = _ops.upsample_trilinear(voxel_grid.densities, output_size)
    return VoxelGrid(
        densities=new_densities,
        features=new_features,
        voxel_size=_rescaled_voxel_size(voxel_grid, output_size),
        **voxel_grid.get_config_dict(),
    )


def scale_voxel_grid_with_required_output_size_attn(
    voxel_grid: VoxelGrid, output_size: Tuple[int, int, int], mode: str = "trilinear"
) -> VoxelGrid:
    """Attention-grid aware variant: features, densities and attn are each resampled.  (The reference's
    version, voxels.py:449-488, slices the concatenated channels inconsistently and has no caller.)"""
    new_grid = scale_voxel_grid_with_required_output_size(voxel_grid, output_size, mode)
    if voxel_grid.attn is not None:
        with torch.no_grad():
            attn = _ops.upsample_trilinear(voxel_grid.attn, tuple(int(v) for v in output_size))
        new_grid.add_attn_params(attn) if new_grid._tunable else setattr(new_grid, "attn", attn)
    return new_grid


def create_voxel_grid_from_saved_info_dict(saved_info: Dict[str, Any]) -> VoxelGrid:
    state = saved_info[THRE3D_REPR][STATE_DICT]
    voxel_grid = VoxelGrid(
        densities=torch.empty_like(state[u_DENSITIES]),
        features=torch.empty_like(state[u_FEATURES]),
        **saved_info[THRE3D_REPR][CONFIG_DICT],
    )
    voxel_grid.load_state_dict(state)
    return voxel_grid


def create_voxel_grid_from_saved_info_dict_attn(saved_info: Dict[str, Any], load_attn: bool = False) -> VoxelGrid:
    """Like the above plus an attention grid: loaded from the checkpoint, or created at -20 (sigmoid ~ 0)
    when the checkpoint has none (voxels.py:501-517)."""
    state = saved_info[THRE3D_REPR][STATE_DICT]
    densities = torch.empty_like(state[u_DENSITIES])
    features = torch.empty_like(state[u_FEATURES])
    config = saved_info[THRE3D_REPR][CONFIG_DICT]
    if load_attn:
        voxel_grid = VoxelGrid(densities=densities, features=features, attn=torch.empty_like(state[u_ATTN]), **config)
        voxel_grid.load_state_dict(state)
        return voxel_grid
    voxel_grid = VoxelGrid(densities=densities, features=features, **config)
    voxel_grid.load_state_dict(state)
    voxel_grid.add_attn_params(torch.full_like(densities, -20.0))
    return voxel_grid
