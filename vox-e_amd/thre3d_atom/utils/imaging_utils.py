"""Camera tuples, range adjustment and synthetic camera paths.

Mirrors the names of the reference's thre3d_atom/utils/imaging_utils.py.  The numeric conventions
that define index math downstream are kept bit-for-bit:
  * adjust_dynamic_range(slack=True) computes scale and bias in np.float32 (imaging_utils.py:57-63);
  * pose_spherical builds float32 4x4 matrices from float64 sines/cosines and multiplies them in
    float32 in the order yaw @ (pitch @ translate) (imaging_utils.py:188-194).
"""
import math
from typing import NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from thre3d_atom.utils.constants import NUM_COLOUR_CHANNELS


class CameraIntrinsics(NamedTuple):
    height: int
    width: int
    focal: float


class CameraPose(NamedTuple):
    rotation: np.array  # [3 x 3]
    translation: np.array  # [3 x 1]


class CameraBounds(NamedTuple):
    near: float
    far: float


def to8b(x: np.array) -> np.array:
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def adjust_dynamic_range(
    data: Union[np.array, Tensor],
    drange_in: Tuple[float, float],
    drange_out: Tuple[float, float],
    slack: bool = False,
):
    """Affine map of `data` from drange_in to drange_out.  slack=True leaves values outside the input
    range un-clipped (this is the form VoxelGrid uses to normalise sample points)."""
    if drange_in == drange_out:
        return data
    in_lo, in_hi = np.float32(drange_in[0]), np.float32(drange_in[1])
    out_lo, out_hi = np.float32(drange_out[0]), np.float32(drange_out[1])
    if slack:
        scale = (out_hi - out_lo) / (in_hi - in_lo)
        bias = out_lo - in_lo * scale
        return data * scale + bias
    data = ((data - in_lo) / (in_hi - in_lo) * (out_hi - out_lo)) + out_lo
    return data.clip(drange_out[0], drange_out[1])


def get_2d_coordinates(height: int, width: int, drange: Tuple[float, float] = (-1.0, 1.0)) -> Tensor:
    lo, hi = drange
    ys = torch.linspace(lo, hi, height, dtype=torch.float32)
    xs = torch.linspace(lo, hi, width, dtype=torch.float32)
    return torch.stack(torch.meshgrid(ys, xs, indexing="ij"), dim=-1)


def postprocess_depth_map(depth_map: np.array, acc_map: Optional[np.array] = None) -> np.array:
    """Depth -> magma colour map, optionally alpha-composited over white using the accumulated weight."""
    import matplotlib.pyplot as plt

    if depth_map.ndim == 3 and depth_map.shape[-1] == 1:
        depth_map = depth_map[..., 0]
    if acc_map is not None:
        lo, hi = depth_map.min(), (depth_map * acc_map[..., 0]).max()
    else:
        lo, hi = depth_map.min(), depth_map.max()
    norm = adjust_dynamic_range(depth_map, drange_in=(lo, hi), drange_out=(0, 1), slack=True)
    coloured = plt.get_cmap("magma", lut=1024)(norm)[..., :NUM_COLOUR_CHANNELS]
    if acc_map is None:
        return to8b(coloured)
    bg = (1.0 - acc_map) ** 2
    return to8b((coloured * acc_map + bg) / (acc_map + bg))


def scale_camera_intrinsics(camera_intrinsics: CameraIntrinsics, scale_factor: float = 1.0) -> CameraIntrinsics:
    return CameraIntrinsics(
        height=int(np.ceil(camera_intrinsics.height * scale_factor)),
        width=int(np.ceil(camera_intrinsics.width * scale_factor)),
        focal=camera_intrinsics.focal * scale_factor,
    )


# ---------------------------------------------------------------------------------------------
# camera-to-world transforms
# ---------------------------------------------------------------------------------------------
def _mat(rows, device) -> Tensor:
    return torch.tensor(rows, dtype=torch.float32, device=device)


def _translate_z(z: float, device=torch.device("cpu")) -> Tensor:
    return _mat([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, z], [0.0, 0.0, 0.0, 1.0]], device)


def _rotate_pitch(pitch: float, device=torch.device("cpu")) -> Tensor:
    c, s = np.cos(pitch), np.sin(pitch)
    return _mat([[1.0, 0.0, 0.0, 0.0], [0.0, c, -s, 0.0], [0.0, s, c, 0.0], [0.0, 0.0, 0.0, 1.0]], device)


def _rotate_yaw(yaw: float, device=torch.device("cpu")) -> Tensor:
    c, s = np.cos(yaw), np.sin(yaw)
    return _mat([[c, -s, 0.0, 0.0], [s, c, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]], device)


def _spherical_c2w(yaw_deg: float, pitch_deg: float, radius: float, device) -> Tensor:
    c2w = _translate_z(radius, device)
    c2w = _rotate_pitch(pitch_deg / 180.0 * np.pi, device) @ c2w
    return _rotate_yaw(yaw_deg / 180.0 * np.pi, device) @ c2w


def pose_spherical(yaw: float, pitch: float, radius: float, device=torch.device("cpu")) -> CameraPose:
    c2w = _spherical_c2w(yaw, pitch, radius, device)
    return CameraPose(rotation=c2w[:3, :3], translation=c2w[:3, 3:])


def view_direction_label(yaw: float, pitch: float) -> str:
    """Prompt suffix bucket of a random SDS pose (imaging_utils.py:206-213)."""
    label = "front"
    if 45.0 < yaw < 315.0:
        label = "side"
    if 120.0 < yaw < 240.0:
        label = "back"
    if pitch < 25.0:
        label = "overhead"
    return label


def get_random_pose(radius: float, device=torch.device("cpu")):
    """Random SDS camera: pitch ~ U[15, 90), yaw ~ U[0, 360) from numpy's global RNG (two draws, pitch
    first, like imaging_utils.py:197-215).  Returns (pose, direction label, pitch, yaw)."""
    pitch = 15.0 + float(np.random.rand(1)[0] * 75.0)
    yaw = float(np.random.rand(1)[0] * 360.0)
    c2w = _spherical_c2w(yaw, pitch, radius, device)
    pose = CameraPose(rotation=c2w[:3, :3], translation=c2w[:3, 3:])
    return pose, view_direction_label(yaw, pitch), pitch, yaw


def get_thre360_animation_poses(hemispherical_radius: float, camera_pitch: float, num_poses: int) -> Sequence[CameraPose]:
    """num_poses - 1 poses on a circle (the closing duplicate is dropped so a looped video is smooth)."""
    return [pose_spherical(yaw, camera_pitch, hemispherical_radius) for yaw in np.linspace(0, 360, num_poses)[:-1]]


def get_thre360_spiral_animation_poses(
    horizontal_radius_range: Tuple[float, float], vertical_camera_height: float, num_rounds: int, num_poses: int
) -> Sequence[CameraPose]:
    radii = np.linspace(*horizontal_radius_range, num_poses)[:-1]
    yaws = np.linspace(0, 360 * num_rounds, num_poses)[:-1]
    poses = []
    for yaw, r in zip(yaws, radii):
        pitch = math.atan(r / vertical_camera_height) * 180 / math.pi
        poses.append(pose_spherical(yaw, pitch, np.sqrt(r ** 2 + vertical_camera_height ** 2)))
    return poses
