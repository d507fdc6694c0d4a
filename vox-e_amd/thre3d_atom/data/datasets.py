"""Posed-image datasets feeding the trainers.

`PosedImagesDataset` reads the reference's on-disk format (a folder of images plus
`<split>_camera_params.json`, reference thre3d_atom/data/datasets.py:32-390 and tools/
convert_from_nerf_blender_dataset.py); `InMemoryPosedImages` wraps tensors (synthetic scenes, tests).
Both expose what the trainers use: `camera_intrinsics`, `camera_bounds`,
`get_hemispherical_radius_estimate()`, `images [N,C,H,W]`, `poses [N,3,4]`, `downsampled(factor)`.
Image decoding is host I/O, outside the render hot path."""
import json
from pathlib import Path
from typing import Any, Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor
from torch.utils.data import Dataset

from thre3d_atom.data.constants import (
    BOUNDS, DIRECTION, EXTRINSIC, FOCAL, HEIGHT, INTRINSIC, ROTATION, TRANSLATION, WIDTH,
)
from thre3d_atom.utils.constants import NUM_COLOUR_CHANNELS
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, CameraPose


class InMemoryPosedImages(Dataset):
    def __init__(self, images: Tensor, poses: Tensor, camera_intrinsics: CameraIntrinsics,
                 camera_bounds: CameraBounds):
        if images.dim() != 4 or poses.shape[1:] != (3, 4) or len(images) != len(poses):
            raise ValueError("images must be [N,C,H,W] and poses [N,3,4]")
        self.images, self.poses = images.float(), poses.float()
        self.camera_intrinsics, self.camera_bounds = camera_intrinsics, camera_bounds
        self.cached_data_mode = True

    def __len__(self) -> int:
        return len(self.images)

    def __getitem__(self, index: int) -> Tuple[Tensor, Tensor, int]:
        return self.images[index], self.poses[index], index

    def get_hemispherical_radius_estimate(self) -> float:
        """mean distance of the camera centres from the origin (datasets.py:251-265)"""
        return float(self.poses[:, :, 3].norm(dim=-1).mean())

    def downsampled(self, factor: float) -> "InMemoryPosedImages":
        """images resized by 1/factor (area filter); intrinsics as the reference derives them (datasets.py:288-306):
        (height, width, focal) / factor with height and width TRUNCATED -- 800 px / 3.0 -> 266 px, focal / 3"""
        if factor == 1.0:
            return self
        h, w, f = self.camera_intrinsics
        nh, nw = max(int(h / factor), 1), max(int(w / factor), 1)
        images = F.interpolate(self.images, size=(nh, nw), mode="area")
        return InMemoryPosedImages(images, self.poses, CameraIntrinsics(nh, nw, f / factor), self.camera_bounds)

    def to(self, device) -> "InMemoryPosedImages":
        return InMemoryPosedImages(self.images.to(device), self.poses.to(device), self.camera_intrinsics,
                                   self.camera_bounds)


class PosedImagesDataset(InMemoryPosedImages):
    """Reads `<images_dir>/*.png|jpg` + `<camera_params_json>`; RGBA is composited over white or black.
    Camera bounds are widened to (0.9 near, 1.1 far) like the reference (datasets.py:267-277)."""

    def __init__(self, images_dir: Path, camera_params_json: Path, image_data_range: Tuple[float, float] = (0.0, 1.0),
                 normalize_scene_scale: bool = False, downsample_factor: float = 1.0, rgba_white_bkgd: bool = False,
                 directional: bool = False):
        """directional: every view carries a direction prompt word (`"dir"` in its camera parameters; reference
        datasets.py:41,85-88,331-335); items are then (image, pose, direction, index) like the reference's."""
        from PIL import Image

        self.directional = bool(directional)
        self._config = dict(images_dir=Path(images_dir), camera_params_json=Path(camera_params_json),
                            image_data_range=image_data_range, normalize_scene_scale=normalize_scene_scale,
                            downsample_factor=downsample_factor, rgba_white_bkgd=rgba_white_bkgd)

        images_dir = Path(images_dir)
        params = json.loads(Path(camera_params_json).read_text())
        files = sorted(p for p in images_dir.iterdir() if p.suffix.lower() in (".png", ".jpg", ".jpeg"))
        files = [p for p in files if p.name in params] or files
        if not files:
            raise FileNotFoundError(f"no images under {images_dir}")
        self._camera_parameters = params
        self.directions = [str(params[p.name][DIRECTION]) for p in files] if self.directional else None
        imgs, poses = [], []
        # normalize_scene_scale: every camera location is divided by the distance of the FARTHEST camera from the origin
        # (max norm over all entries of the camera-parameter file, datasets.py:218-249), and so are the bounds
        scale = 1.0
        if normalize_scene_scale:
            radii = [np.linalg.norm(np.array(entry[EXTRINSIC][TRANSLATION], dtype=np.float32)) for entry in params.values()]
            scale = 1.0 / max(float(np.max(radii)), 1e-8)
        for p in files:
            entry = params[p.name]
            img = np.asarray(Image.open(p), dtype=np.float32) / 255.0
            if img.ndim == 2:
                img = np.repeat(img[..., None], NUM_COLOUR_CHANNELS, axis=-1)
            if img.shape[-1] == 4:
                alpha = img[..., 3:]
                img = img[..., :3] * alpha + ((1.0 - alpha) if rgba_white_bkgd else 0.0)
            imgs.append(torch.from_numpy(np.ascontiguousarray(img[..., :3])).permute(2, 0, 1))
            rot = np.array(entry[EXTRINSIC][ROTATION], dtype=np.float32).reshape(3, 3)
            trans = np.array(entry[EXTRINSIC][TRANSLATION], dtype=np.float32).reshape(3, 1) * scale
            poses.append(torch.from_numpy(np.concatenate([rot, trans], axis=1)))
        first = params[files[0].name]
        intr = CameraIntrinsics(int(first[INTRINSIC][HEIGHT]), int(first[INTRINSIC][WIDTH]), float(first[INTRINSIC][FOCAL]))
        # bounds over ALL cameras of the file: min(near) * 0.9, max(far) * 1.1 (datasets.py:267-277), then the scene scale
        all_bounds = np.vstack([np.array(entry[INTRINSIC][BOUNDS]).astype(np.float32) for entry in params.values()])
        bounds = CameraBounds(float(all_bounds.min() * 0.9) * scale, float(all_bounds.max() * 1.1) * scale)
        lo, hi = image_data_range
        images = torch.stack(imgs) * (hi - lo) + lo
        super().__init__(images, torch.stack(poses), intr, bounds)
        if downsample_factor != 1.0:
            small = self.downsampled(downsample_factor)
            self.images, self.camera_intrinsics = small.images, small.camera_intrinsics

    @property
    def camera_parameters(self) -> Dict[str, Any]:
        return self._camera_parameters

    def get_config_dict(self) -> Dict[str, Any]:
        return dict(self._config)

    @staticmethod
    def extract_pose(camera_params: Dict[str, Any]) -> CameraPose:
        rotation = np.array(camera_params[EXTRINSIC][ROTATION], dtype=np.float32).reshape(3, 3)
        translation = np.array(camera_params[EXTRINSIC][TRANSLATION], dtype=np.float32).reshape(3, 1)
        return CameraPose(rotation=rotation, translation=translation)

    @staticmethod
    def extract_dir(camera_params: Dict[str, Any]) -> str:
        return str(camera_params[DIRECTION])

    def __getitem__(self, index: int):
        if self.directional:
            return self.images[index], self.poses[index], self.directions[index], index
        return super().__getitem__(index)
