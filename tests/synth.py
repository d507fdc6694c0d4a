"""The deterministic synthetic workload of BASELINE.json / SURVEY.md section 8d (grid, cameras, bounds)."""
import numpy as np
import torch

RADIUS = 4.0311
NEAR, FAR = 1.8, 6.6
CAMERA_ANGLE_X = 0.6911112


def focal_for(width: int) -> float:
    return 0.5 * width / np.tan(0.5 * CAMERA_ANGLE_X)


def synth_pose_angles(i: int, n: int):
    return 360.0 * i / n, 15.0 + 75.0 * ((i * 0.618034) % 1.0)


def random_grid(side: int, nfeat: int = 3, seed: int = 42):
    """densities, features ~ U(-1, 1) from torch.Generator(seed) (mirrors the reference CLI's init)."""
    g = torch.Generator().manual_seed(seed)
    dens = torch.empty((side, side, side, 1)).uniform_(-1.0, 1.0, generator=g)
    feat = torch.empty((side, side, side, nfeat)).uniform_(-1.0, 1.0, generator=g)
    return dens, feat


def sphere_grid(side: int, world: float = 3.0, radius: float = 1.0):
    """structured scene: raw density +1 inside a solid sphere, -1 outside; smooth colour field"""
    ax = (torch.arange(side, dtype=torch.float32) + 0.5) / side * world - world / 2
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    r = torch.sqrt(x * x + y * y + z * z)
    dens = torch.where(r < radius, torch.tensor(1.0), torch.tensor(-1.0))[..., None].contiguous()
    feat = torch.stack([torch.sin(2.0 * x), torch.cos(3.0 * y), torch.sin(2.5 * z + 1.0)], dim=-1).contiguous()
    return dens, feat
