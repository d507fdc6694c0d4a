"""The deterministic synthetic workload of BASELINE.json / SURVEY.md section 8d (grid, cameras, bounds): what bench.py and
the tools measure and what the full-size parity tests check (tests/synth.py re-exports it)."""
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


def refine_scene(side: int, seed: int = 5, n_obj: int = 5000):
    """Refinement-stage scene: the solid sphere with a textured colour field; "edit" seeds on its cap (z > 0.8),
    `n_obj` random "object" seeds in the lower part (z < 0.2).  -> densities [S,S,S,1], sigmoid colours [S,S,S,3],
    seed coordinates (bool mask of edit seeds, int array [n,3] of object seeds)"""
    import numpy as np

    dens, feat = sphere_grid(side)
    g = torch.Generator().manual_seed(seed)
    col = torch.sigmoid(feat + 0.3 * torch.randn(feat.shape, generator=g))
    ax = (np.arange(side) + 0.5) / side * 3.0 - 1.5
    z = np.broadcast_to(ax[None, None, :], (side,) * 3)
    inside = dens[..., 0].numpy() > 0
    edit = inside & (z > 0.8)
    cand = np.argwhere(inside & (z < 0.2))
    pick = cand[np.random.default_rng(seed).permutation(len(cand))[:n_obj]]
    return dens, col.contiguous(), edit, pick
