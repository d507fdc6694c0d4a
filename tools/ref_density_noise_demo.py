"""Why `stochastic_density_noise_std != 0` is refused (thre3d_reprs/renderers.py) rather than implemented: run in the build
container, this imports the REFERENCE and renders a voxel grid with the noise on.  accumulate.py:57-63 adds the noise to the
activated density of EVERY sample, including the last one whose delta is 1e10 (accumulate.py:49-53): where that draw is negative
alpha = 1 - exp(+huge) = -inf and the ray renders NaN -- half of all rays, at any std (profiles/r05_density_noise_reference.txt).
The reference never sets it ("used by NeRF not by us", renderers.py:41)."""
import sys, types, torch
_ed = types.ModuleType("easydict"); _ed.EasyDict = dict; sys.modules.setdefault("easydict", _ed)
sys.path.insert(0,'/root/reference')
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize
from thre3d_atom.thre3d_reprs.renderers import render_sh_voxel_grid, SHVoxGridRenderConfig
from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays
from thre3d_atom.utils.imaging_utils import CameraIntrinsics, CameraBounds, pose_spherical
import inspect
torch.manual_seed(0)
g = VoxelGrid(densities=torch.rand(16,16,16,1), features=torch.rand(16,16,16,3), voxel_size=VoxelSize(3/16,3/16,3/16))
pose = pose_spherical(30., -30., 4.0)
rays = flatten_rays(cast_rays(CameraIntrinsics(24,24,30.0), pose))
print(inspect.signature(SHVoxGridRenderConfig))
for std in (0.0, 1.0, 0.01):
    cfg = SHVoxGridRenderConfig(num_samples_per_ray=32, camera_bounds=CameraBounds(2.0,6.0), white_bkgd=True, stochastic_density_noise_std=std)
    out = render_sh_voxel_grid(g, rays, cfg)
    c = out.colour
    print("std", std, "rays", c.shape[0], "non-finite rays", int((~torch.isfinite(c).all(-1)).sum()))
