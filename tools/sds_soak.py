"""Soak run of the SDS editing loop with a stand-in guidance (no diffusion model): 160^3 field, 266x266 renders (the
reference's default SDS image, 800 / 3), density-correlation regulariser on.  Reports iterations/s of everything
except the UNet, and checks that device memory stays flat.   gpurun -- python tools/sds_soak.py [iters]"""
import copy
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402
from voxe_hip.workload import FAR, NEAR, focal_for, sphere_grid  # noqa: E402
from thre3d_atom.modules.sds_trainer import train_sh_vox_grid_vol_mod_with_posed_images_and_sds  # noqa: E402
from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize  # noqa: E402
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics  # noqa: E402


class Tint:
    def __init__(self, dev):
        self.colour = torch.tensor([0.9, 0.3, 0.1], device=dev)

    def training_step(self, output, h, w, directions=None, global_step=-1, logvars=None):
        return ((output - self.colour) ** 2).mean()

    def get_current_max_step_ratio(self):
        return 0.98


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    G, HW = 160, 266
    dens, feat = sphere_grid(G)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    cfg = SHVoxGridRenderConfig(256, CameraBounds(NEAR, FAR), white_bkgd=True, render_num_samples_per_ray=512)
    ref = VolumetricModel(vg, render_sh_voxel_grid, cfg, device=dev)
    sds = copy.deepcopy(ref)
    torch.manual_seed(0)
    np.random.seed(0)
    mem = []
    out = tempfile.mkdtemp()
    t0 = time.perf_counter()
    train_sh_vox_grid_vol_mod_with_posed_images_and_sds(
        sds, ref, None, None, out, num_iterations=iters, learning_rate=0.03, save_freq=10 ** 9, feedback_freq=10 ** 9,
        summary_freq=100, density_correlation_weight=200.0, guidance=Tint(dev),
        camera_intrinsics=CameraIntrinsics(HW, HW, focal_for(HW)), camera_bounds=CameraBounds(NEAR, FAR))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mem.append(torch.cuda.max_memory_allocated() / 2 ** 20)
    print(f"{iters} SDS iterations (160^3, {HW}x{HW}, S=256, DCL on, stand-in guidance): {dt:.2f} s "
          f"= {dt / iters * 1e3:.2f} ms / iteration, peak device memory {mem[-1]:.0f} MiB")
    d = (sds.thre3d_repr.densities - ref.thre3d_repr.densities).abs().max().item()
    assert np.isfinite(d) and d > 0
    print("max |density change|", d)


if __name__ == "__main__":
    main()
