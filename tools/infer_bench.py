"""No-grad frame time of VolumetricModel.render (the render_sh_based_voxel_grid path): 160^3 field, 800x800,
render_num_samples_per_ray samples.   gpurun -- python tools/infer_bench.py [S]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import torch  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, sphere_grid, synth_pose_angles  # noqa: E402
from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize  # noqa: E402
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, pose_spherical  # noqa: E402
from voxe_hip import ops  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    HW = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    dev = torch.device("cuda:0")
    G = 160
    dens, feat = sphere_grid(G)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    vm = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(256, CameraBounds(NEAR, FAR), white_bkgd=True),
                         device=dev)
    intr = CameraIntrinsics(HW, HW, focal_for(HW))
    poses = [pose_spherical(*synth_pose_angles(i, 100), RADIUS) for i in range(20)]
    for p in poses[:3]:
        vm.render(p, intr, num_samples_per_ray=S)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for p in poses:
        out = vm.render(p, intr, num_samples_per_ray=S)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(poses)
    prof = ops.profile_read()
    print(f"render {HW}x{HW}, S={S}, 160^3: {dt * 1e3:.2f} ms / frame ({HW * HW / dt / 1e6:.1f} M rays/s, "
          f"{HW * HW * S / dt / 1e9:.1f} G samples/s); forward kernels {prof['ms_fwd'] / max(prof['n_fwd'], 1):.2f} ms, "
          f"pack {prof['ms_pack'] / max(prof['n_pack'], 1):.3f} ms x {prof['n_pack']}")
    assert out.colour.shape == (HW, HW, 3)


if __name__ == "__main__":
    main()
