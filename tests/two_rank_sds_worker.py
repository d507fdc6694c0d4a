"""Worker of tests/test_bench_two_ranks_gpu.py::test_two_rank_sds_loop_equals_one_process (2 ranks sharing the visible
GPU, gloo): the ray-sharded SDS edit loop -- every rank renders a band of image rows, the bands are all-gathered, the
(stand-in) guidance runs replicated, the per-band gradients meet in one all-reduce -- against the same loop in one
process.  Jitter is off so that a band renders exactly the rows of the full image."""
import copy
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from thre3d_atom.modules.sds_trainer import train_sh_vox_grid_vol_mod_with_posed_images_and_sds  # noqa: E402
from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize  # noqa: E402
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics  # noqa: E402

DEV = torch.device("cuda", 0)


class TintGuidance:
    def __init__(self, colour):
        self.colour = torch.tensor(colour, device=DEV)

    def training_step(self, output, image_height, image_width, directions=None, global_step=-1, logvars=None):
        return ((output - self.colour) ** 2).mean()

    def get_current_max_step_ratio(self):
        return 0.98


def sphere_model(side=24, samples=64):
    ax = (torch.arange(side, dtype=torch.float32) + 0.5) / side * 3.0 - 1.5
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    r = torch.sqrt(x * x + y * y + z * z)
    dens = torch.where(r < 0.9, torch.tensor(1.0), torch.tensor(-1.0))[..., None].contiguous()
    feat = torch.stack([2.0 * torch.sin(2 * x), 2.0 * torch.cos(3 * y), 2.0 * torch.sin(2.5 * z + 1)], dim=-1).contiguous()
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / side, 3.0 / side, 3.0 / side), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    cfg = SHVoxGridRenderConfig(samples, CameraBounds(1.8, 6.6), white_bkgd=True, render_num_samples_per_ray=64,
                                perturb_sampled_points=False)
    return VolumetricModel(vg, render_sh_voxel_grid, cfg, device=DEV)


def run(tag):
    torch.manual_seed(0)
    np.random.seed(0)
    ref = sphere_model()
    sds = copy.deepcopy(ref)
    with tempfile.TemporaryDirectory() as tmp:
        train_sh_vox_grid_vol_mod_with_posed_images_and_sds(
            sds, ref, None, None, tmp, num_iterations=12, learning_rate=0.05, save_freq=1000, feedback_freq=1000,
            summary_freq=1000, density_correlation_weight=5.0, guidance=TintGuidance([1.0, 0.1, 0.1]),
            camera_intrinsics=CameraIntrinsics(40, 40, 55.0), camera_bounds=CameraBounds(1.8, 6.6))
    torch.cuda.synchronize()
    return ref, sds


def main():
    torch.cuda.set_device(0)
    ref, solo = run("solo")                                   # no process group yet: the one-process loop
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    _, multi = run("two ranks")                               # 20 + 20 rows per rank

    def rel(a, b):
        return float((a.detach() - b.detach()).norm() / b.detach().norm())

    out = {"rel_densities": rel(multi.thre3d_repr.densities, solo.thre3d_repr.densities),
           "rel_features": rel(multi.thre3d_repr.features, solo.thre3d_repr.features),
           "moved": rel(solo.thre3d_repr.features, ref.thre3d_repr.features)}
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, out)
    if dist.get_rank() == 0:
        print(json.dumps({"ranks": gathered}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
