"""BASELINE.json configs[2] and configs[3] as LOOPS at their full grid size on the GPU (the kernels underneath are pinned by
tests/test_hip_configs.py; the SD networks are the stand-ins of tests/sds_standins.py -- diffusers / weights are absent):

  configs[2]  global edit: edit_pretrained_relu_field's SDS loop (modules/sds_trainer.py:218-470) -- 160^3 field,
              266 x 266 renders (800-pixel data at the default --data_downsample_factor 3), `scoreDistillationLoss`
              guidance, density-correlation + TV regularisers, FusedGridAdam;
  configs[3]  local edit: the attention-grid refinement loop (modules/attn_grid_trainer.py:226-399) + graph cut + splice at
              160^3 with a stand-in attention source."""
import copy

import numpy as np
import pytest
import torch

import sds_standins as st
from synth import FAR, NEAR, sphere_grid

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from thre3d_atom.modules.attn_grid_trainer import refine_edited_relu_field
    from thre3d_atom.modules.sds_trainer import train_sh_vox_grid_vol_mod_with_posed_images_and_sds
    from thre3d_atom.modules.volumetric_model import VolumetricModel, create_volumetric_model_from_saved_model
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid, render_sh_voxel_grid_attn
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize, create_voxel_grid_from_saved_info_dict
    from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics

    DEV = torch.device("cuda:0")

G = 160


def _model(attn=False):
    dens, feat = sphere_grid(G)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    if attn:
        vg.add_attn_params(torch.full_like(dens, -20.0))
    cfg = SHVoxGridRenderConfig(256, CameraBounds(NEAR, FAR), white_bkgd=True, render_num_samples_per_ray=256)
    return VolumetricModel(vg, render_sh_voxel_grid, cfg, render_procedure_attn=render_sh_voxel_grid_attn, device=DEV)


def test_cfg3_sds_edit_loop_160_on_the_standin_sd_stack(tmp_path):
    torch.manual_seed(1)
    np.random.seed(1)
    ref = _model()
    sds = copy.deepcopy(ref)
    before_d = sds.thre3d_repr.densities.detach().clone()
    before_f = sds.thre3d_repr.features.detach().clone()
    intr = CameraIntrinsics(266, 266, 1111.111 / 3.0)         # 800-pixel data at data_downsample_factor 3
    with st.installed():                                      # scoreDistillationLoss builds its SD pieces from these
        out = train_sh_vox_grid_vol_mod_with_posed_images_and_sds(
            sds, ref, None, None, tmp_path, num_iterations=6, learning_rate=0.01, save_freq=3, feedback_freq=3,
            summary_freq=1, sds_prompt="a yarn doll", density_correlation_weight=200.0, tv_density_weight=1e-3,
            tv_features_weight=1e-3, sds_t_start=2, sds_t_freq=2, sds_t_gamma=0.9,
            camera_intrinsics=intr, camera_bounds=CameraBounds(NEAR, FAR))
    assert out is sds
    d, f = sds.thre3d_repr.densities.detach(), sds.thre3d_repr.features.detach()
    assert torch.isfinite(d).all() and torch.isfinite(f).all()
    assert not torch.equal(d, before_d) and not torch.equal(f, before_f)            # SDS + regularisers moved the grid
    assert float((f - before_f).abs().max()) <= 6 * 0.01 * 1.0001                    # Adam: at most lr per step
    assert sds.thre3d_repr.densities.grad is None                                   # deferred-gradient mode left no .grad
    assert sds.thre3d_repr.voxe_workspace("sh").deferred is None                    # ... and was detached at the end
    assert torch.equal(ref.thre3d_repr.densities.detach(), before_d)
    vm, extra = create_volumetric_model_from_saved_model(tmp_path / "saved_models" / "model_final.pth",
                                                         create_voxel_grid_from_saved_info_dict, device=DEV)
    assert torch.equal(vm.thre3d_repr.features, f) and tuple(extra["camera_intrinsics"])[:2] == (266, 266)
    assert (tmp_path / "training_logs" / "rendered_output" / "sds_6.png").exists()


class _SideAttention:
    """stand-in for the UNet cross-attention maps: token 1 where the render is reddish, token 2 where it is bluish"""

    def get_num_tokens(self, prompt):
        return 4

    def get_attn_map(self, prompt, pred_rgb, timestamp=0, indices_to_fetch=(7,)):
        rgb = pred_rgb[0]
        flat = torch.full_like(rgb[0], 0.01)
        return [(rgb[0] - rgb[2]).clamp(min=0), (rgb[2] - rgb[0]).clamp(min=0), flat, flat], None


def test_cfg4_refinement_loop_160_graph_cut_and_splice(tmp_path):
    torch.manual_seed(2)
    np.random.seed(2)
    reference = _model(attn=True)
    edited = copy.deepcopy(reference)
    with torch.no_grad():      # the "edit": the top cap of the sphere turns blue and grows a little
        ax = (torch.arange(G, dtype=torch.float32) + 0.5) / G * 3.0 - 1.5
        z = ax[None, None, :].expand(G, G, G).to(DEV)
        inside = edited.thre3d_repr.densities[..., 0] > 0
        cap = inside & (z > 0.55)
        feat = edited.thre3d_repr.features
        feat[..., 0].copy_(torch.where(inside, torch.tensor(2.0, device=DEV), feat[..., 0]))
        feat[..., 2].copy_(torch.where(inside, torch.tensor(-2.0, device=DEV), feat[..., 2]))
        feat[cap] = torch.tensor([-2.0, -2.0, 2.0], device=DEV)
    vm_edit, vm_obj, vm_out = copy.deepcopy(edited), copy.deepcopy(edited), copy.deepcopy(edited)
    out = refine_edited_relu_field(
        vm_edit, vm_obj, vm_out, reference, train_dataset=None, hf_auth_token="", output_dir=tmp_path,
        prompt="a ball wearing a hat", edit_idx=[2], timestamp=200, image_dims=None, num_iterations=40, learning_rate=0.3,
        feedback_freq=40, save_freq=40, summary_freq=10, attn_tv_weight=0.001, edit_mask_thresh=0.97,
        num_obj_voxels_thresh=5000, min_num_edit_voxels=300, top_k_edit_thresh=300, top_k_obj_thresh=200,
        attn_guidance=_SideAttention(), camera_intrinsics=CameraIntrinsics(266, 266, 1111.111 / 3.0),
        camera_bounds=CameraBounds(NEAR, FAR))
    assert out is vm_out
    e_attn = vm_edit.thre3d_repr.attn.detach()[..., 0]
    assert torch.isfinite(e_attn).all() and e_attn.max() > -19.0
    assert e_attn[cap].mean() > e_attn[inside & ~cap].mean()                  # the edit attention found the cap
    keep = vm_out.thre3d_repr.attn.detach()[..., 0]
    edit_region = keep == 0
    assert 0 < int(edit_region.sum()) < int(inside.sum())
    occupied_cut = edit_region & inside
    assert float((occupied_cut & cap).sum()) / float(occupied_cut.sum()) > 0.7   # the cut-out region is (mostly) the cap
    new_d, ref_d, old_d = (m.thre3d_repr._densities.detach() for m in (vm_out, reference, edited))
    assert torch.equal(new_d[~edit_region], ref_d[~edit_region]) and torch.equal(new_d[edit_region], old_d[edit_region])
    for name in ("model_final_attn_edit.pth", "model_final_attn_object.pth", "model_final_refined.pth"):
        assert (tmp_path / "saved_models" / name).exists()
