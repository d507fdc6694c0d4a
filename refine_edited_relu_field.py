#!/usr/bin/env python3
"""Local-edit refinement of an SDS-edited field (entry point kept from the reference's
refine_edited_relu_field.py:34-283): optimise an "edit" and an "object" attention grid against diffusion
cross-attention maps, graph-cut the edit region on the GPU and splice the edited voxels into the original field.
Renders, TV loss, Adam, graph construction, minimum cut: HIP library; cross-attention maps: guidance object
(Stable Diffusion under PyTorch-ROCm)."""
import os
import sys
from pathlib import Path

import click
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "vox-e_amd"))

from thre3d_atom.modules.attn_grid_trainer import refine_edited_relu_field  # noqa: E402
from thre3d_atom.modules.volumetric_model import (  # noqa: E402
    create_volumetric_model_from_saved_model,
    create_volumetric_model_from_saved_model_attn,
)
from thre3d_atom.thre3d_reprs.voxels import (  # noqa: E402
    create_voxel_grid_from_saved_info_dict,
    create_voxel_grid_from_saved_info_dict_attn,
)
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS  # noqa: E402
from thre3d_atom.utils.cli_compat import accepted_options, report_unused  # noqa: E402
from thre3d_atom.utils.logging import log  # noqa: E402
from thre3d_atom.utils.misc import log_config_to_disk  # noqa: E402

# options the reference's script declares and never reads on this path (it loads trained models), or that configure
# machinery this build does not have (data-loader workers, wandb accounts, periodic test-set evaluation, LR stages of a
# single-stage run)
COMPAT_ONLY = [
    ("--separate_train_test_folders", click.BOOL, True, 1), ("--grid_dims", click.INT, (160, 160, 160), 3),
    ("--grid_location", click.FLOAT, (0.0, 0.0, 0.0), 3), ("--normalize_scene_scale", click.BOOL, False, 1),
    ("--grid_world_size", click.FLOAT, (3.0, 3.0, 3.0), 3), ("--sh_degree", click.INT, 0, 1),
    ("--use_relu_field", click.BOOL, True, 1), ("--use_softplus_field", click.BOOL, True, 1),
    ("--render_num_samples_per_ray", click.INT, 1024, 1), ("--parallel_rays_chunk_size", click.INT, 32768, 1),
    ("--ray_batch_size", click.INT, 84672, 1), ("--train_num_samples_per_ray", click.INT, 256, 1),
    ("--num_stages", click.INT, 1, 1), ("--scale_factor", click.FLOAT, 2.0, 1),
    ("--lr_decay_steps_per_stage", click.INT, 5000 * 100, 1), ("--lr_decay_gamma_per_stage", click.FLOAT, 0.1, 1),
    ("--stagewise_lr_decay_gamma", click.FLOAT, 0.9, 1), ("--apply_diffuse_render_regularization", click.BOOL, True, 1),
    ("--num_workers", click.INT, 4, 1), ("--test_frequency", click.INT, 250, 1),
    ("--verbose_rendering", click.BOOL, False, 1), ("--directional_dataset", click.BOOL, True, 1),
    ("--wandb_username", click.STRING, "etaisella", 1), ("--wandb_project_name", click.STRING, "Vox-E-refine", 1),
]


@click.command()
@click.option("-i", "--sds_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the SDS-edited model")
@click.option("-r", "--ref_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the pre-trained (un-edited) model")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path for training output")
@click.option("-p", "--prompt", type=click.STRING, required=True, help="prompt used for attention")
@click.option("-eidx", "--edit_idx", required=True, type=click.STRING, help="space separated 1-based token indices of the edit words")
@click.option("-oidx", "--object_idx", type=click.INT, default=None, help="token index of the object (default: max over non-edit tokens)")
@click.option("-d", "--data_path", type=click.Path(file_okay=False, dir_okay=True), default=None,
              help="input dataset (only needed for --data_pose_mode; cameras otherwise come from the checkpoint)")
@click.option("-a", "--hf_auth_token", type=click.STRING, default="", help="hugging face token for stable diffusion 1.4")
@click.option("-t", "--timestamp", type=click.INT, default=200, show_default=True, help="diffusion timestamp")
@click.option("--data_downsample_factor", type=click.FloatRange(min=1.0), default=3.0, show_default=True)
@click.option("--white_bkgd", type=click.BOOL, default=True, show_default=True)
@click.option("--num_iterations_per_stage", type=click.INT, default=1500, show_default=True)
@click.option("--learning_rate", type=click.FLOAT, default=0.028, show_default=True)
@click.option("--save_frequency", type=click.INT, default=250, show_default=True)
@click.option("--feedback_frequency", type=click.INT, default=200, show_default=True)
@click.option("--summary_frequency", type=click.INT, default=50, show_default=True)
@click.option("--data_pose_mode", type=click.BOOL, default=False, show_default=True)
@click.option("--downsample_refine_grid", type=click.BOOL, default=False, show_default=True)
@click.option("--attn_tv_weight", type=click.FLOAT, default=0.01, show_default=True)
@click.option("--kval", type=click.FLOAT, default=5.0, show_default=True)
@click.option("--edit_mask_thresh", type=click.FLOAT, default=0.992, show_default=True)
@click.option("--num_obj_voxels_thresh", type=click.INT, default=5000, show_default=True)
@click.option("--min_num_edit_voxels", type=click.INT, default=300, show_default=True)
@click.option("--top_k_edit_thresh", type=click.INT, default=300, show_default=True)
@click.option("--top_k_obj_thresh", type=click.INT, default=200, show_default=True)
@click.option("--log_wandb", type=click.BOOL, default=False, show_default=True)
@accepted_options(COMPAT_ONLY)
def main(**kwargs) -> None:
    cfg = type("Config", (), kwargs)
    report_unused(kwargs, COMPAT_ONLY, log)
    device = torch.device("cuda")
    output_path = Path(cfg.output_path)
    log_config_to_disk(kwargs, output_path)
    pretrained_vol_mod, _ = create_volumetric_model_from_saved_model(
        Path(cfg.ref_model_path), create_voxel_grid_from_saved_info_dict, device=device)
    attn_models = []
    extra = None
    for _ in range(3):  # edit grid, object grid, output: three independent copies of the SDS-edited field
        vm, extra = create_volumetric_model_from_saved_model_attn(
            Path(cfg.sds_model_path), create_voxel_grid_from_saved_info_dict_attn, device=device)
        vm.render_config.white_bkgd = cfg.white_bkgd
        attn_models.append(vm)
    vol_mod_edit, vol_mod_obj, sds_vol_mod = attn_models
    dataset = None
    if cfg.data_path is not None:   # (its intrinsics are the ones the refinement renders at, like the reference)
        from thre3d_atom.data.datasets import PosedImagesDataset

        dataset = PosedImagesDataset(Path(cfg.data_path) / "train", Path(cfg.data_path) / "train_camera_params.json",
                                     downsample_factor=cfg.data_downsample_factor, rgba_white_bkgd=cfg.white_bkgd)
    refine_edited_relu_field(
        vol_mod_edit=vol_mod_edit, vol_mod_object=vol_mod_obj, vol_mod_ref=pretrained_vol_mod,
        vol_mod_output=sds_vol_mod, train_dataset=dataset, hf_auth_token=cfg.hf_auth_token, output_dir=output_path,
        prompt=cfg.prompt, edit_idx=[int(i) for i in cfg.edit_idx.split()], object_idx=cfg.object_idx,
        timestamp=cfg.timestamp, image_dims=None, num_iterations=cfg.num_iterations_per_stage,
        learning_rate=cfg.learning_rate, save_freq=cfg.save_frequency, feedback_freq=cfg.feedback_frequency,
        summary_freq=cfg.summary_frequency, attn_tv_weight=cfg.attn_tv_weight, kval=cfg.kval,
        edit_mask_thresh=cfg.edit_mask_thresh, num_obj_voxels_thresh=cfg.num_obj_voxels_thresh,
        min_num_edit_voxels=cfg.min_num_edit_voxels, top_k_edit_thresh=cfg.top_k_edit_thresh,
        top_k_obj_thresh=cfg.top_k_obj_thresh, data_pose_mode=cfg.data_pose_mode,
        downsample_refine_grid=cfg.downsample_refine_grid, camera_intrinsics=extra[CAMERA_INTRINSICS],
        camera_bounds=extra[CAMERA_BOUNDS], saved_hemispherical_radius=extra.get(HEMISPHERICAL_RADIUS, 4.0311),
        log_wandb=cfg.log_wandb,
    )


if __name__ == "__main__":
    main()
