#!/usr/bin/env python3
"""Text-guided edit of a pre-trained ReLU/softplus field with score distillation (entry point kept from the
reference's edit_pretrained_relu_field.py:234-319; option names of the global-edit stage).  The render
forward/backward, the density-correlation regulariser and Adam run in the HIP library; Stable Diffusion
(diffusers) runs under PyTorch-ROCm.  `--do_refinement` chains the local-edit refinement stage (attention grids +
GPU graph cut, :321-373; also available stand-alone as refine_edited_relu_field.py) and `--post_process_scc`
restores the original densities outside the largest connected component (:374-427, GPU component labelling)."""
import copy
import os
import sys
from pathlib import Path

import click
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "vox-e_amd"))

from thre3d_atom.modules.attn_grid_trainer import refine_edited_relu_field  # noqa: E402
from thre3d_atom.modules.refinement_functions import restore_outside_largest_component  # noqa: E402
from thre3d_atom.modules.sds_trainer import train_sh_vox_grid_vol_mod_with_posed_images_and_sds  # noqa: E402
from thre3d_atom.modules.volumetric_model import (  # noqa: E402
    create_volumetric_model_from_saved_model,
    create_volumetric_model_from_saved_model_attn,
)
from thre3d_atom.thre3d_reprs.voxels import (  # noqa: E402
    create_voxel_grid_from_saved_info_dict,
    create_voxel_grid_from_saved_info_dict_attn,
)
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS  # noqa: E402
from thre3d_atom.utils.imaging_utils import scale_camera_intrinsics  # noqa: E402
from thre3d_atom.utils.cli_compat import accepted_options, report_unused  # noqa: E402
from thre3d_atom.utils.logging import log  # noqa: E402
from thre3d_atom.utils.misc import log_config_to_disk  # noqa: E402

# options the reference's script declares and never reads on this path (it loads a trained model), or that configure
# machinery this build does not have (data-loader workers, wandb accounts, periodic test-set evaluation)
COMPAT_ONLY = [
    ("--separate_train_test_folders", click.BOOL, True, 1), ("--grid_dims", click.INT, (160, 160, 160), 3),
    ("--grid_location", click.FLOAT, (0.0, 0.0, 0.0), 3), ("--normalize_scene_scale", click.BOOL, False, 1),
    ("--grid_world_size", click.FLOAT, (3.0, 3.0, 3.0), 3), ("--sh_degree", click.INT, 0, 1),
    ("--use_relu_field", click.BOOL, True, 1), ("--use_softplus_field", click.BOOL, True, 1),
    ("--parallel_rays_chunk_size", click.INT, 32768, 1), ("--ray_batch_size", click.INT, 84672, 1),
    ("--scale_factor", click.FLOAT, 2.0, 1), ("--apply_diffuse_render_regularization", click.BOOL, True, 1),
    ("--num_workers", click.INT, 4, 1), ("--wandb_username", click.STRING, "etaisella", 1),
    ("--wandb_project_name", click.STRING, "Vox-E", 1), ("--test_frequency", click.INT, 500, 1),
    ("--verbose_rendering", click.BOOL, False, 1), ("--fast_debug_mode", click.BOOL, False, 1),
]


def edit_stage_intrinsics(checkpoint_intrinsics, dataset, data_downsample_factor: float):
    """Camera intrinsics of the SDS edit and refinement stages.  The reference renders them at the DATASET's intrinsics,
    built at --data_downsample_factor (edit_pretrained_relu_field.py:253-274, sds_trainer.py:152-156,270-277: default 3.0,
    i.e. 266 x 266 for 800-pixel data), whatever resolution the checkpoint was trained at.  With a dataset: exactly
    those.  Without one (this build allows editing from a checkpoint alone): the checkpoint's intrinsics -- the
    training resolution, factor 1.0 with the reference's training defaults -- divided by the factor the way the
    dataset class does it (height / width truncated, focal / factor)."""
    if dataset is not None:
        return dataset.camera_intrinsics
    h, w, f = checkpoint_intrinsics
    factor = float(data_downsample_factor)
    return type(checkpoint_intrinsics)(max(int(h / factor), 1), max(int(w / factor), 1), f / factor)


@click.command()
@click.option("-i", "--ref_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the pre-trained relu field model")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path for training output")
@click.option("-p", "--prompt", type=click.STRING, required=True, help="prompt used for the SDS based loss")
@click.option("-d", "--data_path", type=click.Path(file_okay=False, dir_okay=True), required=False, default=None,
              help="input dataset (only needed for --data_pose_mode / --uncoupled_mode; cameras otherwise come from the checkpoint)")
@click.option("--data_downsample_factor", type=click.FloatRange(min=1.0), default=3.0, show_default=True)
@click.option("--white_bkgd", type=click.BOOL, default=True, show_default=True)
@click.option("--train_num_samples_per_ray", type=click.INT, default=256, show_default=True)
@click.option("--render_num_samples_per_ray", type=click.INT, default=512, show_default=True)
@click.option("--num_iterations_edit", type=click.INT, default=8000, show_default=True)
@click.option("--learning_rate", type=click.FLOAT, default=0.03, show_default=True)
@click.option("--lr_freq", type=click.INT, default=400, show_default=True)
@click.option("--lr_decay_start", type=click.INT, default=5000, show_default=True)
@click.option("--lr_gamma", type=click.FLOAT, default=0.96, show_default=True)
@click.option("--save_frequency", type=click.INT, default=500, show_default=True)
@click.option("--feedback_frequency", type=click.INT, default=200, show_default=True)
@click.option("--summary_frequency", type=click.INT, default=50, show_default=True)
@click.option("--do_sds", type=click.BOOL, default=True, show_default=True)
@click.option("--new_frame_frequency", type=click.INT, default=1, show_default=True)
@click.option("--density_correlation_weight", type=click.FLOAT, default=200.0, show_default=True)
@click.option("--feature_correlation_weight", type=click.FLOAT, default=0.0, show_default=True)
@click.option("--tv_density_weight", type=click.FLOAT, default=0.0, show_default=True)
@click.option("--tv_features_weight", type=click.FLOAT, default=0.0, show_default=True)
@click.option("--sds_t_freq", type=click.INT, default=600, show_default=True)
@click.option("--sds_t_start", type=click.INT, default=4000, show_default=True)
@click.option("--sds_t_gamma", type=click.FLOAT, default=0.75, show_default=True)
@click.option("--uncoupled_mode", type=click.BOOL, default=False, show_default=True)
@click.option("--data_pose_mode", type=click.BOOL, default=False, show_default=True)
@click.option("--do_refinement", type=click.BOOL, default=False, show_default=True)
@click.option("--post_process_scc", type=click.BOOL, default=False, show_default=True,
              help="restore the original densities outside the largest connected component of the edited field")
@click.option("-eidx", "--edit_idx", type=click.STRING, default=None, help="refinement: 1-based token indices of the edit words")
@click.option("-oidx", "--object_idx", type=click.INT, default=None, help="refinement: token index of the object")
@click.option("-t", "--timestamp", type=click.INT, default=200, show_default=True, help="refinement: diffusion timestamp")
@click.option("-a", "--hf_auth_token", type=click.STRING, default="", help="refinement: hugging face token (SD 1.4)")
@click.option("--num_iterations_refine", type=click.INT, default=1500, show_default=True)
@click.option("--learning_rate_refine", type=click.FLOAT, default=None,
              help="alias of --learning_rate_attn_learning (takes precedence when given)")
@click.option("--attn_tv_weight", type=click.FLOAT, default=0.01, show_default=True)
@click.option("--kval", type=click.FLOAT, default=5.0, show_default=True)
@click.option("--edit_mask_thresh", type=click.FLOAT, default=0.992, show_default=True)
@click.option("--num_obj_voxels_thresh", type=click.INT, default=5000, show_default=True)
@click.option("--min_num_edit_voxels", type=click.INT, default=300, show_default=True)
@click.option("--top_k_edit_thresh", type=click.INT, default=300, show_default=True)
@click.option("--top_k_obj_thresh", type=click.INT, default=200, show_default=True)
@click.option("--downsample_refine_grid", type=click.BOOL, default=False, show_default=True)
@click.option("--uncoupled_l2_mode", type=click.BOOL, default=False, show_default=True)
@click.option("--l2_mode", type=click.BOOL, default=False, show_default=True)
@click.option("--l1_mode", type=click.BOOL, default=False, show_default=True)
@click.option("--log_wandb", type=click.BOOL, default=False, show_default=True)
@click.option("--learning_rate_attn_learning", type=click.FLOAT, default=0.035, show_default=True,
              help="learning rate of the attention-grid refinement (the reference's option)")
@click.option("--precise_grad", type=click.BOOL, default=False, show_default=True,
              help="(not in the reference) density gradients at the accuracy of a double-precision backward "
                   "(VoxeDispatch::precise_grad: ~2e-6 instead of ~1e-5 median relative error, ~14 % slower render step)")
@accepted_options(COMPAT_ONLY)
def main(**kwargs) -> None:
    if kwargs.pop("precise_grad", False):
        import contextlib

        from voxe_hip import dispatch as _dispatch

        with contextlib.ExitStack() as stack:   # every render of this run (resolved per render call, pinned for its backward)
            stack.enter_context(_dispatch.override(precise_grad=1))
            return _main(**kwargs)
    return _main(**kwargs)


def _main(**kwargs) -> None:
    cfg = type("Config", (), kwargs)
    report_unused(kwargs, COMPAT_ONLY, log)
    if cfg.learning_rate_refine is None:
        cfg.learning_rate_refine = cfg.learning_rate_attn_learning
    if cfg.do_refinement and not cfg.edit_idx:
        raise click.UsageError("--do_refinement needs --edit_idx (token indices of the edit words in the prompt)")
    device = torch.device("cuda")
    output_path = Path(cfg.output_path)
    log_config_to_disk(kwargs, output_path)
    ref_vol_mod, extra = create_volumetric_model_from_saved_model(Path(cfg.ref_model_path), create_voxel_grid_from_saved_info_dict, device=device)
    ref_vol_mod.render_config.num_samples_per_ray = cfg.train_num_samples_per_ray
    ref_vol_mod.render_config.render_num_samples_per_ray = cfg.render_num_samples_per_ray
    ref_vol_mod.render_config.white_bkgd = cfg.white_bkgd
    sds_vol_mod = copy.deepcopy(ref_vol_mod)
    dataset = None
    if cfg.data_path is not None:
        from thre3d_atom.data.datasets import PosedImagesDataset

        dataset = PosedImagesDataset(Path(cfg.data_path) / "train", Path(cfg.data_path) / "train_camera_params.json",
                                     downsample_factor=cfg.data_downsample_factor, rgba_white_bkgd=cfg.white_bkgd)
    intrinsics = edit_stage_intrinsics(extra[CAMERA_INTRINSICS], dataset, cfg.data_downsample_factor)
    train_sh_vox_grid_vol_mod_with_posed_images_and_sds(
        sds_vol_mod=sds_vol_mod, pretrained_vol_mod=ref_vol_mod, train_dataset=dataset, image_dims=None,
        output_dir=output_path, num_iterations=cfg.num_iterations_edit, learning_rate=cfg.learning_rate,
        lr_decay_start=cfg.lr_decay_start, lr_freq=cfg.lr_freq, lr_gamma=cfg.lr_gamma, save_freq=cfg.save_frequency,
        feedback_freq=cfg.feedback_frequency, summary_freq=cfg.summary_frequency, sds_prompt=cfg.prompt,
        new_frame_frequency=cfg.new_frame_frequency, density_correlation_weight=cfg.density_correlation_weight,
        feature_correlation_weight=cfg.feature_correlation_weight, tv_density_weight=cfg.tv_density_weight,
        tv_features_weight=cfg.tv_features_weight, do_sds=cfg.do_sds, sds_t_freq=cfg.sds_t_freq,
        sds_t_start=cfg.sds_t_start, sds_t_gamma=cfg.sds_t_gamma, uncoupled_mode=cfg.uncoupled_mode,
        data_pose_mode=cfg.data_pose_mode, camera_intrinsics=intrinsics, camera_bounds=extra[CAMERA_BOUNDS],
        saved_hemispherical_radius=extra.get(HEMISPHERICAL_RADIUS, 4.0311), uncoupled_l2_mode=cfg.uncoupled_l2_mode,
        l2_mode=cfg.l2_mode, l1_mode=cfg.l1_mode, log_wandb=cfg.log_wandb,
    )
    saved = output_path / "saved_models"
    extra_info = {CAMERA_BOUNDS: extra[CAMERA_BOUNDS], CAMERA_INTRINSICS: intrinsics,
                  HEMISPHERICAL_RADIUS: extra.get(HEMISPHERICAL_RADIUS, 4.0311)}
    final_name = "model_final.pth"
    if cfg.do_refinement:
        attn_models = [create_volumetric_model_from_saved_model_attn(
            saved / "model_final.pth", create_voxel_grid_from_saved_info_dict_attn, device=device)[0] for _ in range(3)]
        refine_edited_relu_field(
            vol_mod_edit=attn_models[0], vol_mod_object=attn_models[1], vol_mod_output=attn_models[2],
            vol_mod_ref=ref_vol_mod, train_dataset=dataset, hf_auth_token=cfg.hf_auth_token, output_dir=output_path,
            prompt=cfg.prompt, edit_idx=[int(i) for i in cfg.edit_idx.split()], object_idx=cfg.object_idx,
            timestamp=cfg.timestamp, image_dims=None, num_iterations=cfg.num_iterations_refine,
            learning_rate=cfg.learning_rate_refine, save_freq=cfg.save_frequency, feedback_freq=cfg.feedback_frequency,
            summary_freq=cfg.summary_frequency, attn_tv_weight=cfg.attn_tv_weight, kval=cfg.kval,
            edit_mask_thresh=cfg.edit_mask_thresh, num_obj_voxels_thresh=cfg.num_obj_voxels_thresh,
            min_num_edit_voxels=cfg.min_num_edit_voxels, top_k_edit_thresh=cfg.top_k_edit_thresh,
            top_k_obj_thresh=cfg.top_k_obj_thresh, data_pose_mode=cfg.data_pose_mode,
            downsample_refine_grid=cfg.downsample_refine_grid, camera_intrinsics=intrinsics,
            camera_bounds=extra[CAMERA_BOUNDS], saved_hemispherical_radius=extra.get(HEMISPHERICAL_RADIUS, 4.0311),
        )
        final_name = "model_final_refined.pth"
    if cfg.post_process_scc:
        loader = create_volumetric_model_from_saved_model_attn if cfg.do_refinement else create_volumetric_model_from_saved_model
        creator = create_voxel_grid_from_saved_info_dict_attn if cfg.do_refinement else create_voxel_grid_from_saved_info_dict
        vol_mod, _ = loader(saved / final_name, creator, device=device)
        restore_outside_largest_component(vol_mod, ref_vol_mod, k=10)
        torch.save(vol_mod.get_save_info(extra_info=extra_info), saved / final_name)


if __name__ == "__main__":
    main()
