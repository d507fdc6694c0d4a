#!/usr/bin/env python3
"""Segment an SDS-edited field with two optimised attention grids and splice the un-edited field back outside the edit
region (entry point kept from the reference's segment_attn_relu_field.py:105-290; same option names).

This is the last stage of `refine_edited_relu_field` run on its own: `get_edit_region` (voxel graph built and cut on
the GPU: voxe_graph_build / voxe_graphcut) writes the keep-grid into the output model, every voxel outside the edit
region takes the density and features of the reference field, and the result is saved as
`saved_models/model_final_refined.pth`.  Feedback renders are PNG stills under training_logs/rendered_output."""
import os
import sys
from pathlib import Path

import click
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "vox-e_amd"))

from thre3d_atom.data.datasets import PosedImagesDataset  # noqa: E402
from thre3d_atom.modules.attn_grid_trainer import splice_reference_outside_edit_region  # noqa: E402
from thre3d_atom.modules.refinement_functions import get_edit_region  # noqa: E402
from thre3d_atom.modules.volumetric_model import (  # noqa: E402
    create_volumetric_model_from_saved_model,
    create_volumetric_model_from_saved_model_attn,
)
from thre3d_atom.thre3d_reprs.voxels import (  # noqa: E402
    create_voxel_grid_from_saved_info_dict,
    create_voxel_grid_from_saved_info_dict_attn,
)
from thre3d_atom.utils.cli_compat import accepted_options, report_unused  # noqa: E402
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS  # noqa: E402
from thre3d_atom.utils.imaging_utils import CameraPose, to8b  # noqa: E402
from thre3d_atom.utils.logging import log  # noqa: E402
from thre3d_atom.utils.misc import log_config_to_disk  # noqa: E402

COMPAT_ONLY = [
    ("--log_wandb", click.BOOL, False, 1), ("--wandb_username", click.STRING, "etaisella", 1),
    ("--wandb_project_name", click.STRING, "Vox-E-refine", 1),
]


@click.command()
@click.option("-d", "--data_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path to the input dataset")
@click.option("-ie", "--edit_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="model with the optimised EDIT attention grid")
@click.option("-io", "--object_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="model with the optimised OBJECT attention grid")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path for the output")
@click.option("-r", "--ref_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the pre-trained (un-edited) model")
@click.option("-i", "--sds_model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the SDS-edited model")
@click.option("--separate_train_test_folders", type=click.BOOL, default=True, show_default=True)
@click.option("--data_downsample_factor", type=click.FloatRange(min=1.0), default=3.0, show_default=True)
@click.option("--downsample_refine_grid", type=click.BOOL, default=False, show_default=True)
@click.option("--kval", type=click.FLOAT, default=5.0, show_default=True)
@click.option("--edit_mask_thresh", type=click.FLOAT, default=0.992, show_default=True)
@click.option("--num_obj_voxels_thresh", type=click.INT, default=5000, show_default=True)
@click.option("--min_num_edit_voxels", type=click.INT, default=300, show_default=True)
@click.option("--top_k_edit_thresh", type=click.INT, default=300, show_default=True)
@click.option("--top_k_obj_thresh", type=click.INT, default=200, show_default=True)
@accepted_options(COMPAT_ONLY)
def main(**kwargs) -> None:
    cfg = type("Config", (), kwargs)
    report_unused(kwargs, COMPAT_ONLY, log)
    device = torch.device("cuda")
    output_path, data_path = Path(cfg.output_path), Path(cfg.data_path)
    log_config_to_disk(kwargs, output_path)
    separate = cfg.separate_train_test_folders
    # (the reference reads the views at full resolution here, whatever --data_downsample_factor says: :123-138)
    dataset = PosedImagesDataset(data_path / ("train" if separate else "images"),
                                 data_path / ("train_camera_params.json" if separate else "camera_params.json"),
                                 normalize_scene_scale=False, downsample_factor=1.0, rgba_white_bkgd=True)
    vol_mod_ref, _ = create_volumetric_model_from_saved_model(Path(cfg.ref_model_path), create_voxel_grid_from_saved_info_dict, device=device)
    vol_mod_edit, _ = create_volumetric_model_from_saved_model_attn(
        Path(cfg.edit_model_path), create_voxel_grid_from_saved_info_dict_attn, device=device, load_attn=True)
    vol_mod_obj, _ = create_volumetric_model_from_saved_model_attn(
        Path(cfg.object_model_path), create_voxel_grid_from_saved_info_dict_attn, device=device, load_attn=True)
    vol_mod_output, _ = create_volumetric_model_from_saved_model_attn(
        Path(cfg.sds_model_path), create_voxel_grid_from_saved_info_dict_attn, device=device)
    model_dir, render_dir = output_path / "saved_models", output_path / "training_logs" / "rendered_output"
    for directory in (model_dir, render_dir):
        directory.mkdir(exist_ok=True, parents=True)

    log.info("Starting Grid Refinement!")
    get_edit_region(vol_mod_edit=vol_mod_edit, vol_mod_object=vol_mod_obj, vol_mod_output=vol_mod_output, K=cfg.kval,
                    edit_mask_thresh=cfg.edit_mask_thresh, num_obj_voxels_thresh=cfg.num_obj_voxels_thresh,
                    min_num_edit_voxels=cfg.min_num_edit_voxels, top_k_edit_thresh=cfg.top_k_edit_thresh,
                    top_k_obj_thresh=cfg.top_k_obj_thresh, downsample_grid=cfg.downsample_refine_grid)
    splice_reference_outside_edit_region(vol_mod_output, vol_mod_ref)

    pose = CameraPose(rotation=dataset.poses[0][:, :3].to(device), translation=dataset.poses[0][:, 3:].to(device))
    from PIL import Image

    for name, rendered in (("sds_refined", vol_mod_output.render(pose, dataset.camera_intrinsics, gpu_render=True).colour),
                           ("attn_final", vol_mod_output.render_attn(pose, dataset.camera_intrinsics, gpu_render=True).attn)):
        img = rendered.cpu().numpy()
        if img.shape[-1] == 1:   # the keep-grid render: 0 in the edit region, negative elsewhere -> grey levels
            lo, hi = float(img.min()), float(img.max())
            img = ((img - lo) / (hi - lo) if hi > lo else img * 0.0).repeat(3, axis=-1)
        Image.fromarray(to8b(img)).save(render_dir / f"{name}_0.png")

    log.info("Saving the final model-snapshot")
    torch.save(vol_mod_output.get_save_info(extra_info={
        CAMERA_BOUNDS: dataset.camera_bounds, CAMERA_INTRINSICS: dataset.camera_intrinsics,
        HEMISPHERICAL_RADIUS: dataset.get_hemispherical_radius_estimate()}), model_dir / "model_final_refined.pth")


if __name__ == "__main__":
    main()
