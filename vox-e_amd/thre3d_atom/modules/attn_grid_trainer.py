"""Local-edit refinement: fit two attention grids ("edit" token / rest of the object) to diffusion cross-attention
maps, graph-cut them into an edit region and splice the edited field into the original one.

Entry point and loop structure of the reference's thre3d_atom/modules/attn_grid_trainer.py
(`refine_edited_relu_field` :63-627, direction helpers :629-681, `_tv_loss_on_grid` :659-663).  Per iteration:
random camera -> no-grad RGB render of the edited field (HIP forward) -> cross-attention maps from the guidance
object (Stable Diffusion under PyTorch-ROCm; boundary only) -> differentiable attention renders of both grids (HIP
forward/backward, 1-channel kernel) -> masked L1 + total variation (HIP) -> fused HIP Adam.  The closing graph
cut runs on the GPU (voxe_graphcut) instead of PyMaxflow python loops.
"""
import time
from datetime import timedelta
from pathlib import Path
from typing import Any, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from thre3d_atom.modules.optim import FusedGridAdam, VoxeAdam
from thre3d_atom.modules.refinement_functions import calc_loss_on_attn_grid, get_edit_region
from thre3d_atom.modules.volumetric_model import VolumetricModel
from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays
from thre3d_atom.thre3d_reprs.renderers import attn_render_params, render_sh_voxel_grid_attn
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, CameraPose, get_random_pose, to8b
from thre3d_atom.utils.logging import log
from voxe_hip import ops as _ops

HEMISPHERICAL_RADIUS_CONSTANT = 4.0311
dir_to_num_dict = {"side": 0, "overhead": 1, "back": 2, "front": 3}


def _tv_loss_on_grid(grid: Tensor) -> Tensor:
    return _ops.tv_loss_on_grid(grid)


def _pitch_yaw_from_Rt(rotation: Tensor):  # noqa: N802 (reference name)
    tx, ty, tz = rotation[:, -1].cpu().numpy()
    pitch = np.arctan(tz / np.sqrt(tx ** 2 + ty ** 2)) * 180 / np.pi
    yaw = np.arccos(float(rotation[0, 0])) * 180.0 / np.pi
    return pitch, yaw


def _label(pitch: float, yaw: float, side_from: float) -> str:
    label = "front"
    if yaw > side_from:
        label = "side"
    if yaw > 120.0:
        label = "back"
    if pitch > 55.0:
        label = "overhead"
    return label


def get_dir_batch_from_poses(poses: Tensor):
    """view-direction words of dataset poses [B,3,4]; "side" starts at 60 degrees of yaw (:629-647)"""
    return [_label(*_pitch_yaw_from_Rt(p), side_from=60.0) for p in poses]


def _get_dir_batch_from_poses(poses: Tensor):
    """same with "side" starting at 45 degrees (:666-681) -- the variant the training loop uses"""
    return [_label(*_pitch_yaw_from_Rt(p), side_from=45.0) for p in poses]


def split_attention_maps(maps: Sequence[Tensor], edit_idx: Sequence[int], object_idx: Optional[int]):
    """edit map = max over the edit tokens, object map = max over all other tokens (or the given one); token
    indices are 1-based positions in the prompt (:313-329)."""
    edit_map = torch.stack([maps[i - 1] for i in edit_idx], dim=-1).max(dim=-1).values.squeeze()
    if object_idx is None:
        rest = [maps[i - 1] for i in range(1, len(maps) + 1) if i not in edit_idx]
        object_map = torch.stack(rest, dim=-1).max(dim=-1).values.squeeze()
    else:
        object_map = maps[object_idx - 1]
    return edit_map, object_map


def _save_map(path: Path, image: Tensor) -> None:
    """attention map / render as an 8-bit grey PNG (min-max normalised)"""
    a = image.detach().float().cpu().numpy()
    a = (a - a.min()) / max(float(a.max() - a.min()), 1e-12)
    try:
        from PIL import Image

        Image.fromarray(np.uint8(255 * a)).save(path)
    except ImportError:  # pragma: no cover
        np.save(path.with_suffix(".npy"), a)


def refine_edited_relu_field(
    vol_mod_edit: VolumetricModel,
    vol_mod_object: VolumetricModel,
    vol_mod_output: VolumetricModel,
    vol_mod_ref: VolumetricModel,
    train_dataset: Any,                  # None is allowed when camera_intrinsics / camera_bounds are given
    hf_auth_token: str,
    output_dir: Path,
    prompt: str,
    edit_idx: Sequence[int],
    timestamp: int,
    image_dims: Optional[tuple],
    image_batch_cache_size: int = 8,
    num_workers: int = 4,
    object_idx: Optional[int] = None,
    num_iterations: int = 2000,
    ray_batch_size: int = 32768,
    scale_factor: float = 2.0,
    learning_rate: float = 0.03,
    lr_decay_gamma_per_stage: float = 0.1,
    lr_decay_steps_per_stage: int = 2000,
    render_feedback_pose: Optional[CameraPose] = None,
    data_pose_mode: bool = False,
    save_freq: int = 1000,
    feedback_freq: int = 100,
    summary_freq: int = 10,
    apply_diffuse_render_regularization: bool = False,
    verbose_rendering: bool = True,
    attn_tv_weight: float = 0.001,
    kval: float = 5.0,
    edit_mask_thresh: float = 0.992,
    num_obj_voxels_thresh: int = 5000,
    min_num_edit_voxels: int = 300,
    top_k_edit_thresh: int = 300,
    top_k_obj_thresh: int = 200,
    log_wandb: bool = False,
    downsample_refine_grid: bool = False,
    # --- additions of this build -----------------------------------------------------------------
    attn_guidance: Any = None,           # object with get_num_tokens(prompt) / get_attn_map(prompt=, pred_rgb=, ...)
    camera_intrinsics: Optional[CameraIntrinsics] = None,
    camera_bounds: Optional[CameraBounds] = None,
    hemispherical_radius: float = HEMISPHERICAL_RADIUS_CONSTANT,   # distance of the random cameras: the reference hard-codes
    # 4.0311 (attn_grid_trainer.py:54,277); pass another value only as an explicit override
    saved_hemispherical_radius: Optional[float] = None,            # radius estimate stored in the checkpoints (no dataset)
    fused_grid_step: bool = True,        # FusedGridAdam on the attention grids (gradient stays in the kernels' workspace)
) -> VolumetricModel:
    """Optimise the attention grids of `vol_mod_edit` / `vol_mod_object` (copies of the SDS-edited field), cut the
    edit region and write the refined field into `vol_mod_output` (returned).  Checkpoints as in the reference:
    saved_models/model_final_attn_edit.pth, model_final_attn_object.pth, model_final_refined.pth."""
    for vm in (vol_mod_edit, vol_mod_object, vol_mod_output):
        if not isinstance(vm.thre3d_repr, VoxelGrid) or vm._render_procedure_attn != render_sh_voxel_grid_attn:
            raise AssertionError("this procedure needs SH-based VoxelGrid models with the attention render procedure")
    if prompt == "none":
        raise AssertionError("sorry, you have to supply a text prompt to use SDS")
    if train_dataset is not None:
        camera_intrinsics = train_dataset.camera_intrinsics      # the dataset's, like the reference (:162-166)
        camera_bounds = camera_bounds or train_dataset.camera_bounds
        extra_radius = train_dataset.get_hemispherical_radius_estimate()
    else:
        extra_radius = hemispherical_radius if saved_hemispherical_radius is None else saved_hemispherical_radius
        if data_pose_mode:
            raise ValueError("data_pose_mode needs a dataset of posed images")
    if camera_intrinsics is None or camera_bounds is None:
        raise ValueError("camera_intrinsics and camera_bounds are required without a dataset")
    if attn_guidance is None:
        from thre3d_atom.thre3d_reprs.sd import StableDiffusion

        attn_guidance = StableDiffusion(vol_mod_edit.device, "1.4", auth_token=hf_auth_token)
    if not (hasattr(attn_guidance, "get_attn_map") and hasattr(attn_guidance, "get_num_tokens")):
        raise TypeError("`attn_guidance` needs get_num_tokens(prompt) and get_attn_map(prompt=, pred_rgb=, timestamp=, "
                        "indices_to_fetch=) -> (list of [H, W] maps, aux)")
    device = vol_mod_edit.device
    im_h, im_w = (int(v) for v in (image_dims if image_dims is not None else camera_intrinsics[:2]))

    output_dir = Path(output_dir)
    model_dir, render_dir = output_dir / "saved_models", output_dir / "training_logs" / "rendered_output"
    for d in (model_dir, render_dir):
        d.mkdir(exist_ok=True, parents=True)
    extra_info = {CAMERA_BOUNDS: camera_bounds, CAMERA_INTRINSICS: camera_intrinsics, HEMISPHERICAL_RADIUS: extra_radius}

    edit_grid, object_grid = vol_mod_edit.thre3d_repr, vol_mod_object.thre3d_repr
    if fused_grid_step:
        optimizer_edit = FusedGridAdam(edit_grid, lr=learning_rate, betas=(0.9, 0.999), kind="attn")
        optimizer_object = FusedGridAdam(object_grid, lr=learning_rate, betas=(0.9, 0.999), kind="attn")
    else:
        optimizer_edit = VoxeAdam([{"params": [edit_grid.attn], "lr": learning_rate}], betas=(0.9, 0.999))
        optimizer_object = VoxeAdam([{"params": [object_grid.attn], "lr": learning_rate}], betas=(0.9, 0.999))
    lr_scheduler_edit = torch.optim.lr_scheduler.ExponentialLR(optimizer_edit, gamma=lr_decay_gamma_per_stage)

    log.info(f"voxel grid resolution: {edit_grid.grid_dims} training images resolution: [{im_h} x {im_w}]")
    trained_time, last = 0.0, time.perf_counter()
    data_cursor = 0
    pose = None
    step_losses = step_renders = None      # device buffers of the fused step (masked L1 / TV per grid, rendered attention images)
    try:
        for global_step in range(1, num_iterations + 1):
            if data_pose_mode:
                _, pose_mat, _ = train_dataset[data_cursor % len(train_dataset)]
                data_cursor += 1
                pose = CameraPose(rotation=pose_mat[:, :3], translation=pose_mat[:, 3:])
                direction = _get_dir_batch_from_poses(pose_mat[None])[0]
            else:
                pose, direction, _, _ = get_random_pose(hemispherical_radius)
            rays_batch = flatten_rays(cast_rays(camera_intrinsics, pose, device=device))

            # RGB render of the edited field -> cross-attention maps of the prompt tokens (boundary to the UNet)
            rgb = vol_mod_edit.render(pose, camera_intrinsics).colour
            out_imgs = rgb.unsqueeze(0).permute(0, 3, 1, 2).to(device)
            m_prompt = prompt + f", {direction} view"
            num_tokens = attn_guidance.get_num_tokens(m_prompt)
            maps, _ = attn_guidance.get_attn_map(prompt=m_prompt, pred_rgb=out_imgs, timestamp=timestamp,
                                                 indices_to_fetch=list(range(1, num_tokens + 1)))
            edit_attn_map, object_attn_map = split_attention_maps(maps, edit_idx, object_idx)

            if fused_grid_step:
                # everything behind the UNet's maps is grid work: per attention grid ONE library call (voxe_attn_refine_step:
                # attention render -> masked L1 + TV -> backward -> Adam), the reference's lines :335-378 with the same arithmetic
                num_rays = rays_batch.origins.shape[0]
                if step_losses is None:
                    step_losses = torch.zeros((2, 2), dtype=torch.float32, device=device)
                    step_renders = torch.empty((2, num_rays), dtype=torch.float32, device=device)
                for i, (vm, opt, amap) in enumerate(((vol_mod_edit, optimizer_edit, edit_attn_map),
                                                     (vol_mod_object, optimizer_object, object_attn_map))):
                    params = attn_render_params(vm.thre3d_repr, rays_batch, vm.render_config)
                    opt.attention_refinement_step(params, rays_batch.origins, rays_batch.directions, amap.reshape(-1),
                                                  attn_tv_weight, step_losses[i], rng=_ops._next_rng(), attn_render=step_renders[i])
                edit_attn_loss, object_attn_loss = step_losses[0, 0], step_losses[1, 0]
                edit_render, object_render = step_renders[0], step_renders[1]
            else:
                edit_render = vol_mod_edit.render_rays_attn(rays_batch).attn
                object_render = vol_mod_object.render_rays_attn(rays_batch).attn
                edit_attn_loss = calc_loss_on_attn_grid(edit_render, edit_attn_map, token="edit", global_step=global_step)
                object_attn_loss = calc_loss_on_attn_grid(object_render, object_attn_map, token="object", global_step=global_step)
                total_loss_edit = edit_attn_loss + _tv_loss_on_grid(edit_grid.attn) * attn_tv_weight
                total_loss_object = object_attn_loss + _tv_loss_on_grid(object_grid.attn) * attn_tv_weight

                total_loss_edit.backward()
                optimizer_edit.step()
                optimizer_edit.zero_grad()
                total_loss_object.backward()
                optimizer_object.step()
                optimizer_object.zero_grad()
            trained_time += time.perf_counter() - last

            if global_step % summary_freq == 0 or global_step in (1, num_iterations):
                log.info(f"Global Iteration: {global_step} attn_loss: {float(edit_attn_loss.detach()): .3f} "
                         f"object_attn_loss: {float(object_attn_loss.detach()): .3f}")
            if global_step % lr_decay_steps_per_stage == 0:
                lr_scheduler_edit.step()
                log.info(f"Adjusted learning rate | learning rates: {[g['lr'] for g in optimizer_edit.param_groups]}")
            if global_step % feedback_freq == 0 or global_step in (1, num_iterations):
                log.info(f"TIME CHECK: time spent actually training till now: {timedelta(seconds=trained_time)}")
                with torch.no_grad():
                    _save_map(render_dir / f"edit_gt_attn_{global_step}.png", edit_attn_map)
                    _save_map(render_dir / f"object_gt_attn_{global_step}.png", object_attn_map)
                    _save_map(render_dir / f"edit_render_attn_{global_step}.png", edit_render.reshape(edit_attn_map.shape))
                    _save_map(render_dir / f"object_render_attn_{global_step}.png", object_render.reshape(edit_attn_map.shape))
                if global_step % save_freq == 0 or global_step in (1, num_iterations):
                    torch.save(vol_mod_edit.get_save_info(extra_info=extra_info), model_dir / f"model_edit_iter_{global_step}.pth")
                    torch.save(vol_mod_object.get_save_info(extra_info=extra_info), model_dir / f"model_object_iter_{global_step}.pth")
            last = time.perf_counter()
    finally:
        if fused_grid_step:   # (also when the loop raised: the deferred-gradient mode must not outlive its optimiser)
            optimizer_edit.detach()
            optimizer_object.detach()
    # ---- graph cut and splice ----------------------------------------------------------------------------
    log.info("Starting Grid Refinement!")
    t0 = time.perf_counter()
    get_edit_region(vol_mod_edit=vol_mod_edit, vol_mod_object=vol_mod_object, vol_mod_output=vol_mod_output,
                    K=kval, edit_mask_thresh=edit_mask_thresh, num_obj_voxels_thresh=num_obj_voxels_thresh,
                    min_num_edit_voxels=min_num_edit_voxels, top_k_edit_thresh=top_k_edit_thresh,
                    top_k_obj_thresh=top_k_obj_thresh, downsample_grid=downsample_refine_grid)
    splice_reference_outside_edit_region(vol_mod_output, vol_mod_ref)
    log.info(f"graph cut + splice: {time.perf_counter() - t0:.3f} s")

    feedback_pose = render_feedback_pose or pose
    if feedback_pose is not None:
        out = vol_mod_output.render(feedback_pose, camera_intrinsics,
                                    num_samples_per_ray=vol_mod_output.render_config.render_num_samples_per_ray)
        try:
            from PIL import Image

            Image.fromarray(to8b(out.colour.cpu().numpy())).save(render_dir / "sds_refined.png")
        except ImportError:  # pragma: no cover
            np.save(render_dir / "sds_refined.npy", out.colour.cpu().numpy())

    log.info("Saving the final model-snapshot :)! Almost there ... yay!")
    torch.save(vol_mod_edit.get_save_info(extra_info=extra_info), model_dir / "model_final_attn_edit.pth")
    torch.save(vol_mod_object.get_save_info(extra_info=extra_info), model_dir / "model_final_attn_object.pth")
    torch.save(vol_mod_output.get_save_info(extra_info=extra_info), model_dir / "model_final_refined.pth")
    log.info("Training complete")
    log.info(f"Total actual training time: {timedelta(seconds=trained_time)}")
    return vol_mod_output


def splice_reference_outside_edit_region(vol_mod_output: VolumetricModel, vol_mod_ref: VolumetricModel) -> None:
    """Every voxel whose keep-grid value is non-zero (not in the edit region) takes the density and features of the
    un-edited reference field (attn_grid_trainer.py:539-550)."""
    out_repr, ref_repr = vol_mod_output.thre3d_repr, vol_mod_ref.thre3d_repr
    with torch.no_grad():
        keep = out_repr.attn.detach() != 0
        dens = torch.where(keep, ref_repr._densities.detach().to(keep.device), out_repr._densities.detach())
        feat = torch.where(keep, ref_repr._features.detach().to(keep.device), out_repr._features.detach())
    out_repr._densities = torch.nn.Parameter(dens)
    out_repr._features = torch.nn.Parameter(feat)
