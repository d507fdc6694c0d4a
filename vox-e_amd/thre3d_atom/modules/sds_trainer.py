"""Score-distillation editing loop (global edit stage).

Keeps the entry point and the loss helpers of the reference's thre3d_atom/modules/sds_trainer.py
(`train_sh_vox_grid_vol_mod_with_posed_images_and_sds` :47-469, `density_correlation_loss_fn` :494-505,
`_density_correlation_loss` :507-524, `_tv_loss_on_grid` :563-567, `_get_dir_batch_from_poses` :542-561);
per step: random hemispherical pose -> whole-image differentiable render (fused HIP forward/backward) ->
guidance loss (Stable Diffusion under PyTorch-ROCm, or any object with `training_step`) + density-correlation
regulariser against the frozen reference grid (HIP) -> fused HIP Adam.
"""
import time
from datetime import timedelta
from pathlib import Path
from typing import Any, Optional

import numpy as np
import torch
from torch import Tensor

from thre3d_atom.modules import parallel
from thre3d_atom.modules.optim import FusedGridAdam, VoxeAdam
from thre3d_atom.modules.volumetric_model import VolumetricModel
from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays
from thre3d_atom.thre3d_reprs.renderers import render_sh_voxel_grid
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, CameraPose, get_random_pose, to8b
from thre3d_atom.utils.logging import log
from voxe_hip import ops as _ops

dir_to_num_dict = {"side": 0, "overhead": 1, "back": 2, "front": 3}
HEMISPHERICAL_RADIUS_CONSTANT = 4.0311


def _density_correlation_loss(sds_density: Tensor, regular_density: Tensor):
    """1 - mean((a - mean a)(b - mean b)) / (sqrt(var a var b) + 1e-7), fused forward+gradient on the GPU.
    The reference also returns the detached per-voxel correlation grid for an (unused by default)
    feature regulariser; it is produced lazily by `correlation_grid` to keep the step free of extra passes."""
    return _ops.density_correlation_loss(sds_density, regular_density), None


def density_correlation_loss_fn(sds_density: Tensor, regular_density: Tensor, l2_mode: bool = False, l1_mode: bool = False):
    """sds_trainer.py:494-505 of the reference: mse_loss / l1_loss of the two grids with l2_mode / l1_mode (l2 first), else the
    correlation -- all three as HIP passes (value + gradient), differentiable w.r.t. `sds_density`"""
    if l2_mode or l1_mode:
        return _ops.density_diff_loss(sds_density, regular_density, l2_mode=bool(l2_mode)), None
    return _density_correlation_loss(sds_density, regular_density)


def _tv_loss_on_grid(grid: Tensor) -> Tensor:
    return _ops.tv_loss_on_grid(grid)


def _feature_correlation_loss(sds_features: Tensor, regular_features: Tensor, density_cov_grid=None) -> Tensor:
    """sum over voxels of (sum_c sigmoid(f_sds) - sigmoid(f_ref))^2 (sds_trainer.py:526-534; weight 0 by default): one HIP
    pass for value + gradient"""
    return _ops.feature_correlation_loss(sds_features, regular_features.detach())


def _pitch_yaw_from_Rt(pose: Tensor):  # noqa: N802 (reference name)
    tx, ty, tz = pose[:, -1].cpu().numpy()
    pitch = np.arctan(tz / np.sqrt(tx ** 2 + ty ** 2)) * 180 / np.pi
    yaw = np.arccos(float(pose[0, 0])) * 180.0 / np.pi
    return pitch, yaw


def _get_dir_batch_from_poses(poses: Tensor):
    """view-direction prompt suffix of dataset poses [B,3,4]"""
    out = []
    for pose in poses:
        pitch, yaw = _pitch_yaw_from_Rt(pose)
        label = "front"
        if yaw > 45.0:
            label = "side"
        if yaw > 120.0:
            label = "back"
        if pitch > 55.0:
            label = "overhead"
        out.append(label)
    return out


def train_sh_vox_grid_vol_mod_with_posed_images_and_sds(
    sds_vol_mod: VolumetricModel,
    pretrained_vol_mod: VolumetricModel,
    train_dataset: Any,                 # None is allowed when camera_intrinsics / camera_bounds are given
    image_dims: Optional[tuple],
    output_dir: Path,
    image_batch_cache_size: int = 8,
    ray_batch_size: int = 32768,
    num_iterations: int = 2000,
    scale_factor: float = 2.0,
    learning_rate: float = 0.03,
    lr_decay_start: int = 5000,
    lr_freq: int = 400,
    lr_gamma: float = 0.8,
    render_feedback_pose: Optional[CameraPose] = None,
    save_freq: int = 1000,
    feedback_freq: int = 100,
    summary_freq: int = 10,
    apply_diffuse_render_regularization: bool = True,
    num_workers: int = 4,
    verbose_rendering: bool = True,
    sds_prompt: str = "none",
    new_frame_frequency: int = 1,
    density_correlation_weight: float = 0.0,
    feature_correlation_weight: float = 0.0,
    tv_density_weight: float = 0.0,
    tv_features_weight: float = 0.0,
    do_sds: bool = True,
    sds_t_freq: int = 200,
    sds_t_start: int = 1500,
    sds_t_gamma: float = 1.0,
    uncoupled_mode: bool = False,
    data_pose_mode: bool = False,
    uncoupled_l2_mode: bool = False,
    log_wandb: bool = False,
    l2_mode: bool = False,
    l1_mode: bool = False,
    # --- additions of this build -----------------------------------------------------------------
    guidance: Any = None,               # object with training_step(...) / get_current_max_step_ratio()
    camera_intrinsics: Optional[CameraIntrinsics] = None,
    camera_bounds: Optional[CameraBounds] = None,
    hemispherical_radius: float = HEMISPHERICAL_RADIUS_CONSTANT,   # distance of the random SDS cameras; the reference
    # hard-codes 4.0311 (sds_trainer.py:45,270) whatever the scene -- pass another value only as an explicit override
    fused_grid_step: bool = True,       # FusedGridAdam (single process): the render gradient stays in the workspace
    saved_hemispherical_radius: Optional[float] = None,            # radius estimate written into the checkpoints when
    # there is no dataset to estimate it from (the reference stores train_dataset.get_hemispherical_radius_estimate())
) -> VolumetricModel:
    """Edit `sds_vol_mod` (a copy of `pretrained_vol_mod`) with score distillation.  Returns it."""
    for vm in (sds_vol_mod, pretrained_vol_mod):
        if not isinstance(vm.thre3d_repr, VoxelGrid) or vm.render_procedure != render_sh_voxel_grid:
            raise AssertionError("this train procedure needs SH-based VoxelGrid volumetric models")
    if uncoupled_mode or data_pose_mode:
        if train_dataset is None:
            raise ValueError("uncoupled_mode / data_pose_mode need a dataset of posed images")
    if train_dataset is not None:
        # like the reference (sds_trainer.py:152-156): a dataset's intrinsics are the ones the rays AND the target pixels
        # live at, so they win over a caller-supplied value (uncoupled_mode compares rendered and dataset pixels)
        camera_intrinsics = train_dataset.camera_intrinsics
        camera_bounds = camera_bounds or train_dataset.camera_bounds
        extra_radius = train_dataset.get_hemispherical_radius_estimate()
    else:
        extra_radius = hemispherical_radius if saved_hemispherical_radius is None else saved_hemispherical_radius
    if camera_intrinsics is None or camera_bounds is None:
        raise ValueError("camera_intrinsics and camera_bounds are required without a dataset")
    im_h, im_w = (int(v) for v in (image_dims if image_dims is not None else camera_intrinsics[:2]))
    device = sds_vol_mod.device

    output_dir = Path(output_dir)
    model_dir, render_dir = output_dir / "saved_models", output_dir / "training_logs" / "rendered_output"
    for d in (model_dir, render_dir):
        d.mkdir(exist_ok=True, parents=True)

    regular_density = pretrained_vol_mod.thre3d_repr.densities.detach().to(device)
    regular_features = pretrained_vol_mod.thre3d_repr.features.detach().to(device)

    if do_sds and guidance is None:
        from thre3d_atom.thre3d_reprs.sd import scoreDistillationLoss

        guidance = scoreDistillationLoss(device, sds_prompt, t_sched_start=sds_t_start, t_sched_freq=sds_t_freq,
                                         t_sched_gamma=sds_t_gamma, directional=True)   # the reference's prompts are always '<prompt>, <dir> view'

    grid = sds_vol_mod.thre3d_repr
    # ---- ray-sharded data parallelism (one process per GPU; no-op for a single process) ---------------------
    # Every rank holds a replica of the grid, renders ITS band of image rows, the bands are all-gathered so that
    # every rank runs the identical guidance step on the full image (same seeds => same timestep / noise), the
    # gradient w.r.t. its own band flows back through the HIP backward and the grid gradient is summed with one
    # RCCL all-reduce.  Whole-grid regularisers are identical on all ranks, hence scaled by 1 / world.
    rank, world = parallel.world_info()
    flat = parallel.FlatGrid(grid) if world > 1 else None
    if flat is not None:
        flat.broadcast_param(0)
        # identical random poses / diffusion timesteps / noise on every rank: share rank 0's seed
        seed = torch.tensor([int(torch.initial_seed() % (2 ** 31))], dtype=torch.int64, device=device)
        torch.distributed.broadcast(seed, src=0)
        torch.manual_seed(int(seed.item()))
        np.random.seed(int(seed.item()))
    row_lo, row_hi = parallel.shard_rows(im_h, rank, world)
    if fused_grid_step and flat is None:
        optimizer = FusedGridAdam(grid, lr=learning_rate, betas=(0.9, 0.999))
    else:   # (data-parallel runs exchange the flat .grad buffer: ordinary gradients)
        optimizer = VoxeAdam([{"params": grid.parameters(), "lr": learning_rate}], betas=(0.9, 0.999))
    lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=lr_gamma)
    # the default regulariser of the edit (density correlation with the pretrained field, sds_trainer.py:305-309 of the
    # reference) is evaluated INSIDE the fused grid step: no autograd node, no gradient tensor, no extra pass over the grid
    # (only with trainable densities: a features-only edit keeps the autograd term, whose gradient simply goes nowhere; the fused
    #  step runs with world == 1, where the regularisers' 1 / world scale is 1 -- passed anyway, so the two paths cannot diverge)
    # r06: with l2_mode / l1_mode (sds_trainer.py:494-503) and for the feature-correlation term (:526-534) as well
    dcl_in_step = (isinstance(optimizer, FusedGridAdam) and not uncoupled_mode
                   and density_correlation_weight != 0.0 and optimizer.trains_densities)
    if dcl_in_step:
        optimizer.set_density_correlation(regular_density, density_correlation_weight * (1.0 / world), l2_mode=l2_mode, l1_mode=l1_mode)
    featcorr_in_step = (isinstance(optimizer, FusedGridAdam) and feature_correlation_weight > 0.0 and optimizer.trains_features
                        and grid.features.shape[-1] <= 3)
    if featcorr_in_step:
        optimizer.set_feature_correlation(regular_features, feature_correlation_weight * (1.0 / world))
    extra_info = {CAMERA_BOUNDS: camera_bounds, CAMERA_INTRINSICS: camera_intrinsics, HEMISPHERICAL_RADIUS: extra_radius}

    log.info(f"SDS editing: grid {grid.grid_dims}, image [{im_h} x {im_w}], {num_iterations} iterations")
    intr = CameraIntrinsics(im_h, im_w, camera_intrinsics.focal * im_w / camera_intrinsics.width)
    trained_time, last = 0.0, time.perf_counter()
    rays_batch, direction_batch, pose, pixels_batch = None, None, None, None
    data_cursor = 0

    try:
        for global_step in range(1, num_iterations + 1):
            if global_step % new_frame_frequency == 0 or global_step == 1:
                if uncoupled_mode or data_pose_mode:
                    image, pose_mat, _ = train_dataset[data_cursor % len(train_dataset)]
                    data_cursor += 1
                    pose = CameraPose(rotation=pose_mat[:, :3], translation=pose_mat[:, 3:])
                    direction_batch = _get_dir_batch_from_poses(pose_mat[None])
                    pixels_batch = image.to(device).permute(1, 2, 0).reshape(-1, image.shape[0])
                else:
                    pose, direction, _, _ = get_random_pose(hemispherical_radius)
                    direction_batch = [direction]
                full_rays = cast_rays(intr, pose, device=device)
                if world > 1:  # this rank's band of rows (a smaller image-ordered ray batch)
                    from thre3d_atom.rendering.volumetric.render_interface import Rays

                    band = Rays(full_rays.origins[row_lo:row_hi], full_rays.directions[row_lo:row_hi],
                                image_shape=(row_hi - row_lo, im_w))
                    rays_batch = flatten_rays(band)
                else:
                    rays_batch = flatten_rays(full_rays)

            rendered = sds_vol_mod.render_rays(rays_batch)
            colour = rendered.colour
            if world > 1:
                colour = parallel.gather_image_rows(colour.reshape(row_hi - row_lo, im_w, -1), im_h).reshape(im_h * im_w, -1)
            reg_scale = 1.0 / world
            total_loss = 0
            if do_sds:
                total_loss = total_loss + guidance.training_step(colour, im_h, im_w, directions=direction_batch,
                                                                 global_step=global_step)
            if uncoupled_mode:
                fit = torch.nn.functional.mse_loss if uncoupled_l2_mode else torch.nn.functional.l1_loss
                total_loss = total_loss + fit(colour, pixels_batch) * density_correlation_weight
            elif not dcl_in_step:
                dcl, _ = density_correlation_loss_fn(grid.densities, regular_density, l2_mode=l2_mode, l1_mode=l1_mode)
                total_loss = total_loss + dcl * (density_correlation_weight * reg_scale)
            if feature_correlation_weight > 0.0 and not featcorr_in_step:
                total_loss = total_loss + _feature_correlation_loss(grid.features, regular_features) * (feature_correlation_weight * reg_scale)
            if tv_density_weight > 0:
                total_loss = total_loss + _tv_loss_on_grid(torch.relu(grid.densities)) * (tv_density_weight * reg_scale)
            if tv_features_weight > 0:
                total_loss = total_loss + _tv_loss_on_grid(grid.features) * (tv_features_weight * reg_scale)

            if flat is not None:
                flat.zero_grad()
            if torch.is_tensor(total_loss) and total_loss.requires_grad:   # (everything may live inside the grid step)
                total_loss.backward()
            if flat is not None:
                flat.all_reduce_grad()      # sum of the per-band render gradients (+ world x regulariser / world)
            optimizer.step()
            if flat is None:
                optimizer.zero_grad()
            trained_time += time.perf_counter() - last

            if global_step % summary_freq == 0 or global_step in (1, num_iterations):
                shown = float(total_loss.detach()) if torch.is_tensor(total_loss) else float(total_loss)
                if dcl_in_step:    # (its value comes out of the grid step)
                    shown += density_correlation_weight * float(optimizer.dcl_loss)
                if featcorr_in_step:
                    shown += feature_correlation_weight * float(optimizer.featcorr_loss)
                log.info(f"Iteration: {global_step}, total_loss: {shown: .3f}")
            if global_step % lr_freq == 0 and global_step >= lr_decay_start:
                lr_scheduler.step()
                log.info(f"Adjusted learning rate | learning rates: {[g['lr'] for g in optimizer.param_groups]}")
            if rank != 0:
                last = time.perf_counter()
                continue
            if global_step % feedback_freq == 0 or global_step in (1, num_iterations):
                log.info(f"TIME CHECK: time spent actually training till now: {timedelta(seconds=trained_time)}")
                _write_feedback(sds_vol_mod, render_feedback_pose or pose, intr, render_dir / f"sds_{global_step}.png")
            if global_step % save_freq == 0 or global_step in (1, num_iterations):
                torch.save(sds_vol_mod.get_save_info(extra_info=extra_info), model_dir / f"model_iter_{global_step}.pth")
            last = time.perf_counter()
    finally:
        if isinstance(optimizer, FusedGridAdam):   # (also when the loop raised: the mode must not outlive its optimiser)
            optimizer.detach()
    if rank == 0:
        torch.save(sds_vol_mod.get_save_info(extra_info=extra_info), model_dir / "model_final.pth")
    log.info("Training complete")
    return sds_vol_mod


def _write_feedback(vol_mod: VolumetricModel, pose: CameraPose, intrinsics: CameraIntrinsics, path: Path) -> None:
    """no-grad render at render_num_samples_per_ray, written as PNG when an image writer is available"""
    out = vol_mod.render(pose, intrinsics, num_samples_per_ray=vol_mod.render_config.render_num_samples_per_ray)
    try:
        from PIL import Image

        Image.fromarray(to8b(out.colour.cpu().numpy())).save(path)
    except ImportError:  # pragma: no cover
        np.save(path.with_suffix(".npy"), out.colour.cpu().numpy())
