"""Coarse-to-fine reconstruction of an SH voxel grid from posed images.

Entry point of the reference's thre3d_atom/modules/trainers.py (`train_sh_vox_grid_vol_mod_with_posed_images`
:55-506): `num_stages` stages, the grid doubling each stage (trilinear upsampling, HIP kernel), images
down-sampled by the matching factor, per iteration a random batch of rays over `image_batch_cache_size`
images, L1 loss on the specular render plus (optionally) on the diffuse render, Adam with a per-stage decay.
Rendering and its backward are the fused HIP kernels; the optimiser is the fused HIP Adam.
"""
import time
from functools import partial
from pathlib import Path
from typing import Any, Callable, Optional

import torch
from torch import Tensor

from thre3d_atom.modules.optim import FusedGridAdam, VoxeAdam
from thre3d_atom.modules.testers import test_sh_vox_grid_vol_mod_with_posed_images
from thre3d_atom.modules.volumetric_model import VolumetricModel
from thre3d_atom.rendering.volumetric.utils.misc import sample_random_rays_and_pixels_from_cameras
from thre3d_atom.thre3d_reprs.renderers import _render_params, render_sh_voxel_grid
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, scale_voxel_grid_with_required_output_size
from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS
from thre3d_atom.utils.imaging_utils import CameraPose, to8b
from thre3d_atom.utils.logging import log
from thre3d_atom.utils.metric_utils import mse2psnr
from thre3d_atom.utils.misc import compute_thre3d_grid_sizes
from voxe_hip.ops import _next_rng


class _PinnedStaging:
    """Host -> device copies of the per-iteration camera picks that do not stall the host: a pageable tensor's .to(device) waits for
    the stream (every kernel of the previous iterations) before it returns, which makes the loop host-paced; a small ring of pinned
    buffers + non_blocking copies does not.  A buffer is reused only once its copy has completed (event per buffer)."""

    def __init__(self, n: int, device, depth: int = 4):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.bufs = [torch.empty(n, dtype=torch.int64, pin_memory=self.cuda) for _ in range(depth)]
        self.events = [None] * depth
        self.turn = 0

    def to_device(self, fill) -> Tensor:
        i = self.turn % len(self.bufs)
        self.turn += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        fill(self.bufs[i])
        if not self.cuda:
            return self.bufs[i].clone()
        out = self.bufs[i].to(self.device, non_blocking=True)
        self.events[i] = torch.cuda.Event()
        self.events[i].record()
        return out


def train_sh_vox_grid_vol_mod_with_posed_images(
    vol_mod: VolumetricModel,
    train_dataset: Any,
    output_dir: Path,
    random_initializer: Callable[[Tensor], Tensor] = partial(torch.nn.init.uniform_, a=-1.0, b=1.0),
    test_dataset: Optional[Any] = None,
    image_batch_cache_size: int = 8,
    ray_batch_size: int = 32768,
    num_stages: int = 4,
    num_iterations_per_stage: int = 2000,
    scale_factor: float = 2.0,
    learning_rate: float = 0.03,
    lr_decay_gamma_per_stage: float = 0.1,
    lr_decay_steps_per_stage: int = 1000,
    stagewise_lr_decay_gamma: float = 0.9,
    render_feedback_pose: Optional[CameraPose] = None,
    save_freq: int = 1000,
    test_freq: int = 1000,
    feedback_freq: int = 100,
    summary_freq: int = 10,
    apply_diffuse_render_regularization: bool = True,
    num_workers: int = 4,
    verbose_rendering: bool = True,
    fast_debug_mode: bool = False,
    lpips_weight: float = 0.0,
    fused_grid_step: bool = True,      # addition of this build: FusedGridAdam (gradient stays in the kernels' workspace)
    fused_iteration: bool = True,      # addition of this build: the whole iteration (batch, 2 renders, L1, backward, Adam) as
                                       # ONE library call (voxe_recon_step) -- needs fused_grid_step; same arithmetic
    prefetch_batches: bool = True,     # addition of this build (fused_iteration only): the next iteration's pixel batch and its
                                       # segment tables are assembled while this iteration's backward / Adam run
                                       # (voxe_recon_prefetch); same batches, same arithmetic
) -> VolumetricModel:
    if not isinstance(vol_mod.thre3d_repr, VoxelGrid) or vol_mod.render_procedure != render_sh_voxel_grid:
        raise AssertionError("this train procedure needs an SH-based VoxelGrid volumetric model")
    if lpips_weight != 0.0:
        raise NotImplementedError("LPIPS needs the external `lpips` network; not part of this build")
    device = vol_mod.device
    output_dir = Path(output_dir)
    model_dir = output_dir / "saved_models"
    model_dir.mkdir(exist_ok=True, parents=True)

    grid_sizes = compute_thre3d_grid_sizes(vol_mod.thre3d_repr.grid_dims, num_stages, scale_factor)
    stage_datasets = [train_dataset.downsampled(scale_factor ** (num_stages - 1 - s)).to(device) for s in range(num_stages)]

    with torch.no_grad():
        vol_mod.thre3d_repr = scale_voxel_grid_with_required_output_size(vol_mod.thre3d_repr, grid_sizes[0])
        random_initializer(vol_mod.thre3d_repr.densities)
        random_initializer(vol_mod.thre3d_repr.features)

    extra_info = {CAMERA_BOUNDS: train_dataset.camera_bounds, CAMERA_INTRINSICS: train_dataset.camera_intrinsics,
                  HEMISPHERICAL_RADIUS: train_dataset.get_hemispherical_radius_estimate()}
    # rendered feedback (trainers.py:164-175,409-432 of the reference): the given pose, else the first held-out / training
    # view; stills go to training_logs/rendered_output as [specular | diffuse] PNGs
    render_dir = output_dir / "training_logs" / "rendered_output"
    if not fast_debug_mode:
        render_dir.mkdir(exist_ok=True, parents=True)
    if render_feedback_pose is None:
        first = (test_dataset if test_dataset is not None else train_dataset).poses[0].to(device)
        render_feedback_pose = CameraPose(rotation=first[:, :3], translation=first[:, 3:])

    def rendered_feedback(step: int) -> None:
        from PIL import Image

        intr_full = train_dataset.camera_intrinsics
        views = [vol_mod.render(render_feedback_pose, intr_full, gpu_render=True, verbose=verbose_rendering,
                                render_diffuse=diffuse).colour for diffuse in (False, True)]
        Image.fromarray(to8b(torch.cat(views, dim=1).cpu().numpy())).save(render_dir / f"default_{step}.png")

    global_step, trained = 0, 0.0
    gen = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
    for stage in range(1, num_stages + 1):
        data = stage_datasets[stage - 1]
        intr = data.camera_intrinsics
        lr = learning_rate * (stagewise_lr_decay_gamma ** (stage - 1))
        if fused_grid_step:
            optimizer = FusedGridAdam(vol_mod.thre3d_repr, lr=lr, betas=(0.9, 0.999))
        else:
            optimizer = VoxeAdam([{"params": vol_mod.thre3d_repr.parameters(), "lr": lr}], betas=(0.9, 0.999))
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=lr_decay_gamma_per_stage)
        # (the one-call iteration reads RGB targets straight from the image stack)
        one_call = fused_grid_step and fused_iteration and data.images.shape[1] == 3 and data.images.is_contiguous()
        fused_losses = torch.zeros(4, dtype=torch.float32, device=device)
        log.info(f"stage {stage}: grid {vol_mod.thre3d_repr.grid_dims}, images [{intr.height} x {intr.width}], lr {lr:.4f}")

        staging = _PinnedStaging(min(image_batch_cache_size, len(data)), device)

        def draw_batch():
            # a cache of `image_batch_cache_size` random views; the batch is a random subset of ALL their pixels
            # (cast + collate + randperm of the reference, restricted to the pixels that are kept)
            picks = staging.to_device(lambda out: torch.randint(0, len(data), (out.numel(),), generator=gen, out=out))
            if not one_call:
                return picks, None, None
            return picks, data.poses[picks].contiguous(), _next_rng()

        try:
            # (the one-call loop draws an iteration's cameras / streams one iteration AHEAD -- before the previous step is
            #  enqueued -- whether or not the hint below is given: the batches do not depend on `prefetch_batches`)
            upcoming = draw_batch()
            for it in range(1, num_iterations_per_stage + 1):
                t0 = time.perf_counter()
                picks, poses_it, rng_it = upcoming
                upcoming = draw_batch() if it < num_iterations_per_stage else None
                if one_call:
                    # ONE library call: random pixel batch over the picked cameras -> specular (+ diffuse) render -> L1 ->
                    # backward -> Adam with this optimiser's state and learning rate (voxe_recon_step).  The loss values stay
                    # on the device until they are logged.
                    params_it = _render_params(vol_mod.thre3d_repr, None, vol_mod.render_config, attn=False)
                    batch_it = min(ray_batch_size, picks.numel() * intr.height * intr.width)
                    optimizer.reconstruction_step(params_it, intr.height, intr.width, float(intr.focal), poses_it, picks, data.images,
                                                  batch_it, apply_diffuse_render_regularization, fused_losses, rng_it)
                    if prefetch_batches and upcoming is not None:
                        optimizer.reconstruction_prefetch(params_it, intr.height, intr.width, float(intr.focal), upcoming[1], upcoming[0],
                                                          data.images, batch_it, apply_diffuse_render_regularization, fused_losses,
                                                          upcoming[2])
                else:
                    rays_batch, pixels_batch = sample_random_rays_and_pixels_from_cameras(
                        intr, data.poses[picks], data.images, ray_batch_size, image_ids=picks,
                        # (sorting the batch by (camera, row, column) helps the ray-ordered gather of small batches; batches of
                        #  16384+ rays take the space-binned render, which does not care about the order: skip the sort)
                        memory_order=ray_batch_size < 16384, fast_subset=True)
                    specular = vol_mod.render_rays(rays_batch).colour
                    loss = torch.nn.functional.l1_loss(specular, pixels_batch)
                    psnr = mse2psnr(torch.nn.functional.mse_loss(specular.detach(), pixels_batch))
                    if apply_diffuse_render_regularization:
                        diffuse = vol_mod.render_rays(rays_batch, render_diffuse=True).colour
                        loss = loss + torch.nn.functional.l1_loss(diffuse, pixels_batch)
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
                global_step += 1
                trained += time.perf_counter() - t0
                if global_step % summary_freq == 0 or it in (1, num_iterations_per_stage):
                    if one_call:
                        l1_spec, mse_spec, l1_diff, _ = fused_losses.tolist()
                        loss = torch.tensor(l1_spec + (l1_diff if apply_diffuse_render_regularization else 0.0))
                        psnr = mse2psnr(torch.tensor(mse_spec))
                    log.info(f"Stage: {stage} Global Iteration: {global_step} Stage Iteration: {it} "
                             f"loss: {float(loss.detach()): .3f} psnr: {float(psnr): .3f}")
                if it % lr_decay_steps_per_stage == 0:
                    scheduler.step()
                last = it == num_iterations_per_stage
                if not fast_debug_mode and (global_step % feedback_freq == 0 or it == 1 or last):
                    log.info(f"TIME CHECK: time spent actually training till now: {trained:.1f} s")
                    rendered_feedback(global_step)
                if test_dataset is not None and not fast_debug_mode and (global_step % test_freq == 0 or last):
                    test_sh_vox_grid_vol_mod_with_posed_images(vol_mod, test_dataset, parallel_rays_chunk_size=ray_batch_size,
                                                               global_step=global_step)
                if it % save_freq == 0 and not fast_debug_mode:
                    torch.save(vol_mod.get_save_info(extra_info), model_dir / f"model_stage_{stage}_iter_{it}.pth")
        finally:
            if fused_grid_step:   # leave the deferred-gradient mode even when the loop raised (renders would return no .grad)
                optimizer.detach()
        if stage != num_stages:
            with torch.no_grad():
                vol_mod.thre3d_repr = scale_voxel_grid_with_required_output_size(vol_mod.thre3d_repr, grid_sizes[stage])
    torch.save(vol_mod.get_save_info(extra_info), model_dir / "model_final.pth")
    log.info(f"Training complete; time spent actually training: {trained:.1f} s")
    return vol_mod
