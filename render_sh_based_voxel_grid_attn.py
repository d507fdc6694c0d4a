#!/usr/bin/env python3
"""Render a camera path of a refined field together with its attention grid (entry point kept from the reference's
render_sh_based_voxel_grid_attn.py:84-209; same option names).

Per frame: one fused HIP forward of the colour field and one of the 1-channel attention grid
(`VolumetricModel.render_attn`); the attention render is normalised per frame and colour-mapped ("jet", like the
reference's visualisation).  The video holds the colour frames (the reference's `output_only`); the saved stills are
[colour | attention] side by side.  `--use_sd` (attention straight from Stable Diffusion instead of the stored grid) is
accepted for command-line compatibility and needs diffusers; without `--load_attention` this is the plain renderer."""
import os
import sys
from pathlib import Path

import click
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "vox-e_amd"))

from thre3d_atom.modules.volumetric_model import (  # noqa: E402
    create_volumetric_model_from_saved_model,
    create_volumetric_model_from_saved_model_attn,
)
from thre3d_atom.thre3d_reprs.voxels import (  # noqa: E402
    create_voxel_grid_from_saved_info_dict,
    create_voxel_grid_from_saved_info_dict_attn,
)
from thre3d_atom.utils.constants import CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS  # noqa: E402
from thre3d_atom.utils.imaging_utils import (  # noqa: E402
    CameraPose,
    get_thre360_animation_poses,
    get_thre360_spiral_animation_poses,
    scale_camera_intrinsics,
    to8b,
)
from thre3d_atom.utils.logging import log  # noqa: E402


def jet(x: np.ndarray) -> np.ndarray:
    """matplotlib's "jet" as piecewise-linear ramps: x in [0, 1] -> RGB in [0, 1] (no matplotlib dependency)"""
    x = np.clip(x, 0.0, 1.0)
    r = np.interp(x, [0.0, 0.35, 0.66, 0.89, 1.0], [0.0, 0.0, 1.0, 1.0, 0.5])
    g = np.interp(x, [0.0, 0.125, 0.375, 0.64, 0.91, 1.0], [0.0, 0.0, 1.0, 1.0, 0.0, 0.0])
    b = np.interp(x, [0.0, 0.11, 0.34, 0.65, 1.0], [0.5, 1.0, 1.0, 0.0, 0.0])
    return np.stack([r, g, b], axis=-1)


def colour_mapped(attn: np.ndarray) -> np.ndarray:
    lo, hi = float(attn.min()), float(attn.max())
    return jet((attn - lo) / (hi - lo) if hi > lo else np.zeros_like(attn))


@click.command()
@click.option("-i", "--model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the trained model")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path for saving rendered output")
@click.option("-r", "--ref_path", type=click.Path(file_okay=True, dir_okay=False), default=None, help="reference model whose camera info is used")
@click.option("-d", "--data_path", type=click.Path(file_okay=False, dir_okay=True), default=None, help="dataset (camera_path=dataset)")
@click.option("--overridden_num_samples_per_ray", type=click.IntRange(min=1), default=512, show_default=True)
@click.option("--render_scale_factor", type=click.FLOAT, default=2.0, show_default=True)
@click.option("--camera_path", type=click.Choice(["thre360", "spiral", "dataset"]), default="thre360", show_default=True)
@click.option("--camera_pitch", type=click.FLOAT, default=60.0, show_default=True)
@click.option("--num_frames", type=click.IntRange(min=1), default=180, show_default=True)
@click.option("--vertical_camera_height", type=click.FLOAT, default=3.0, show_default=True)
@click.option("--num_spiral_rounds", type=click.IntRange(min=1), default=2, show_default=True)
@click.option("--fps", type=click.IntRange(min=1), default=60, show_default=True)
@click.option("--timestamp", type=click.INT, default=0, show_default=True, help="diffusion timestamp (--use_sd)")
@click.option("--use_sd", type=click.BOOL, default=False, show_default=True)
@click.option("--load_attention", type=click.BOOL, default=True, show_default=True)
@click.option("--sds_prompt", type=click.STRING, required=False, default="")
@click.option("--index_to_attn", type=click.INT, required=False, default=11, show_default=True)
@click.option("--save_freq", type=click.INT, default=None, help="write every n-th frame as PNG (default: all)")
def main(**kwargs) -> None:
    cfg = type("Config", (), kwargs)
    if cfg.use_sd:
        raise click.UsageError("--use_sd re-computes the attention with Stable Diffusion per frame; this entry point "
                               "renders the attention grid stored in the model (--use_sd False)")
    device = torch.device("cuda")
    out = Path(cfg.output_path)
    out.mkdir(exist_ok=True, parents=True)
    if cfg.sds_prompt is not None:
        (out / "prompt.txt").write_text(cfg.sds_prompt)
    if cfg.load_attention:
        vol_mod, extra = create_volumetric_model_from_saved_model_attn(
            Path(cfg.model_path), create_voxel_grid_from_saved_info_dict_attn, device=device, load_attn=True)
    else:
        vol_mod, extra = create_volumetric_model_from_saved_model(Path(cfg.model_path), create_voxel_grid_from_saved_info_dict, device=device)
    if cfg.ref_path is not None:
        _, extra = create_volumetric_model_from_saved_model(Path(cfg.ref_path), create_voxel_grid_from_saved_info_dict, device=device)
    radius, intrinsics = extra[HEMISPHERICAL_RADIUS], extra[CAMERA_INTRINSICS]
    if cfg.camera_path == "thre360":
        poses = get_thre360_animation_poses(radius, cfg.camera_pitch, cfg.num_frames)
    elif cfg.camera_path == "spiral":
        poses = get_thre360_spiral_animation_poses((radius / 8.0, radius), cfg.vertical_camera_height, cfg.num_spiral_rounds, cfg.num_frames)
    else:
        from thre3d_atom.data.datasets import PosedImagesDataset

        data = PosedImagesDataset(Path(cfg.data_path) / "train", Path(cfg.data_path) / "train_camera_params.json",
                                  rgba_white_bkgd=vol_mod.render_config.white_bkgd)
        poses = [CameraPose(p[:, :3], p[:, 3:]) for p in data.poses]
    intrinsics = scale_camera_intrinsics(intrinsics, cfg.render_scale_factor)
    frames = []
    for n, pose in enumerate(poses):
        log.info(f"rendering frame number: ({n + 1}/{len(poses)})")
        rendered = vol_mod.render(pose, intrinsics, gpu_render=True, num_samples_per_ray=cfg.overridden_num_samples_per_ray)
        colour = to8b(rendered.colour.cpu().numpy())
        frames.append(colour)
        still = colour
        if cfg.load_attention:
            attn = vol_mod.render_attn(pose, intrinsics, gpu_render=True, num_samples_per_ray=cfg.overridden_num_samples_per_ray)
            still = np.concatenate([colour, to8b(colour_mapped(attn.attn.squeeze(-1).cpu().numpy()))], axis=1)
        if cfg.save_freq is None or n % cfg.save_freq == 0:
            from PIL import Image

            Image.fromarray(still).save(out / f"frame_{n:04d}.png")
    try:
        import imageio

        imageio.mimwrite(out / "rendered_video.mp4", frames, fps=cfg.fps)
    except ImportError:
        from PIL import Image

        stills = [Image.fromarray(f) for f in frames]
        stills[0].save(out / "rendered_video.png", save_all=True, append_images=stills[1:], duration=int(1000 / cfg.fps), loop=0)
        print(f"imageio not installed: wrote {len(frames)} PNG frames and rendered_video.png (APNG) to {out} instead of an mp4")


if __name__ == "__main__":
    main()
