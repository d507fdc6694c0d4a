#!/usr/bin/env python3
"""Render a camera path of a trained SH voxel grid (entry point kept from the reference's
render_sh_based_voxel_grid.py:74-170; same option names).  Every frame is one fused HIP forward launch.
Frames are written as PNGs, plus rendered_video.mp4 when `imageio` is installed (an animated PNG otherwise)."""
import os
import sys
from pathlib import Path

import click
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "vox-e_amd"))

from thre3d_atom.modules.volumetric_model import create_volumetric_model_from_saved_model  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import create_voxel_grid_from_saved_info_dict  # noqa: E402
from thre3d_atom.utils.constants import CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS  # noqa: E402
from thre3d_atom.utils.imaging_utils import (  # noqa: E402
    CameraPose,
    get_thre360_animation_poses,
    get_thre360_spiral_animation_poses,
    scale_camera_intrinsics,
    to8b,
)


@click.command()
@click.option("-i", "--model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the trained (reconstructed) model")
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
@click.option("--save_freq", type=click.INT, default=None, help="write every n-th frame as PNG (default: all)")
@click.option("-p", "--sds_prompt", type=click.STRING, required=False, default=None)
def main(**kwargs) -> None:
    cfg = type("Config", (), kwargs)
    device = torch.device("cuda")
    out = Path(cfg.output_path)
    out.mkdir(exist_ok=True, parents=True)
    if cfg.sds_prompt is not None:
        (out / "prompt.txt").write_text(cfg.sds_prompt)
    vol_mod, extra = create_volumetric_model_from_saved_model(Path(cfg.model_path), create_voxel_grid_from_saved_info_dict, device=device)
    vol_mod.render_config.white_bkgd = True  # the reference forces a white background at inference (:97-98)
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
                                  rgba_white_bkgd=True)
        poses = [CameraPose(p[:, :3], p[:, 3:]) for p in data.poses]
    intrinsics = scale_camera_intrinsics(intrinsics, cfg.render_scale_factor)
    frames = []
    for n, pose in enumerate(poses):
        rendered = vol_mod.render(pose, intrinsics, gpu_render=True, num_samples_per_ray=cfg.overridden_num_samples_per_ray)
        frames.append(to8b(rendered.colour.cpu().numpy()))
        if cfg.save_freq is None or n % cfg.save_freq == 0:
            from PIL import Image

            Image.fromarray(frames[-1]).save(out / f"frame_{n:04d}.png")
    try:
        import imageio

        imageio.mimwrite(out / "rendered_video.mp4", frames, fps=cfg.fps)
    except ImportError:
        # no video encoder available: an animated PNG of the same frames (plays in browsers) next to the stills
        from PIL import Image

        stills = [Image.fromarray(f) for f in frames]
        stills[0].save(out / "rendered_video.png", save_all=True, append_images=stills[1:], duration=int(1000 / cfg.fps), loop=0)
        print(f"imageio not installed: wrote {len(frames)} PNG frames and rendered_video.png (APNG) to {out} instead of an mp4")


if __name__ == "__main__":
    main()
