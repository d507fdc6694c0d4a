#!/usr/bin/env python3
"""NeRF-synthetic ("blender") scene -> the camera-parameter files the datasets of this package read.

Same command line and output as the reference's tools/convert_from_nerf_blender_dataset.py:33-90: for every split
present (`transforms_{train,val,test}.json`) one `<split>_camera_params.json` keyed by image file name, with the 3x3
rotation / 3x1 translation of `transform_matrix`, the image size, `focal = 0.5 W / tan(0.5 camera_angle_x)` and the
scene bounds [2, 6] (:15,:58).  Differences: splits that do not exist are skipped instead of aborting the run, the
image size is read with PIL (imageio is not a dependency), and `--link_images` can symlink the image folders next to
the converted files so that the output directory is directly usable as `-d` of the training script."""
import json
import math
import os
import sys
from pathlib import Path

import click

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vox-e_amd"))

from thre3d_atom.data.constants import BOUNDS, EXTRINSIC, FOCAL, HEIGHT, INTRINSIC, ROTATION, TRANSLATION, WIDTH  # noqa: E402
from thre3d_atom.utils.logging import log  # noqa: E402

SPLITS = ("train", "val", "test")
NEAR, FAR = 2.0, 6.0


def convert_split(data_path: Path, split: str) -> dict:
    from PIL import Image

    meta = json.loads((data_path / f"transforms_{split}.json").read_text())
    frames = meta["frames"]
    if not frames:
        return {}
    first = data_path / split / (frames[0]["file_path"].split("/")[-1] + ".png")
    with Image.open(first) as img:
        width, height = img.size
    focal = 0.5 * width / math.tan(0.5 * float(meta["camera_angle_x"]))
    out = {}
    for frame in frames:
        matrix = frame["transform_matrix"]
        out[frame["file_path"].split("/")[-1] + ".png"] = {
            INTRINSIC: {BOUNDS: [NEAR, FAR], HEIGHT: height, WIDTH: width, FOCAL: focal},
            EXTRINSIC: {ROTATION: [[float(v) for v in row[:3]] for row in matrix[:3]],
                        TRANSLATION: [[float(row[3])] for row in matrix[:3]]},
        }
    return out


@click.command()
@click.option("-d", "--data_path", type=click.Path(file_okay=False, dir_okay=True), required=True,
              help="path to the original nerf synthetic dataset scene")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True,
              help="path for outputting the converted scene")
@click.option("--link_images", type=click.BOOL, default=False, show_default=True,
              help="also symlink <data_path>/<split> into the output directory")
def main(data_path: str, output_path: str, link_images: bool) -> None:
    src, dst = Path(data_path), Path(output_path)
    dst.mkdir(parents=True, exist_ok=True)
    done = []
    for split in SPLITS:
        if not (src / f"transforms_{split}.json").exists():
            log.info(f"no transforms_{split}.json under {src}: split skipped")
            continue
        params = convert_split(src, split)
        (dst / f"{split}_camera_params.json").write_text(json.dumps(params, ensure_ascii=False, indent=4), encoding="utf-8")
        if link_images and not (dst / split).exists():
            os.symlink((src / split).resolve(), dst / split, target_is_directory=True)
        done.append(f"{split} ({len(params)} views)")
    if not done:
        raise click.UsageError(f"no transforms_<split>.json found under {src}")
    log.info(f"converted {', '.join(done)} -> {dst}")


if __name__ == "__main__":
    main()
