"""How many of the 8 trilinear corners do CONSECUTIVE samples of a ray share at the bench workload (160^3, 400x400, S = 256,
the reference's always-on jitter)?  That is what a "pending footprint" (keep the current cell's corners in registers,
deposit only the corners the ray leaves) could save -- per LANE; a wave instruction is only skipped when all 64 rays
of the tile keep a corner in the same step.  CPU only (oracle probe):  python tools/footprint_overlap.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), ROOT]
import numpy as np  # noqa: E402

from oracle import voxe_oracle as vo  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi  # noqa: E402
from voxe_hip.desc import make_render_cfg  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles  # noqa: E402


def main():
    dens, feat = random_grid(160)
    grid = vo.Grid(dens.numpy(), feat.numpy(), [(-1.5, 1.5)] * 3, 100 / 3, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    for cam in (3, 40, 77):
        pose = pose_spherical(*synth_pose_angles(cam, 100), RADIUS)
        o, d = vo.cast_rays(400, 400, focal_for(400), pose.rotation.numpy(), pose.translation.numpy())
        # one 8x8 pixel tile in 25 (rows / columns 0..7 of every 40): whole tiles, so the wave-level statistic is exact
        yy, xx = np.meshgrid(np.arange(400), np.arange(400), indexing="ij")
        keep = ((yy % 40) < 8) & ((xx % 40) < 8)
        sel = np.nonzero(keep.reshape(-1))[0]
        cfg = make_render_cfg(256, NEAR, FAR, perturb=True, white_bkgd=True, seed=42, rng_offset=1)
        # (the in-kernel jitter stream is keyed by the ray's index in the launch: renumbering the subset draws other uniforms of
        #  the same distribution, which is all a statistic needs)
        pr = vo.sample_probe(grid, cfg, np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]))
        idx, inside = pr["idx"].astype(np.int64), pr["inside"]
        both = inside[:, 1:] & inside[:, :-1]
        delta = np.abs(idx[:, 1:] - idx[:, :-1])
        shared = np.prod(np.clip(2 - delta, 0, 2), axis=-1)            # corners of sample k that sample k+1 also touches
        sh = shared[both]
        hist = np.bincount(sh, minlength=9)[[0, 1, 2, 4, 8]] / sh.size
        # wave level: tile t = 64 rays (8x8 pixels), lock step over k: a corner's deposit instruction can only be skipped when
        # every ray of the tile that is inside at k and k+1 keeps its whole cell
        ty, tx = (yy.reshape(-1)[sel] // 40), (xx.reshape(-1)[sel] // 40)
        tile = ty * 10 + tx
        same_cell = (shared == 8) | ~both
        any_pair = both
        frac_uniform = []
        for t in np.unique(tile):
            m = tile == t
            has = any_pair[m].any(axis=0)
            frac_uniform.append((same_cell[m].all(axis=0) & has).sum() / max(has.sum(), 1))
        print(f"camera {cam}: consecutive in-AABB samples share {sh.mean():.2f} of 8 corners on average "
              f"(0 / 1 / 2 / 4 / 8 shared: {' / '.join(f'{h:.2f}' for h in hist)}); "
              f"steps in which ALL rays of an 8x8 tile keep their cell: {100 * np.mean(frac_uniform):.2f} %")


if __name__ == "__main__":
    main()
