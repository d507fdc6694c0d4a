import os, sys
ROOT="/root/repo"
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import gpu_helpers as gh
from helpers import band_errors, rel_l2
from voxe_hip.workload import *
from voxe_hip import abi
from voxe_hip.dispatch import Dispatch
from voxe_hip.desc import make_render_cfg
from oracle import voxe_oracle as vo
from thre3d_atom.utils.imaging_utils import pose_spherical
dens, feat = random_grid(160)
grid = vo.Grid(dens.numpy(), feat.numpy(), [(-1.5, 1.5)] * 3, 100 / 3, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
yaw, pitch = synth_pose_angles(3, 100); pose = pose_spherical(yaw, pitch, RADIUS)
o, d = vo.cast_rays(400, 400, focal_for(400), pose.rotation.numpy(), pose.translation.numpy())
cfg = make_render_cfg(256, NEAR, FAR, perturb=True, white_bkgd=True, seed=42, rng_offset=7)
gc = np.random.default_rng(43).standard_normal((o.shape[0], 3)).astype(np.float32)
perm = np.random.default_rng(11).permutation(o.shape[0])
inv = np.argsort(perm)
rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
res = {}
res["tile"] = gh.hip_backward(grid, cfg, o, d, gc, rng=(42, 7), image_width=400)
# VoxeDispatch::precise_grad (r05): the in-segment suffix sums from the forward's segment-local sums in double
res["tile_precise"] = gh.hip_backward(grid, cfg, o, d, gc, rng=(42, 7), image_width=400, dispatch=Dispatch(precise_grad=1))
res["tile_general_kernel"] = gh.hip_backward(grid, cfg, o, d, gc, rng=(42, 7), image_width=400, dispatch=Dispatch(tile_lean=-1))
# NOTE: the in-kernel jitter is keyed by ray index: a permuted batch draws other jitter -> compare permuted runs with an oracle run on the permuted rays
op, dp, gp = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm]), np.ascontiguousarray(gc[perm])
rdp, rfp = vo.render_bwd(grid, cfg, op, dp, gp)
res["region"] = gh.hip_backward(grid, cfg, op, dp, gp, rng=(42, 7), dispatch=Dispatch(region_min_rays=16384))
res["scatter"] = gh.hip_backward(grid, cfg, op, dp, gp, rng=(42, 7), dispatch=Dispatch(region_min_rays=-1))
bands = [(1e-3, 1.0), (1e-6, 1e-3), (1e-9, 1e-6)]
for k, (gd, gf) in res.items():
    r_d, r_f = (rd, rf) if k.startswith("tile") else (rdp, rfp)
    print(k, "rel_l2", rel_l2(gd, r_d), rel_l2(gf, r_f))
    for nm, got, ref in (("dens", gd, r_d), ("feat", gf, r_f)):
        print("  ", nm, {b: tuple(f"{x:.2e}" if isinstance(x, float) else x for x in v) for b, v in band_errors(got, ref, bands).items()})
print("region vs scatter")
for nm, a, b in (("dens", res["region"][0], res["scatter"][0]), ("feat", res["region"][1], res["scatter"][1])):
    print("  ", nm, {bb: tuple(f"{x:.2e}" if isinstance(x, float) else x for x in v) for bb, v in band_errors(a, b, bands).items()})
# tile vs scatter on the SAME rays in image order (scatter without the width hint)
sc = gh.hip_backward(grid, cfg, o, d, gc, rng=(42, 7), dispatch=Dispatch(region_min_rays=-1))
print("tile vs scatter(image order, no hint)")
for nm, a, b in (("dens", res["tile"][0], sc[0]), ("feat", res["tile"][1], sc[1])):
    print("  ", nm, {bb: tuple(f"{x:.2e}" if isinstance(x, float) else x for x in v) for bb, v in band_errors(a, b, bands).items()})
