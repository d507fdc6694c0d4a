"""debug aid: the lean tile backward vs the general one vs the oracle on one small case (GPU)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vox-e_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from conftest import load_golden
from helpers import cfg_from_bounds, grid_from_golden, rel_l2
import gpu_helpers as gh
from oracle import voxe_oracle as vo
from voxe_hip import dispatch as dp

h, w, cam = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = int(sys.argv[4]) if len(sys.argv) > 4 else 96
g = load_golden("frames32.npz")
grid = grid_from_golden(g, "", "softplus")
o, d = vo.cast_rays(h, w, 0.5 * w / np.tan(0.5 * 0.6911112), g["rot"][cam], g["trans"][cam])
cfg = cfg_from_bounds(g["bounds"], S, white_bkgd=True)
rng = np.random.default_rng(8 + cam)
gc = rng.standard_normal((h * w, 3)).astype(np.float32)
gdep = rng.standard_normal(h * w).astype(np.float32) * 0.1
rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep)
for lean in (0, -1):
    with dp.override(tile_min_rays=-1, tile_kl=8, tile_lean=lean):
        gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, image_width=w)
    print("lean" if lean == 0 else "general", "rel_l2 dens", rel_l2(gd, rd), "feat", rel_l2(gf, rf))
    e = np.abs(gd - rd)[..., 0]
    idx = np.unravel_index(np.argsort(e.ravel())[-5:], e.shape)
    print("   worst density voxels", list(zip(*[i.tolist() for i in idx])), e[idx], rd[..., 0][idx])
    ef = np.abs(gf - rf).sum(-1)
    print("   planes with feature error > 1e-4 x max: x", np.unique(np.nonzero(ef > 1e-4 * np.abs(rf).max())[0])[:40])
    print("                                          y", np.unique(np.nonzero(ef > 1e-4 * np.abs(rf).max())[1])[:40])
    print("                                          z", np.unique(np.nonzero(ef > 1e-4 * np.abs(rf).max())[2])[:40])
    print("   sum grad", gf.sum(), rf.sum(), gd.sum(), rd.sum())
