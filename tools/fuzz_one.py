"""One seed of tests/test_hip_fuzz.py::test_random_configuration with the numbers printed.   gpurun -- python tools/fuzz_one.py SEED"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np  # noqa: E402
import gpu_helpers as gh  # noqa: E402
import test_hip_fuzz as tf  # noqa: E402
from oracle import voxe_oracle as vo  # noqa: E402
from voxe_hip.dispatch import Dispatch  # noqa: E402

seed = int(sys.argv[1])
grid, cfg, o, d, jitter, (h, w), rng = tf._case(seed)
ordered = seed % 3 != 2
width = w if ordered else 0
if not ordered:
    perm = rng.permutation(h * w)
    o, d = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
    jitter = None if jitter is None else np.ascontiguousarray(jitter[perm])
print("dims", grid.densities.shape, "hw", h, w, "S", cfg.num_samples, "near/far", cfg.near, cfg.far, "clip", cfg.aabb_clip, "lindisp",
      cfg.linear_disparity, "ordered", ordered, "post", grid.density_post_act, "scale", grid.density_scale)
cout = grid.cout
gc = rng.standard_normal((h * w, cout)).astype(np.float32)
gdep = (0.2 * rng.standard_normal(h * w)).astype(np.float32)
gacc = (0.2 * rng.standard_normal(h * w)).astype(np.float32)
rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, d_acc=gacc, jitter=jitter)
for name, disp in (("shipped", None), ("scatter only", Dispatch(bwd_mode=1)), ("region off", Dispatch(region_min_rays=-1))):
    kw = {} if disp is None else {"dispatch": disp}
    try:
        gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, g_acc=gacc, jitter=jitter, image_width=width, **kw)
    except Exception as e:  # noqa: BLE001
        print(name, "failed:", e)
        continue
    for nm, got, ref in (("densities", gd, rd), ("features", gf, rf)):
        err = float(np.linalg.norm(got.astype(np.float64) - ref.astype(np.float64)))
        print(f"{name:14s} {nm:10s} err {err:.3e}  bound {1e-4 * float(np.linalg.norm(ref)) + 5e-5:.3e}  |ref| {float(np.linalg.norm(ref)):.3e}  max|diff| {float(np.abs(got - ref).max()):.3e}")
