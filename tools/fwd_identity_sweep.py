"""Bit-identity of the LDS-window forwards (SH-0: 16-byte texels; SH 1-3: whole texels, r06) against the ray-ordered forward over
random cameras / image sizes / grids / sample counts / SH degrees (the default dispatch and the window forced onto every tile, z-march
included).   python tools/fwd_identity_sweep.py [n] [degrees, e.g. 0123]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vox-e_amd"))
from voxe_hip import ops  # noqa: E402
from voxe_hip.dispatch import Dispatch  # noqa: E402
from voxe_hip.workload import synth_pose_angles, RADIUS, NEAR, FAR, focal_for, random_grid, sphere_grid  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
degs = [int(ch) for ch in (sys.argv[2] if len(sys.argv) > 2 else "0")]
rng = np.random.default_rng(123)
dev = torch.device("cuda", 0)
MODES = {"off": Dispatch(fwd_window=-1), "default": Dispatch(),
         "forced": Dispatch(fwd_zdom=-1.0, fwd_max_adv=9.0, fwd_fit_lat=7.0, fwd_fit_m=7.0)}
bad = 0
for it in range(n):
    deg = int(rng.choice(degs))
    side = int(rng.choice([48, 64, 96, 128, 160] if deg < 2 else [48, 64, 96, 128]))
    hw = int(rng.choice([64, 100, 133, 200, 266, 320, 400]))
    S = int(rng.choice([64, 128, 192, 256]))
    cam = int(rng.integers(0, 100))
    dens, feat = (random_grid if rng.random() < 0.7 else sphere_grid)(side)
    if deg > 0:   # 3 (deg + 1)^2 coefficient channels
        feat = torch.empty(feat.shape[:3] + (3 * (deg + 1) ** 2,)).uniform_(-1.0, 1.0, generator=torch.Generator().manual_seed(1000 + it))
    dens, feat = dens.to(dev), feat.to(dev)
    spec = ops.GridSpec(aabb=[(-1.5, 1.5)] * 3, density_scale=100.0 / 3.0)
    pose = pose_spherical(*synth_pose_angles(cam, 100), RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
    prm = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=bool(rng.random() < 0.7), white_bkgd=True, image_width=hw, sh_degree=deg)
    outs = {}
    for mode in ("off", "default", "forced"):
        prm.dispatch = MODES[mode]
        with torch.no_grad():
            c, d, a, _ = ops.render(spec, prm, dens, feat, ro, rd, None, rng=(7, it))
        torch.cuda.synchronize()
        outs[mode] = (c.clone(), d.clone(), a.clone())
    ok = all(torch.equal(x, y) for m in ("default", "forced") for x, y in zip(outs["off"], outs[m]))
    bad += 0 if ok else 1
    print(f"{it:3d} SH-{deg} grid {side:3d} image {hw:3d} S {S:3d} cam {cam:2d} perturb {prm.perturb}: {'identical' if ok else 'DIFFERENT'}")
print(f"{n - bad} / {n} identical")
sys.exit(1 if bad else 0)
