"""Forward of view-dependent grids (SH 1-3) at 160^3 / 400x400: ray-ordered forward vs the LDS window of whole texels
(voxe_render_tilew.hip), per camera, with the share of the render the window served (VoxeDispatch::fwd_window = 2 marks the rest).
    gpurun -- python tools/sh_fwd_window.py [degrees, e.g. 123] [cameras, e.g. 3,12,53,77] [image side]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import torch  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, synth_pose_angles  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi, dispatch, ops  # noqa: E402


def main():
    degs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "123")]
    cams = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "3,12,53,77").split(",")]
    hw = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    G = 160
    dev = torch.device("cuda:0")
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS)
    gen = torch.Generator().manual_seed(42)
    dens = torch.empty((G, G, G, 1)).uniform_(-1.0, 1.0, generator=gen).to(dev)
    R = hw * hw
    outs = [torch.empty((R, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
    for deg in degs:
        F = 3 * (deg + 1) ** 2
        feat = torch.empty((G, G, G, F)).uniform_(-1.0, 1.0, generator=gen).to(dev)
        params = ops.RenderParams(num_samples=256, near=NEAR, far=FAR, perturb=True, white_bkgd=True, sh_degree=deg, image_width=hw)
        ws = ops.Workspace()
        for cam in cams:
            pose = pose_spherical(*synth_pose_angles(cam, 100), RADIUS)
            ro, rd = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
            line = f"SH-{deg} camera {cam:2d}:"
            for name, over in (("ray-ordered", dict(fwd_window=-1)), ("window", dict(fwd_window=0)),
                               ("window+z", dict(fwd_window=0, fwd_zdom=-1.0))):
              with dispatch.override(**over):
                for _ in range(2):
                    ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *outs, ws, (42, 1))
                torch.cuda.synchronize()
                ops.profile_enable(True)
                for i in range(5):
                    ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *outs, ws, (42, 2 + i))
                torch.cuda.synchronize()
                p = ops.profile_read()
                ops.profile_enable(False)
                line += f"  {name} {p['ms_fwd'] / max(p['n_fwd'], 1):.3f} ms"
              if name != "ray-ordered":
                with dispatch.override(**dict(over, fwd_window=2)):
                    ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *outs, ws, (42, 1))
                    torch.cuda.synchronize()
                    acc = outs[2][:, 0]
                    line += f" (served {100.0 * float((~torch.isnan(acc)).float().mean()):.0f} % of the pixels)"
            print(line, flush=True)
        del feat, ws


if __name__ == "__main__":
    main()
