"""View-dependent grids (SH degree 1-3: 12 / 27 / 48 feature channels) -- render forward + backward time.
    gpurun -- python tools/sh_bench.py [grid side] [image side] [degrees, e.g. 0123]
The reference's unit test model is 27 features (SH-2, tests/test_volumetric_model.py); its scenes train SH-0."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import torch  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, synth_pose_angles  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi, ops  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    hw = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    degs = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "0123")]
    order = sys.argv[4] if len(sys.argv) > 4 else "image"      # "random": the same rays in a random permutation
    dev = torch.device("cuda:0")
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS)
    pose = pose_spherical(*synth_pose_angles(3, 100), RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
    R = hw * hw
    if order == "random":
        perm = torch.randperm(R, generator=torch.Generator().manual_seed(7)).to(dev)
        ro, rd = ro[perm].contiguous(), rd[perm].contiguous()
    gen = torch.Generator().manual_seed(42)
    dens = torch.empty((G, G, G, 1)).uniform_(-1.0, 1.0, generator=gen).to(dev)
    g_colour = torch.randn((R, 3), generator=torch.Generator().manual_seed(43)).to(dev)
    outs = [torch.empty((R, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
    for deg in degs:
        F = 3 * (deg + 1) ** 2
        feat = torch.empty((G, G, G, F)).uniform_(-1.0, 1.0, generator=gen).to(dev)
        d_dens, d_feat = torch.zeros_like(dens), torch.zeros_like(feat)
        params = ops.RenderParams(num_samples=256, near=NEAR, far=FAR, perturb=True, white_bkgd=True, sh_degree=deg,
                                  image_width=hw if order == "image" else 0)
        ws = ops.Workspace()
        n_it = [0]

        def step():
            n_it[0] += 1
            rng = (42, n_it[0])
            ws.invalidate()          # a training step changes the grid: re-pack it
            ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *outs, ws, rng)
            ops.render_bwd_into(spec, params, dens, feat, ro, rd, None, outs[0], outs[1], outs[2], g_colour, None, None,
                                d_dens, d_feat, ws, rng)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.profile_enable(True)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        p = ops.profile_read()
        ops.profile_enable(False)
        print(f"SH-{deg} ({F + 1} channels, grid {G}^3 = {G ** 3 * (F + 1) * 4 / 1e6:.0f} MB), {hw}x{hw} ({order} order), S=256: {dt * 1e3:.2f} ms per "
              f"fwd+bwd ({R / dt / 1e6:.2f} M rays/s); kernels: fwd {p['ms_fwd'] / max(p['n_fwd'], 1):.3f} "
              f"pack {p['ms_pack'] / max(p['n_pack'], 1):.3f} bwd {p['ms_bwd'] / max(p['n_bwd'], 1):.3f} memset {p['ms_memset'] / max(p['n_memset'], 1):.3f} "
              f"unpack {p['ms_unpack'] / max(p['n_unpack'], 1):.3f} ms", flush=True)
        del feat, d_feat, ws


if __name__ == "__main__":
    main()
