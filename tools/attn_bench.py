"""Attention-grid render (1 channel) forward + backward time at the refinement loop's shape: 160^3, one image, S=256.
    gpurun -- python tools/attn_bench.py [hw]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import torch  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, sphere_grid, synth_pose_angles  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi, ops  # noqa: E402


def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 else 266
    dev = torch.device("cuda:0")
    G = 160
    dens, _ = sphere_grid(G)
    dens = dens.to(dev)
    attn = torch.full((G, G, G, 1), -2.0, device=dev).requires_grad_(True)
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS, feature_kind=abi.FEAT_ATTN)
    pose = pose_spherical(*synth_pose_angles(3, 100), RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
    params = ops.RenderParams(num_samples=256, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw)
    ws = ops.Workspace()
    g = torch.randn(hw * hw, 1, device=dev)

    def step():
        a, _, _, _ = ops.render(spec, params, dens, attn, ro, rd, workspace=ws)
        (a * g).sum().backward()
        attn.grad = None
        with torch.no_grad():
            attn.add_(1e-6)   # parameters change every iteration: re-pack like a training step

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ops.profile_enable(True)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    p = ops.profile_read()
    print(f"attention render {hw}x{hw}, 160^3, S=256: {dt * 1e3:.3f} ms per fwd+bwd ({hw * hw / dt / 1e6:.1f} M rays/s); "
          f"kernels: pack {p['ms_pack'] / max(p['n_pack'], 1):.3f} fwd {p['ms_fwd'] / max(p['n_fwd'], 1):.3f} "
          f"bwd {p['ms_bwd'] / max(p['n_bwd'], 1):.3f} unpack {p['ms_unpack'] / max(p['n_unpack'], 1):.3f} ms")


if __name__ == "__main__":
    main()
