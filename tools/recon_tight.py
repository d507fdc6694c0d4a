"""The reconstruction iteration as bench.py's secondary.recon_iteration times it (tight loop over ops.recon_step_, fixed cameras, no torch
op between the calls), with / without voxe_recon_prefetch:   gpurun -- python tools/recon_tight.py [iters]     RECON_NO_PREFETCH=1: no hint"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), ROOT]
import torch  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi, ops  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
G, HW, S, K, B = 160, 400, 256, 8, 32768
dens, feat = random_grid(G)
d2, f2 = dens.to(dev), feat.to(dev)
spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY, density_post_act=abi.ACT_SOFTPLUS)
st_d, st_f = (torch.zeros_like(d2), torch.zeros_like(d2)), (torch.zeros_like(f2), torch.zeros_like(f2))
cams = [pose_spherical(*synth_pose_angles(3 + 11 * i, 100), RADIUS) for i in range(K)]
poses = torch.stack([torch.cat([p.rotation, p.translation], dim=-1) for p in cams]).to(dev).contiguous()
images = torch.rand((K, 3, HW, HW), generator=torch.Generator().manual_seed(45)).to(dev)
losses = torch.zeros(4, device=dev)
ws_a, ws_b = ops.Workspace(), ops.Workspace()
pr = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True)
hint = not os.environ.get("RECON_NO_PREFETCH")


def it(n):
    ops.recon_step_(spec, pr, d2, f2, ws_a, ws_b, HW, HW, focal_for(HW), poses, None, images, B, True, st_d, st_f, n, n, 1e-4, losses,
                    (77, 10 * n), zero_gradient_first=(n == 1))
    if hint:
        ops.recon_prefetch_(spec, pr, d2, f2, ws_a, ws_b, HW, HW, focal_for(HW), poses, None, images, B, True, losses, (77, 10 * (n + 1)))


gc.collect()
gc.disable()
for n in range(1, 11):
    it(n)
torch.cuda.synchronize()
t = time.perf_counter()
for n in range(11, 11 + iters):
    it(n)
torch.cuda.synchronize()
e = (time.perf_counter() - t) / iters
import ctypes
st = (ctypes.c_int64 * 3)()
ops.lib().voxe_recon_prefetch_stats(st)
print(f"recon tight loop ({'hint' if hint else 'no hint'}): {e * 1e3:.4f} ms per iteration; hints issued / taken / dropped {list(st)}")
