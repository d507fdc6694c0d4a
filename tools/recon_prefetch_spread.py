"""Run-to-run spread of the reconstruction iterations' losses with and without hints (float-atomic gradient sums + Adam's first steps
make the iterations non-deterministic at the 1e-4 level on small grids): is a hinted run inside the spread of the plain ones?
    gpurun -- python tools/recon_prefetch_spread.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch  # noqa: E402
import test_recon_step_gpu as T  # noqa: E402
from voxe_hip import ops  # noqa: E402
from synth import focal_for  # noqa: E402


def run(with_hints, steps=9, side=40, hw=64, K=6, batch=18000, lr=2e-2):
    dens0, feat0, poses_all, images, spec, params = T._setup(side, hw, 10)
    gen = torch.Generator().manual_seed(7)
    d_b, f_b = dens0.clone(), feat0.clone()
    st_d = (torch.zeros_like(d_b), torch.zeros_like(d_b))
    st_f = (torch.zeros_like(f_b), torch.zeros_like(f_b))
    wa, wb = ops.Workspace(), ops.Workspace()
    losses = torch.zeros(4, device=T.DEV)
    cams = [torch.randint(0, 10, (K,), generator=gen).to(T.DEV) for _ in range(steps + 1)]
    cam_poses = [poses_all[c].contiguous() for c in cams]
    torch.cuda.synchronize()
    out = []
    for it in range(steps):
        ops.recon_step_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), cam_poses[it], cams[it], images, batch, True, st_d, st_f,
                        it + 1, it + 1, lr, losses, (3, 100 * it), zero_gradient_first=(it == 0))
        out.append(losses[0].item())
        if with_hints and it + 1 < steps:
            ops.recon_prefetch_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), cam_poses[it + 1], cams[it + 1], images, batch, True,
                                losses, (3, 100 * (it + 1)))
    torch.cuda.synchronize()
    return out


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
plain = [run(False) for _ in range(reps)]
hinted = [run(True) for _ in range(reps)]
for name, runs in (("plain", plain), ("hinted", hinted)):
    print(name)
    for r in runs:
        print("   ", " ".join(f"{x:.7f}" for x in r))
for it in range(len(plain[0])):
    p = [r[it] for r in plain]
    h = [r[it] for r in hinted]
    print(f"iteration {it}: plain spread {max(p) - min(p):.2e}  hinted spread {max(h) - min(h):.2e}  |mean difference| {abs(sum(p) / len(p) - sum(h) / len(h)):.2e}")
