"""Worker of tests/test_bench_two_ranks_gpu.py::test_two_ranks_equal_one_process (launched by torch.distributed.run with
2 ranks sharing the one visible GPU, gloo): K optimiser steps of a data-parallel job -- every rank renders ITS camera,
ShardedGridAdam exchanges and steps -- against the same K steps in ONE process that accumulates both cameras' gradients
in its workspace before each fused step.  Rank 0 prints one JSON line with the differences."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from synth import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles  # noqa: E402
from thre3d_atom.modules.parallel import ShardedGridAdam  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi, ops  # noqa: E402


def main():
    exchange, G, HW, S, K = sys.argv[1], 32, 64, 48, 4
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS)
    params = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=HW)
    dens0, feat0 = random_grid(G)
    cams = []
    for r in range(world):
        pose = pose_spherical(*synth_pose_angles(3 + 7 * r, 100), RADIUS)
        ro, rd = ops.cast_rays(HW, HW, focal_for(HW), pose.rotation, pose.translation, dev)
        g = torch.randn((HW * HW, 3), generator=torch.Generator().manual_seed(50 + r)).to(dev)
        cams.append((ro, rd, g))

    def run(my_cams, opt_kwargs):
        dens, feat = dens0.to(dev).clone(), feat0.to(dev).clone()
        outs = [torch.empty((HW * HW, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
        ws = ops.Workspace()
        opt = ShardedGridAdam(spec, dens, feat, lr=1e-2, **opt_kwargs)
        for step in range(1, K + 1):
            layout = abi.GRAD_ANY
            for i, (cam_id, (ro, rd, g)) in enumerate(my_cams):
                rng = (42, 1000 * step + cam_id)          # the jitter stream of a camera does not depend on who renders it
                ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *outs, ws, rng)
                layout = ops.render_bwd_acc(spec, params, dens, feat, ro, rd, None, outs[0], outs[1], outs[2], g, None, None,
                                            ws, rng, zero_first=(step == 1 and i == 0))
            opt.step(ws, layout)
        opt.gather_parameters()
        torch.cuda.synchronize()
        return dens, feat, opt.mode

    d2, f2, mode = run([(rank, cams[rank])], dict(exchange=exchange))
    # every rank also computes the one-process reference (cheap): the same optimiser with its collectives disabled
    class _Solo(ShardedGridAdam):
        def _collective(self):
            return False

    dens, feat = dens0.to(dev).clone(), feat0.to(dev).clone()
    outs = [torch.empty((HW * HW, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
    ws = ops.Workspace()
    solo = _Solo(spec, dens, feat, lr=1e-2)
    for step in range(1, K + 1):
        for i, (ro, rd, g) in enumerate(cams):
            rng = (42, 1000 * step + i)
            ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *outs, ws, rng)
            layout = ops.render_bwd_acc(spec, params, dens, feat, ro, rd, None, outs[0], outs[1], outs[2], g, None, None, ws,
                                        rng, zero_first=(step == 1 and i == 0))
        solo.step(ws, layout)
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    result = {"exchange": exchange, "mode": mode, "rel_densities": rel(d2, dens), "rel_features": rel(f2, feat),
              "moved": rel(feat, feat0.to(dev))}
    gathered = [None] * world
    dist.all_gather_object(gathered, result)
    if rank == 0:
        print(json.dumps({"ranks": gathered}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
