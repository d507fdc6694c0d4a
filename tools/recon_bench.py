"""Per-iteration time of the reconstruction trainer's inner loop at BASELINE.json configs[1] scale: 160^3 SH-0
softplus field, 100 views @ 400x400 (synthetic images), 32768 random rays over 8 cached images per iteration,
specular + diffuse L1, fused Adam.   gpurun -- python tools/recon_bench.py [iters]
RECON_PYTHON_ITER=1: the iteration composed in Python from the separate ops (r02) instead of voxe_recon_step;
RECON_NO_PREFETCH=1: without the hint that lets the next iteration's batch / segment tables be assembled ahead (r06)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]

import torch  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles  # noqa: E402
from thre3d_atom.modules.optim import FusedGridAdam, VoxeAdam  # noqa: E402
from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.rendering.volumetric.utils.misc import (  # noqa: E402
    cast_rays,
    collate_rays,
    flatten_rays,
    sample_random_rays_and_pixels_from_cameras,
    sample_random_rays_and_pixels_synchronously,
)
from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid  # noqa: E402
from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize  # noqa: E402
from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, CameraPose, pose_spherical  # noqa: E402
from voxe_hip import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    G, HW, NV, B = 160, 400, 100, 32768
    dens, feat = random_grid(G)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    vm = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(256, CameraBounds(NEAR, FAR), white_bkgd=True),
                         device=dev)
    intr = CameraIntrinsics(HW, HW, focal_for(HW))
    poses = []
    for i in range(NV):
        p = pose_spherical(*synth_pose_angles(i, NV), RADIUS)
        poses.append(torch.cat([p.rotation, p.translation], dim=-1))
    poses = torch.stack(poses).to(dev)
    images = torch.rand(NV, 3, HW, HW, device=dev)
    if os.environ.get("RECON_SPLIT_ADAM"):   # the ordinary path: .grad tensors (un-pack) + one Adam kernel per tensor
        opt = VoxeAdam([{"params": vm.thre3d_repr.parameters(), "lr": 0.03}])
    else:                                    # what the trainer uses: gradient left in the workspace + one fused pass
        opt = FusedGridAdam(vm.thre3d_repr, lr=0.03)
    gen = torch.Generator().manual_seed(0)
    marks = {}

    def mark(name, t0):
        torch.cuda.synchronize()
        marks[name] = marks.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()

    one_call = not os.environ.get("RECON_PYTHON_ITER") and isinstance(opt, FusedGridAdam)
    losses = torch.zeros(4, device=dev)
    from thre3d_atom.thre3d_reprs.renderers import _render_params
    rparams = _render_params(vm.thre3d_repr, None, vm.render_config, attn=False)

    prefetch = one_call and not os.environ.get("RECON_NO_PREFETCH")
    upcoming = []

    from thre3d_atom.modules.trainers import _PinnedStaging
    staging = _PinnedStaging(8, dev)

    def draw():
        if os.environ.get("RECON_PAGEABLE_PICKS"):   # (the r05 loop: a pageable host-to-device copy per iteration waits for the stream)
            picks = torch.randint(0, NV, (8,), generator=gen).to(dev)
        else:
            picks = staging.to_device(lambda out: torch.randint(0, NV, (out.numel(),), generator=gen, out=out))
        return picks, poses[picks].contiguous(), ops._next_rng()

    def iteration(profile):
        t = time.perf_counter()
        if one_call:   # what the trainer runs: the whole iteration as ONE library call (voxe_recon_step), the next iteration's
            #            cameras drawn BEFORE it and announced behind it (voxe_recon_prefetch; RECON_NO_PREFETCH=1: no hint)
            if not upcoming:
                upcoming.append(draw())
            picks, poses_it, rng_it = upcoming.pop()
            upcoming.append(draw())
            opt.reconstruction_step(rparams, HW, HW, focal_for(HW), poses_it, picks, images, B, True, losses, rng_it)
            if prefetch:
                opt.reconstruction_prefetch(rparams, HW, HW, focal_for(HW), upcoming[0][1], upcoming[0][0], images, B, True, losses,
                                            upcoming[0][2])
            if profile:
                mark("whole iteration (one library call)", t)
            return
        if not os.environ.get("RECON_OLD_BATCH") and not os.environ.get("RECON_SORT"):
            picks = torch.randint(0, NV, (8,), generator=gen).to(dev)
            rays_b, pix_b = sample_random_rays_and_pixels_from_cameras(intr, poses[picks], images, B, image_ids=picks,
                                                                       memory_order=not os.environ.get("RECON_DRAW_ORDER"),
                                                                       fast_subset=not os.environ.get("RECON_RANDPERM"))
            rays = pixels = None
        else:
            picks = torch.randint(0, NV, (8,), generator=gen).tolist()
            rays = collate_rays([flatten_rays(cast_rays(intr, CameraPose(poses[i][:, :3], poses[i][:, 3:]), device=dev))
                                 for i in picks])
            pixels = torch.cat([images[i].permute(1, 2, 0).reshape(-1, 3) for i in picks])
        if rays is None:
            pass
        elif os.environ.get("RECON_SORT"):
            tile = int(os.environ["RECON_SORT"])
            n = rays.origins.shape[0]
            subset = torch.randperm(n, device=dev)[:B]
            img, rem = subset // (HW * HW), subset % (HW * HW)
            y, x = rem // HW, rem % HW
            nt = (HW + tile - 1) // tile
            key = ((img * nt + y // tile) * nt + x // tile) * (tile * tile) + (y % tile) * tile + x % tile
            subset = subset[torch.argsort(key)]
            from thre3d_atom.rendering.volumetric.render_interface import Rays
            rays_b, pix_b = Rays(rays.origins[subset], rays.directions[subset]), pixels[subset]
        else:
            rays_b, pix_b = sample_random_rays_and_pixels_synchronously(rays, pixels, B)
        if profile:
            t = mark("batch assembly (rays + target pixels)", t)
        spec = vm.render_rays(rays_b).colour
        loss = torch.nn.functional.l1_loss(spec, pix_b)
        diff = vm.render_rays(rays_b, render_diffuse=True).colour
        loss = loss + torch.nn.functional.l1_loss(diff, pix_b)
        if profile:
            t = mark("2 forward renders + loss", t)
        opt.zero_grad()
        loss.backward()
        if profile:
            t = mark("backward (2 renders)", t)
        opt.step()
        if profile:
            mark("Adam", t)

    for _ in range(5):
        iteration(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration(False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"reconstruction iteration (160^3, 8x400x400 cached, {B} random rays, spec+diffuse): {dt * 1e3:.3f} ms "
          f"-> {1 / dt:.1f} it/s, {2 * B / dt / 1e6:.1f} M rendered rays/s (fwd+bwd)")
    ops.profile_enable(True)
    for _ in range(10):
        iteration(True)
    for k, v in marks.items():
        print(f"  {k:45s} {v / 10 * 1e3:8.3f} ms")
    print("  kernel phases (per iteration):", {k: round(v / 10, 4) for k, v in ops.profile_read().items() if k.startswith("ms_")})


if __name__ == "__main__":
    main()
