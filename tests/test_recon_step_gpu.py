"""voxe_recon_step (one reconstruction iteration in one library call) against the same iteration composed from the separate
entry points with the same random streams: batch selection, both renders, L1 losses, backward, Adam."""
import numpy as np
import pytest
import torch

from synth import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles
from voxe_hip import abi

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import ops

    DEV = torch.device("cuda:0")


def _setup(side, hw, nviews):
    dens, feat = random_grid(side)
    poses = []
    for i in range(nviews):
        p = pose_spherical(*synth_pose_angles(i, nviews), RADIUS)
        poses.append(torch.cat([p.rotation, p.translation], dim=-1))
    poses = torch.stack(poses).to(DEV).contiguous()
    images = torch.rand(nviews, 3, hw, hw, generator=torch.Generator().manual_seed(1)).to(DEV)
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS)
    params = ops.RenderParams(num_samples=64, near=NEAR, far=FAR, perturb=True, white_bkgd=True)
    return dens.to(DEV), feat.to(DEV), poses, images, spec, params


@pytest.mark.parametrize("batch,diffuse", [(20000, True), (3000, True), (20000, False)])
def test_recon_step_equals_the_composed_iteration(batch, diffuse):
    """20000 rays take the space-binned route, 3000 the ray-ordered scatter: both inside the one call"""
    side, hw, K = 48, 64, 6
    dens0, feat0, poses, images, spec, params = _setup(side, hw, K)
    rows = torch.tensor([4, 1, 5, 0, 2, 3], device=DEV)
    lr, steps = 2e-2, 3
    # ---- composed from the separate entry points (autograd + torch.optim.Adam arithmetic through VoxeAdam's kernel)
    d_a, f_a = dens0.clone().requires_grad_(True), feat0.clone().requires_grad_(True)
    opt = torch.optim.Adam([d_a, f_a], lr=lr, betas=(0.9, 0.999))
    ws = ops.Workspace()
    ref_losses = []
    for it in range(steps):
        seed, off = 11, 1000 * it
        subset = ops.random_subset(K * hw * hw, batch, DEV, rng=(seed, off))
        o, d = ops.cast_rays_indexed(hw, hw, focal_for(hw), poses, subset)
        cam = subset // (hw * hw)
        rem = subset - cam * hw * hw
        target = images[rows[cam], :, rem // hw, rem % hw]
        opt.zero_grad()
        c1 = ops.render(spec, params, d_a, f_a, o, d, workspace=ws, rng=(seed, off + 1))[0]
        loss = torch.nn.functional.l1_loss(c1, target)
        mse = torch.nn.functional.mse_loss(c1.detach(), target)
        l1d = torch.zeros(())
        if diffuse:
            import dataclasses

            c2 = ops.render(spec, dataclasses.replace(params, render_diffuse=True), d_a, f_a, o, d, workspace=ws, rng=(seed, off + 2))[0]
            l1d = torch.nn.functional.l1_loss(c2, target)
            loss = loss + l1d
        ref_losses.append((float(loss - l1d), float(mse), float(l1d)))
        loss.backward()
        opt.step()
    # ---- the one call
    d_b, f_b = dens0.clone(), feat0.clone()
    st_d = (torch.zeros_like(d_b), torch.zeros_like(d_b))
    st_f = (torch.zeros_like(f_b), torch.zeros_like(f_b))
    wa, wb = ops.Workspace(), ops.Workspace()
    losses = torch.zeros(4, device=DEV)
    for it in range(steps):
        ops.recon_step_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), poses, rows, images,
                        batch, diffuse, st_d, st_f, it + 1, it + 1, lr, losses, (11, 1000 * it), zero_gradient_first=(it == 0))
        got = losses.tolist()
        assert abs(got[0] - ref_losses[it][0]) < 2e-6 * max(1.0, abs(ref_losses[it][0])) + 2e-6, (it, got, ref_losses[it])
        assert abs(got[1] - ref_losses[it][1]) < 1e-5, (it, got, ref_losses[it])
        if diffuse:
            assert abs(got[2] - ref_losses[it][2]) < 2e-6 + 2e-6 * abs(ref_losses[it][2]), (it, got, ref_losses[it])
    torch.cuda.synchronize()
    moved = float((d_a.detach() - dens0).abs().max())
    assert moved > 1e-3
    # Adam's first steps turn the rounding noise of near-zero gradients (float atomics) into +-lr moves: measure against the
    # movement, like the two-rank tests
    for a, b in ((d_a.detach(), d_b), (f_a.detach(), f_b)):
        rel = float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(a - (dens0 if a.shape[-1] == 1 else feat0)))
        assert rel < 0.05, rel
    # the first iteration alone is deterministic up to float rounding of the gradient sums: |update| = lr for every voxel a
    # ray touched, so both runs must agree on WHICH voxels moved
    assert float(((d_a.detach() - dens0).abs() > 0).float().mean()) > 0.05


@pytest.mark.parametrize("batch", [20000, 3000])
def test_recon_step_first_iteration_against_the_oracle(batch):
    """the test above is HIP against HIP.  Here the first iteration of voxe_recon_step against the CPU oracle composed
    by hand the way modules/trainers.py:288-351 reads: random subset of the K * H * W pixels -> rays + target pixels -> specular
    and diffuse render (jitter streams offset + 1 / + 2) -> L1 losses and their gradients -> both backward passes summed -> Adam.
    Compared where Adam's normalisation has not yet amplified rounding: the losses, exp_avg = (1 - beta1) * gradient, and the
    parameters of every voxel whose gradient is far above the noise of the float atomics."""
    import dataclasses

    from oracle import voxe_oracle as vo
    from voxe_hip.desc import make_render_cfg

    side, hw, K = 40, 72, 5
    dens0, feat0, poses, images, spec, params = _setup(side, hw, K)
    rows = torch.tensor([3, 0, 4, 1, 2], device=DEV)
    lr, seed, off = 2e-2, 13, 4000
    # ---- oracle
    subset = vo.random_subset(K * hw * hw, batch, seed, off)
    o, d = vo.cast_rays_indexed(hw, hw, focal_for(hw), poses.cpu().numpy(), subset)
    cam, rem = subset // (hw * hw), subset % (hw * hw)
    target = images.cpu().numpy()[rows.cpu().numpy()[cam], :, rem // hw, rem % hw].astype(np.float32)
    grid = vo.Grid(dens0.cpu().numpy(), feat0.cpu().numpy(), [(-1.5, 1.5)] * 3, 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    gd_sum, gf_sum, ref_l1 = np.zeros_like(grid.densities), np.zeros_like(grid.features), []
    for i, diffuse in enumerate((False, True)):
        cfg = make_render_cfg(params.num_samples, NEAR, FAR, perturb=True, white_bkgd=True, seed=seed, rng_offset=off + 1 + i,
                              render_diffuse=diffuse)
        col = vo.render_fwd(grid, cfg, o, d)["colour"]
        diff = col - target
        ref_l1.append(float(np.abs(diff).astype(np.float64).mean()))
        g_col = (np.sign(diff) * np.float32(1.0 / diff.size)).astype(np.float32)
        gd, gf = vo.render_bwd(grid, cfg, o, d, g_col)
        gd_sum += gd
        gf_sum += gf
    pd, pf = grid.densities.reshape(-1).copy(), grid.features.reshape(-1).copy()
    for p_, g_ in ((pd, gd_sum.reshape(-1)), (pf, gf_sum.reshape(-1))):
        vo.adam_step(p_, np.ascontiguousarray(g_), np.zeros_like(p_), np.zeros_like(p_), lr, 0.9, 0.999, 1e-8, 1)
    # ---- the one call
    d_b, f_b = dens0.clone(), feat0.clone()
    st_d = (torch.zeros_like(d_b), torch.zeros_like(d_b))
    st_f = (torch.zeros_like(f_b), torch.zeros_like(f_b))
    losses = torch.zeros(4, device=DEV)
    ops.recon_step_(spec, params, d_b, f_b, ops.Workspace(), ops.Workspace(), hw, hw, focal_for(hw), poses, rows, images, batch, True,
                    st_d, st_f, 1, 1, lr, losses, (seed, off), zero_gradient_first=True)
    got = losses.tolist()
    assert abs(got[0] - ref_l1[0]) < 2e-6 and abs(got[2] - ref_l1[1]) < 2e-6, (got, ref_l1)
    for m1, g_ref, p_lib, p_ref in ((st_d[0], gd_sum, d_b, pd), (st_f[0], gf_sum, f_b, pf)):
        g_lib = m1.cpu().numpy().reshape(-1) / np.float32(0.1)
        g_ref = g_ref.reshape(-1)
        assert np.linalg.norm(g_lib - g_ref) / np.linalg.norm(g_ref) < 1e-4
        big = np.abs(g_ref) > 1e-3 * np.abs(g_ref).max()
        assert big.sum() > 100
        assert np.abs(p_lib.cpu().numpy().reshape(-1) - p_ref)[big].max() < 1e-3 * lr


def test_fused_grid_adam_reconstruction_step_drives_the_trainer_state():
    """the optimiser-level wrapper: step counters, learning rate and Adam state are the optimiser's"""
    from thre3d_atom.modules.optim import FusedGridAdam
    from thre3d_atom.modules.volumetric_model import VolumetricModel
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, _render_params, render_sh_voxel_grid
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize
    from thre3d_atom.utils.imaging_utils import CameraBounds

    side, hw, K = 32, 48, 4
    dens, feat, poses, images, _, _ = _setup(side, hw, K)
    vg = VoxelGrid(dens.cpu(), feat.cpu(), VoxelSize(3.0 / side, 3.0 / side, 3.0 / side), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=3.0, tunable=True)
    model = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(48, CameraBounds(NEAR, FAR), white_bkgd=True), device=DEV)
    grid = model.thre3d_repr
    before = grid.densities.detach().clone()
    losses = torch.zeros(4, device=DEV)
    params = _render_params(grid, None, model.render_config, attn=False)
    with FusedGridAdam(grid, lr=1e-2) as opt:
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
        first = None
        for it in range(6):
            opt.reconstruction_step(params, hw, hw, focal_for(hw), poses, None, images, 4096, True, losses, (5, 100 * it))
            if it == 2:
                sched.step()
            if first is None:
                first = losses.tolist()
        assert opt.state[grid.densities]["step"] == 6 and opt.state[grid.features]["step"] == 6
        assert abs(opt.param_groups[0]["lr"] - 5e-3) < 1e-9
        last = losses.tolist()
    assert last[0] + last[2] < first[0] + first[2]                       # the loss went down
    assert float((grid.densities.detach() - before).abs().max()) > 1e-3
    # ordinary renders work again afterwards (the mode ended with the context manager)
    assert grid.voxe_workspace("sh").deferred is None


def test_recon_step_bounds_its_image_rows_and_a_failed_call_leaves_the_optimiser_consistent():
    """(a) image_rows are bounded by VoxeReconStep::num_images -- a row outside the image stack reads nothing and
    shows up as a NaN loss, K cameras without a row table need K <= N; (b) a reconstruction_step that raises does not advance
    the optimiser's step counters, leaves no half-accumulated gradient behind, and the scheduler sees optimiser steps"""
    import warnings

    from voxe_hip.runtime import VoxeError

    side, hw, K = 32, 48, 4
    dens, feat, poses, images, spec, params = _setup(side, hw, K)
    st_d = (torch.zeros_like(dens), torch.zeros_like(dens))
    st_f = (torch.zeros_like(feat), torch.zeros_like(feat))
    losses = torch.zeros(4, device=DEV)
    d_b, f_b = dens.clone(), feat.clone()
    bad_rows = torch.tensor([0, 1, K, 2], device=DEV)                     # row K is one past the stack
    ops.recon_step_(spec, params, d_b, f_b, ops.Workspace(), ops.Workspace(), hw, hw, focal_for(hw), poses, bad_rows, images,
                    4096, False, st_d, st_f, 1, 1, 1e-2, losses, (3, 0))
    assert bool(torch.isnan(losses[0]))
    with pytest.raises(VoxeError):                                        # K cameras, no row table, only K - 1 images
        ops.recon_step_(spec, params, dens.clone(), feat.clone(), ops.Workspace(), ops.Workspace(), hw, hw, focal_for(hw), poses,
                        None, images[: K - 1].contiguous(), 4096, False, st_d, st_f, 1, 1, 1e-2, losses, (3, 0))

    from thre3d_atom.modules.optim import FusedGridAdam
    from thre3d_atom.modules.volumetric_model import VolumetricModel
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, _render_params, render_sh_voxel_grid
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize
    from thre3d_atom.utils.imaging_utils import CameraBounds

    vg = VoxelGrid(dens.cpu(), feat.cpu(), VoxelSize(3.0 / side, 3.0 / side, 3.0 / side), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=3.0, tunable=True)
    model = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(48, CameraBounds(NEAR, FAR), white_bkgd=True), device=DEV)
    grid = model.thre3d_repr
    rp = _render_params(grid, None, model.render_config, attn=False)
    with FusedGridAdam(grid, lr=1e-2) as opt:
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
        opt.reconstruction_step(rp, hw, hw, focal_for(hw), poses, None, images, 4096, True, losses, (5, 0))
        with warnings.catch_warnings():
            warnings.simplefilter("error")       # "lr_scheduler.step() before optimizer.step()" must not fire
            sched.step()
        ws = grid.voxe_workspace("sh")
        with pytest.raises(VoxeError):           # fewer images than cameras: rejected inside the library call
            opt.reconstruction_step(rp, hw, hw, focal_for(hw), poses, None, images[: K - 1].contiguous(), 4096, True, losses, (5, 1))
        assert opt.state[grid.densities]["step"] == 1 and opt.state[grid.features]["step"] == 1
        assert ws.deferred.clean_ptr == 0 and ws.key is None      # the next call clears the gradient region and re-packs
        opt.reconstruction_step(rp, hw, hw, focal_for(hw), poses, None, images, 4096, True, losses, (5, 2))
        assert opt.state[grid.densities]["step"] == 2 and bool(torch.isfinite(losses).all())


def _prefetch_stats():
    import ctypes

    out = (ctypes.c_int64 * 3)()
    assert ops.lib().voxe_recon_prefetch_stats(out) == 0
    return list(out)


def _run_recon(hint, steps, batch=20000, wrong_hint_at=(), side=48, hw=64, K=6):
    """`steps` iterations over changing cameras; hint: None | "ahead" (voxe_recon_prefetch after every step, cameras drawn BEFORE the
    step as the trainer does).  Iterations in `wrong_hint_at` are announced with ANOTHER stream offset than the step then passes."""
    dens0, feat0, poses_all, images, spec, params = _setup(side, hw, 12)
    gen = torch.Generator().manual_seed(5)
    d_b, f_b = dens0.clone(), feat0.clone()
    st_d = (torch.zeros_like(d_b), torch.zeros_like(d_b))
    st_f = (torch.zeros_like(f_b), torch.zeros_like(f_b))
    wa, wb = ops.Workspace(), ops.Workspace()
    losses = torch.zeros(4, device=DEV)
    out = []

    def draw(it):
        rows = torch.randint(0, 12, (K,), generator=gen).to(DEV)
        return rows, poses_all[rows].contiguous(), (11, 1000 * it)

    upcoming = draw(0)
    for it in range(steps):
        rows, poses, rng = upcoming
        upcoming = draw(it + 1)
        ops.recon_step_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), poses, rows, images, batch, True, st_d, st_f, it + 1,
                        it + 1, 2e-2, losses, rng, zero_gradient_first=(it == 0))
        if hint == "ahead":
            nrng = upcoming[2] if (it + 1) not in wrong_hint_at else (11, 1000 * (it + 1) + 500)
            ops.recon_prefetch_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), upcoming[1], upcoming[0], images, batch, True,
                                losses, nrng)
        out.append(losses.tolist())
    torch.cuda.synchronize()
    return out, d_b, f_b, dens0, feat0


def test_recon_prefetch_changes_nothing_but_the_schedule():
    """voxe_recon_prefetch (ABI v12): the batch + segment tables of iteration i + 1 assembled on the library's side stream behind
    iteration i's forward.  Same batches, same tables -> the same iterations: the first iteration's losses are bit-identical (the
    forward is deterministic), the later ones agree to the rounding noise of the float-atomic gradient sums, and every hint is
    TAKEN (the stats say so).  A hint for other arguments than the step's is dropped and costs nothing but the wait."""
    steps = 7
    s0 = _prefetch_stats()
    plain, d_p, f_p, dens0, feat0 = _run_recon(None, steps)
    # (no hint: none issued, none taken; a STALE hint an earlier test left on record at an address the allocator hands out again is
    #  dropped by the first step that meets it -- the "dropped" counter may move by one per run for that reason, never more)
    sp = _prefetch_stats()
    assert sp[:2] == s0[:2] and sp[2] - s0[2] in (0, 1), (s0, sp)
    ahead, d_a, f_a, _, _ = _run_recon("ahead", steps)
    s1 = _prefetch_stats()
    assert s1[0] - s0[0] == steps and s1[1] - s0[1] == steps - 1 and s1[2] - sp[2] in (0, 1), (s0, sp, s1)   # (the last hint has no step behind it)
    assert plain[0] == ahead[0]
    for it in range(steps):
        for j in (0, 1, 2, 3):
            assert abs(plain[it][j] - ahead[it][j]) < 2e-6 + 2e-5 * abs(plain[it][j]), (it, j, plain[it], ahead[it])
    for a, b, x0 in ((d_p, d_a, dens0), (f_p, f_a, feat0)):
        rel = float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(a - x0))
        assert rel < 0.05, rel
    # two of the hints announce another jitter stream than the step then uses: dropped, the iterations are the plain ones
    mixed, d_m, f_m, _, _ = _run_recon("ahead", steps, wrong_hint_at=(2, 5))
    s2 = _prefetch_stats()
    # (+ 1 when the allocator hands this run the previous run's workspace address: that run's last hint is still on record there
    #  and is dropped by the first step here -- a stale hint costs a wait, nothing else)
    assert s2[2] - s1[2] in (2, 3) and s2[1] - s1[1] == steps - 3, (s1, s2)
    assert plain[0] == mixed[0]
    for it in range(steps):
        for j in (0, 1, 2, 3):
            assert abs(plain[it][j] - mixed[it][j]) < 2e-6 + 2e-5 * abs(plain[it][j]), (it, j, plain[it], mixed[it])


def test_recon_prefetch_is_dropped_where_no_paired_render_runs():
    """small batches (ray-ordered route) and view-dependent grids take no hint: the call returns without touching anything"""
    s0 = _prefetch_stats()
    out, *_ = _run_recon("ahead", 3, batch=3000)
    assert _prefetch_stats() == s0
    ref, *_ = _run_recon(None, 3, batch=3000)
    assert out[0] == ref[0]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_recon_prefetch_random_call_sequences(seed):
    """hints in every order a caller could produce -- none, right, wrong stream, wrong cameras, two in a row, a batch size that regrows
    the workspaces (the binding synchronises before it frees a buffer the side stream may still write), a stale hint from a smaller
    batch -- against the same iterations without any hint: the losses of every iteration agree to the float-atomic noise of the
    gradient sums and the first iteration exactly."""
    import random

    rnd = random.Random(100 + seed)
    side, hw, K, steps = 40, 64, 6, 9
    plan = []
    for it in range(steps):
        plan.append({"batch": rnd.choice([18000, 18000, 18000, 22000]), "hint": rnd.choice(["right", "right", "none", "wrong_rng", "wrong_cams", "twice"])})

    def run(with_hints):
        dens0, feat0, poses_all, images, spec, params = _setup(side, hw, 10)
        gen = torch.Generator().manual_seed(7 + seed)
        d_b, f_b = dens0.clone(), feat0.clone()
        st_d = (torch.zeros_like(d_b), torch.zeros_like(d_b))
        st_f = (torch.zeros_like(f_b), torch.zeros_like(f_b))
        wa, wb = ops.Workspace(), ops.Workspace()
        losses = torch.zeros(4, device=DEV)
        cams = [torch.randint(0, 10, (K,), generator=gen).to(DEV) for _ in range(steps + 1)]
        cam_poses = [poses_all[c].contiguous() for c in cams]
        other = poses_all[torch.randint(0, 10, (K,), generator=gen).to(DEV)].contiguous()
        torch.cuda.synchronize()
        out = []
        for it in range(steps):
            ops.recon_step_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), cam_poses[it], cams[it], images, plan[it]["batch"], True,
                            st_d, st_f, it + 1, it + 1, 2e-2, losses, (3, 100 * it), zero_gradient_first=(it == 0))
            out.append(losses.tolist())
            if not with_hints or it + 1 >= steps:
                continue
            kind, nb = plan[it]["hint"], plan[it + 1]["batch"]

            def hint(poses, rng, batch=nb):
                ops.recon_prefetch_(spec, params, d_b, f_b, wa, wb, hw, hw, focal_for(hw), poses, cams[it + 1], images, batch, True, losses, rng)

            if kind == "right":
                hint(cam_poses[it + 1], (3, 100 * (it + 1)))
            elif kind == "wrong_rng":
                hint(cam_poses[it + 1], (3, 100 * (it + 1) + 7))
            elif kind == "wrong_cams":
                hint(other, (3, 100 * (it + 1)))
            elif kind == "twice":
                hint(other, (3, 100 * (it + 1) + 9), batch=plan[it]["batch"])
                hint(cam_poses[it + 1], (3, 100 * (it + 1)))
        torch.cuda.synchronize()
        return out

    plain, hinted = run(False), run(True)
    assert plain[0] == hinted[0]
    for it in range(steps):
        for j in range(4):
            assert abs(plain[it][j] - hinted[it][j]) < 2e-6 + 5e-5 * abs(plain[it][j]), (it, j, plan[it], plain[it], hinted[it])
