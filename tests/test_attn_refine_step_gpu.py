"""voxe_attn_refine_step (one attention grid's share of a refinement iteration in one library call, BASELINE configs[3]) against
(a) the same iterations composed the way the reference writes them (modules/attn_grid_trainer.py:335-378: render_rays_attn ->
calc_loss_on_attn_grid -> + attn_tv_weight * _tv_loss_on_grid -> backward -> torch.optim.Adam) through the binding's autograd
entry points, and (b) the CPU oracle composed by hand for the first step (render + gradient, TV, Adam)."""
import numpy as np
import pytest
import torch

from synth import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles
from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from oracle import voxe_oracle as vo
    from thre3d_atom.modules.refinement_functions import calc_loss_on_attn_grid
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import ops

    DEV = torch.device("cuda:0")


def _setup(side, hw, S, camera=3):
    dens, _ = random_grid(side)
    g = torch.Generator().manual_seed(9)
    attn0 = (torch.rand((side, side, side, 1), generator=g) * 3.0 - 1.0)          # positive and negative attention values
    amap = torch.rand((hw, hw), generator=g)
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=6.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS, feature_kind=abi.FEAT_ATTN)
    p = pose_spherical(*synth_pose_angles(camera, 100), RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), p.rotation, p.translation, DEV)
    params = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw)
    return dens.to(DEV), attn0.to(DEV), amap.to(DEV), spec, params, ro, rd


@pytest.mark.parametrize("hw,tv_weight", [(96, 0.01), (96, 0.0), (40, 0.5)])
def test_refine_step_equals_the_composed_iterations(hw, tv_weight):
    """96x96 rays take the LDS-window kernels, 40x40 the scatter route; three Adam steps each"""
    side, S, lr, steps = 40, 64, 0.035, 3
    dens, attn0, amap, spec, params, ro, rd = _setup(side, hw, S)
    # ---- (a) composed: autograd render + the reference's loss expressions + torch.optim.Adam
    a_ref = attn0.clone().requires_grad_(True)
    opt = torch.optim.Adam([a_ref], lr=lr, betas=(0.9, 0.999))
    ws = ops.Workspace()
    ref_losses, ref_render = [], None
    for it in range(steps):
        att = ops.render(spec, params, dens, a_ref, ro, rd, workspace=ws, rng=(5, 100 + it))[0]
        l1 = calc_loss_on_attn_grid(att, amap)
        tv = ops.tv_loss_on_grid(a_ref)
        (l1 + tv * tv_weight).backward()
        opt.step()
        opt.zero_grad()
        ref_losses.append((float(l1), float(tv)))
        if it == 0:
            ref_render = att.detach().clone()
    # ---- the library call
    a_lib = attn0.clone()
    state = (torch.zeros_like(a_lib), torch.zeros_like(a_lib))
    ws2 = ops.Workspace()
    losses = torch.zeros(2, device=DEV)
    render = torch.empty(hw * hw, device=DEV)
    got_losses = []
    for it in range(steps):
        ops.attn_refine_step_(spec, params, dens, a_lib, ro, rd, amap.reshape(-1), ws2, it + 1, lr, state, tv_weight, losses,
                              rng=(5, 100 + it), attn_render=render)
        got_losses.append(tuple(float(v) for v in losses))
        if it == 0:
            assert torch.equal(render, ref_render.reshape(-1))         # the same forward kernel on the same inputs
    for (l1, tv), (gl1, gtv) in zip(ref_losses, got_losses):
        assert abs(l1 - gl1) <= 2e-6 * max(1.0, abs(l1)), (l1, gl1)
        assert abs(tv - gtv) <= 2e-6 * max(1.0, abs(tv)), (tv, gtv)
    moved = float((a_ref.detach() - attn0).abs().max())
    assert moved > 0.5 * lr                                            # Adam's first steps move every touched voxel by ~lr
    # Adam normalises: a voxel whose tiny gradient differs in the last bits (float atomics: summation order) can move differently
    # by a visible fraction of lr; compare in the L2 sense and bound the outliers
    diff = (a_lib - a_ref.detach()).abs()
    rel = float(torch.linalg.norm(diff) / torch.linalg.norm(a_ref.detach() - attn0))
    assert rel < 2e-3, rel
    assert float((diff > 0.1 * lr).float().mean()) < 1e-3


def test_refine_step_first_iteration_against_the_oracle():
    side, hw, S, lr, tv_weight = 32, 64, 48, 0.035, 0.02
    dens, attn0, amap, spec, params, ro, rd = _setup(side, hw, S, camera=12)
    grid = vo.Grid(dens.cpu().numpy(), attn0.cpu().numpy(), [(-1.5, 1.5)] * 3, 6.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_ATTN)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True, perturb=True, seed=5, rng_offset=77)
    o, d = ro.cpu().numpy(), rd.cpu().numpy()
    out = vo.render_fwd(grid, cfg, o, d)
    r = np.asarray(out["colour"], dtype=np.float32).reshape(-1)
    m = amap.cpu().numpy().reshape(-1)
    mask = (r > 0).astype(np.float32)
    msum = np.float32(mask.sum())
    loss_ref = np.float32(np.abs(r - m)[mask > 0].astype(np.float64).sum()) / msum
    g_r = ((np.float32(1.0) / msum) * mask) * np.sign(r - m).astype(np.float32)
    gd, gf = vo.render_bwd(grid, cfg, o, d, g_r.reshape(-1, 1))
    tv_ref, tv_g = vo.tv_fwd_bwd(attn0.cpu().numpy(), tv_weight)
    total = (np.asarray(gf, dtype=np.float32) + np.asarray(tv_g, dtype=np.float32)).reshape(-1)
    p = attn0.cpu().numpy().reshape(-1).copy()
    m1, m2 = np.zeros_like(p), np.zeros_like(p)
    vo.adam_step(p, total, m1, m2, lr, 0.9, 0.999, 1e-8, 1)
    # the library call
    a_lib = attn0.clone()
    state = (torch.zeros_like(a_lib), torch.zeros_like(a_lib))
    losses = torch.zeros(2, device=DEV)
    render = torch.empty(hw * hw, device=DEV)
    ops.attn_refine_step_(spec, params, dens, a_lib, ro, rd, amap.reshape(-1), ops.Workspace(), 1, lr, state, tv_weight, losses,
                          rng=(5, 77), attn_render=render)
    assert np.abs(render.cpu().numpy() - r).max() < 2e-6
    assert abs(float(losses[0]) - float(loss_ref)) < 2e-6 * max(1.0, abs(float(loss_ref)))
    assert abs(float(losses[1]) - float(tv_ref)) < 2e-6 * max(1.0, abs(float(tv_ref)))
    # exp_avg = (1 - beta1) * gradient: the gradient itself, before Adam's normalisation amplifies last-bit differences
    g_lib = state[0].cpu().numpy().reshape(-1) / np.float32(0.1)
    rel = np.linalg.norm(g_lib - total) / np.linalg.norm(total)
    assert rel < 1e-4, rel
    got = a_lib.cpu().numpy().reshape(-1)
    big = np.abs(total) > 1e-3 * np.abs(total).max()                    # voxels whose gradient is far above rounding noise
    assert np.abs(got - p)[big].max() < 1e-3 * lr, np.abs(got - p)[big].max()


def test_refine_step_rejects_what_it_cannot_run():
    from voxe_hip.runtime import VoxeError
    side, hw, S = 16, 24, 16
    dens, attn0, amap, spec, params, ro, rd = _setup(side, hw, S)
    state = (torch.zeros_like(attn0), torch.zeros_like(attn0))
    sh_spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=6.0)
    with pytest.raises(VoxeError):
        ops.attn_refine_step_(sh_spec, params, dens, attn0.clone(), ro, rd, amap.reshape(-1), ops.Workspace(), 1, 0.01, state, 0.0)
    with pytest.raises(VoxeError):
        ops.attn_refine_step_(spec, params, dens, attn0.clone(), ro, rd, amap.reshape(-1)[:-1].contiguous(), ops.Workspace(), 1, 0.01,
                              state, 0.0)


def test_masked_l1_kernel_vs_the_reference_golden(golden):
    """voxe_attn_masked_l1 against calc_loss_on_attn_grid of the REFERENCE (modules/refinement_functions.py:42-77; loss and autograd
    gradient recorded by tools/gen_golden.py in refine_graph.npz), against this package's Python restatement, and on a large image
    (several blocks, every element exercised: gradient bit for bit against the torch expression)"""
    g = golden("refine_graph.npz")
    for tag in ("a", "b"):
        render, amap = torch.from_numpy(g[f"loss_{tag}_render"]).to(DEV), torch.from_numpy(g[f"loss_{tag}_map"]).to(DEV)
        loss, grad = ops.attn_masked_l1(render, amap)
        assert abs(float(loss) - float(g[f"loss_{tag}_value"])) < 1e-6
        np.testing.assert_allclose(grad.cpu().numpy(), g[f"loss_{tag}_grad"], rtol=1e-6, atol=0)
    gen = torch.Generator().manual_seed(3)
    render = (torch.rand((400 * 400, 1), generator=gen) * 1.5 - 0.5).to(DEV)
    render[::7] = 0.0                                      # exactly on the mask's edge (render > 0 is false)
    amap = torch.rand((400, 400), generator=gen).to(DEV)
    render[5::11, 0] = amap.reshape(-1)[5::11]             # exact ties: sign(0) = 0
    r = render.clone().requires_grad_(True)
    ref = calc_loss_on_attn_grid(r, amap)
    ref.backward()
    loss, grad = ops.attn_masked_l1(render, amap)
    assert abs(float(loss) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    assert torch.equal(grad, r.grad)
    # nothing above zero: 0 / 0 like the reference
    loss0, grad0 = ops.attn_masked_l1(-torch.ones((50, 1), device=DEV), torch.rand((5, 10), device=DEV))
    assert torch.isnan(loss0) and torch.isnan(grad0).all()
