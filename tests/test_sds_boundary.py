"""SDS boundary: the build's thre3d_atom/thre3d_reprs/sd.py against what the REFERENCE's sd.py
returned on the same stand-in SD stack (tests/sds_standins.py) with the same random draws
(tests/golden/sds_boundary.npz, written by tools/gen_golden.py::g16_sds_boundary by importing the reference).
Pinned: image layout and 512^2 bilinear resize, 2x-1 / 0.18215 latent scaling through the differentiated VAE encoder,
timestep range + max-step schedule (gamma, 0.22 floor), add_noise, classifier-free guidance at scale 100,
w(t) = 1 - alpha_bar_t, nan_to_num, the log-variance factor, SpecifyGradient (zero loss value, gradient / batch)."""
import os

import numpy as np
import pytest
import torch

import sds_standins as st
from conftest import load_golden
from helpers import rel_l2

DEVICES = [pytest.param("cpu", id="cpu"), pytest.param("cuda", id="cuda", marks=pytest.mark.gpu)]


def _guidance(device, **kw):
    with st.installed():
        from thre3d_atom.thre3d_reprs.sd import scoreDistillationLoss

        return scoreDistillationLoss(torch.device(device), "a yarn doll", **kw)


def _draws(g, tag):
    return [torch.from_numpy(g[f"{tag}_draw{i}"]) for i in range(int(g[f"{tag}_ndraws"]))]


@pytest.mark.parametrize("device", DEVICES)
def test_training_step_matches_the_reference(device):
    g = load_golden("sds_boundary.npz")
    h, w = (int(v) for v in g["hw"])
    guide = _guidance(device, t_sched_start=2, t_sched_freq=2, t_sched_gamma=0.5, directional=True)
    np.testing.assert_allclose(guide.text_encodings["front"].cpu().numpy(), g["text_front"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(guide.sd_model.alphas.cpu().numpy(), g["alphas"], rtol=1e-6, atol=0)
    assert guide.sd_model.get_num_tokens("a yarn doll") == int(g["num_tokens"])
    tol = 2e-5 if device == "cpu" else 2e-4       # (GPU convolutions accumulate in another order)
    for step in range(1, int(g["steps"]) + 1):
        tag = f"s{step}"
        colour = torch.from_numpy(g[f"{tag}_colour"]).to(device).requires_grad_(True)
        with st.RandomTape(record=False, draws=_draws(g, tag), device=device) as tape:
            loss = guide.training_step(colour, h, w, directions=[str(g["directions"][step - 1])], global_step=step)
            loss.backward()
        assert tape._cursor == int(g[f"{tag}_ndraws"])                    # same number of random draws, same order
        assert float(loss.detach()) == float(g[f"{tag}_loss"][0]) == 0.0   # the SDS "loss" is a dummy zero (sd.py:20-34)
        assert guide.get_current_max_step_ratio() == pytest.approx(float(g[f"{tag}_ratio"]), abs=1e-12)
        assert guide.sd_model.max_step == int(g[f"{tag}_max_step"])
        assert rel_l2(colour.grad.cpu().numpy(), g[f"{tag}_grad"]) < tol, (step, rel_l2(colour.grad.cpu().numpy(), g[f"{tag}_grad"]))
    assert guide.get_current_max_step_ratio() == pytest.approx(0.22)       # 0.1225 was floored


@pytest.mark.parametrize("device", DEVICES)
def test_non_directional_step_with_log_variance(device):
    g = load_golden("sds_boundary.npz")
    h, w = (int(v) for v in g["hw"])
    guide = _guidance(device, directional=False)
    colour = torch.from_numpy(g["b_colour"]).to(device).requires_grad_(True)
    with st.RandomTape(record=False, draws=_draws(g, "b"), device=device):
        guide.training_step(colour, h, w, global_step=7, logvars=torch.tensor(float(g["b_logvar"]), device=device)).backward()
    assert rel_l2(colour.grad.cpu().numpy(), g["b_grad"]) < (2e-5 if device == "cpu" else 2e-4)


def test_directional_guidance_needs_directions_and_specify_gradient_divides_by_the_batch():
    from thre3d_atom.thre3d_reprs.sd import SpecifyGradient

    guide = _guidance("cpu", directional=True)
    with pytest.raises(AssertionError):
        guide.training_step(torch.rand(20 * 28, 3), 20, 28)
    x = torch.zeros(2, 4, 3, 3, requires_grad=True)
    grad = torch.arange(72, dtype=torch.float32).reshape(2, 4, 3, 3)
    loss = SpecifyGradient.apply(x, grad)
    assert loss.shape == (1,) and float(loss.detach()) == 0.0
    loss.backward()
    assert torch.equal(x.grad, grad / 2)


@pytest.mark.gpu
def test_sds_gradient_flows_into_the_hip_renderer():
    """the boundary end to end on the GPU: HIP render -> stand-in guidance (build's sd.py) -> backward through the VAE
    stand-in into dL/dcolour -> HIP backward: finite, non-zero grid gradients; the same dL/dcolour fed to the oracle gives
    the same grid gradient"""
    from synth import FAR, NEAR, RADIUS, focal_for, sphere_grid, synth_pose_angles
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import abi, ops
    from voxe_hip.desc import make_render_cfg

    from oracle import voxe_oracle as vo

    dev = torch.device("cuda")
    G, hw = 48, 40
    dens, feat = sphere_grid(G)
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS)
    pose = pose_spherical(*synth_pose_angles(4, 100), RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
    params = ops.RenderParams(num_samples=64, near=NEAR, far=FAR, white_bkgd=True, image_width=hw)
    d, f = dens.to(dev).requires_grad_(True), feat.to(dev).requires_grad_(True)
    colour, _, _, _ = ops.render(spec, params, d, f, ro, rd)
    colour.retain_grad()
    guide = _guidance("cuda", directional=True)
    torch.manual_seed(3)
    guide.training_step(colour, hw, hw, directions=["side"], global_step=1).backward()
    g_col = colour.grad
    assert torch.isfinite(d.grad).all() and torch.isfinite(f.grad).all() and float(f.grad.abs().max()) > 0
    grid = vo.Grid(dens.numpy(), feat.numpy(), [(-1.5, 1.5)] * 3, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    rdn, rfn = vo.render_bwd(grid, make_render_cfg(64, NEAR, FAR, white_bkgd=True), ro.cpu().numpy(), rd.cpu().numpy(),
                             g_col.cpu().numpy())
    assert rel_l2(d.grad.cpu().numpy(), rdn) < 1e-4 and rel_l2(f.grad.cpu().numpy(), rfn) < 1e-4
