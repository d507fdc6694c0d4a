"""BASELINE-size checks (160^3 SH-0 softplus grid, 400x400, S=256) on the GPU: oracle comparison on a
subset of rays + size-independent properties (mapping invariance, linearity of the backward,
forward/backward consistency by directional finite differences)."""
import numpy as np
import pytest
import torch

from helpers import rel_l2
from synth import FAR, NEAR, RADIUS, focal_for, random_grid, sphere_grid, synth_pose_angles
from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

from oracle import voxe_oracle as vo

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import gpu_helpers as gh
    from thre3d_atom.utils.imaging_utils import pose_spherical

AABB = [(-1.5, 1.5)] * 3
S = 256


def _grid(kind="random"):
    dens, feat = random_grid(160) if kind == "random" else sphere_grid(160)
    return vo.Grid(dens.numpy(), feat.numpy(), AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)


def _rays(hw, i=3, n=100):
    yaw, pitch = synth_pose_angles(i, n)
    pose = pose_spherical(yaw, pitch, RADIUS)
    return vo.cast_rays(hw, hw, focal_for(hw), pose.rotation.numpy(), pose.translation.numpy())


@pytest.mark.parametrize("kind", ["random", "sphere"])
def test_400x400_forward_subset_vs_oracle(kind):
    grid = _grid(kind)
    o, d = _rays(400)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    out = gh.hip_forward(grid, cfg, o, d, image_width=400)
    sel = np.random.default_rng(1).choice(o.shape[0], 3000, replace=False)
    ref = vo.render_fwd(grid, cfg, o[sel], d[sel])
    np.testing.assert_allclose(out["colour"][sel], ref["colour"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["depth"][sel], ref["depth"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["acc"][sel], ref["acc"], rtol=0, atol=1e-5)
    # index math on the same subset, bit exact
    pr, po = gh.hip_probe(grid, cfg, o[sel[:400]], d[sel[:400]]), vo.sample_probe(grid, cfg, o[sel[:400]], d[sel[:400]])
    np.testing.assert_array_equal(pr["idx"], po["idx"])
    np.testing.assert_array_equal(pr["inside"], po["inside"])
    frac_inside = po["inside"].mean()
    assert 0.3 < frac_inside < 0.8  # ~57 % of samples are inside the AABB


def test_100x100_backward_vs_oracle():
    grid = _grid("sphere")
    o, d = _rays(100, i=7)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    gc = np.random.default_rng(43).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, image_width=100)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < 1e-4 and rel_l2(gf, rf) < 1e-4


@pytest.mark.parametrize("hw,cam", [(64, 5), (120, 9), (160, 5), (200, 2), (266, 9), (266, 40), (320, 7)])
def test_mid_resolution_backward_vs_oracle(hw, cam):
    """image sizes between 64 and 320 pixels walk through every tile decomposition of the LDS-window backward (whole
    tile, 32-lane halves, 16-lane quadrants; as consecutive passes or as sibling blocks): gradients vs the oracle"""
    grid = _grid("sphere" if cam % 2 else "random")
    o, d = _rays(hw, i=cam)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    gc = np.random.default_rng(hw + cam).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, image_width=hw)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < 1e-4 and rel_l2(gf, rf) < 1e-4


def test_400x400_backward_properties():
    grid = _grid("random")
    o, d = _rays(400, i=11)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    rng = np.random.default_rng(5)
    g1 = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    g2 = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    d1, f1 = gh.hip_backward(grid, cfg, o, d, g1, image_width=400)
    d2, f2 = gh.hip_backward(grid, cfg, o, d, g2, image_width=400)
    d12, f12 = gh.hip_backward(grid, cfg, o, d, g1 + g2, image_width=400)
    # linearity of the backward in the upstream gradient
    assert rel_l2(d1 + d2, d12) < 1e-4 and rel_l2(f1 + f2, f12) < 1e-4
    # thread->ray mapping invariance (linear order vs 2-D tiles)
    # (the scatter kernel and the LDS-window kernel evaluate the suffix sum `total - prefix` in float32 in
    #  two different orders: density gradients agree to a few 1e-5, both ~4e-5 from the double oracle)
    dl, fl = gh.hip_backward(grid, cfg, o, d, g1)
    assert rel_l2(dl, d1) < 1e-4 and rel_l2(fl, f1) < 1e-5
    # forward/backward consistency: <grad, v> vs central finite difference of sum(colour * g1)
    v = rng.standard_normal(grid.features.shape).astype(np.float32)
    eps = 1e-2

    def loss(feat):
        gg = vo.Grid(grid.densities, feat, AABB, grid.density_scale, grid.density_pre_act, grid.density_post_act)
        c = gh.hip_forward(gg, cfg, o, d, image_width=400)["colour"]
        return float(np.sum(c.astype(np.float64) * g1))

    fd = (loss(grid.features + eps * v) - loss(grid.features - eps * v)) / (2 * eps)
    an = float(np.sum(f1.astype(np.float64) * v))
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0), (fd, an)
