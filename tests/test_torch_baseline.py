"""The PyTorch restatement timed as `gpu_baseline` by bench.py (tools/torch_baseline.py) is the same function as the
reference's render: pinned to the reference's outputs and autograd gradients recorded in tests/golden/render_sh0.npz."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import rel_l2
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import torch_baseline as tb  # noqa: E402  (a measurement aid, not part of the product package)


@pytest.mark.parametrize("tag,kind,scale,act", [("softplus_jit_", "softplus", 100.0 / 3.0, "softplus"),
                                                ("softplus_S64_w1_", "softplus", 100.0 / 3.0, "softplus"),
                                                ("relu_jit_", "relu", 100.0 / 3.0, "relu")])
def test_torch_baseline_matches_reference_outputs_and_gradients(tag, kind, scale, act):
    g = load_golden("render_sh0.npz")
    dens = torch.from_numpy(g[kind + "_densities"]).requires_grad_(True)
    feat = torch.from_numpy(g[kind + "_features"]).requires_grad_(True)
    aabb = [tuple(float(v) for v in r) for r in g[kind + "_aabb"]]
    o, d = torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"])
    jit = torch.from_numpy(g[tag + "jitter"]) if tag + "jitter" in g.files else None
    near, far = (float(v) for v in g["bounds"])
    colour, depth, acc = tb.render(dens, feat, aabb, scale, o, d, 64, near, far, jitter=jit, white_bkgd=True, post_act=act)
    np.testing.assert_allclose(colour.detach().numpy(), g[tag + "colour"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(acc.detach().numpy(), g[tag + "acc"].reshape(-1), rtol=0, atol=5e-6)
    np.testing.assert_allclose(depth.detach().numpy(), g[tag + "depth"].reshape(-1), rtol=1e-5, atol=5e-6)
    (colour * torch.from_numpy(g[tag + "g_colour"])).sum().backward()
    assert rel_l2(dens.grad.numpy(), g[tag + "grad_densities"]) < 2e-5
    assert rel_l2(feat.grad.numpy(), g[tag + "grad_features"]) < 2e-5
