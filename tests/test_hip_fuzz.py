"""Randomised parity sweep (fixed seeds): grid shapes, world boxes, activations, channel kinds, sample counts, sampling
modes, cameras (inside / outside / grazing the box), image-ordered and unordered rays -- HIP vs the oracle:
sample indices and masks bit for bit, renders to 5e-6, gradients to 1e-4 rel-L2 (or absolute when they vanish)."""
import os

import numpy as np
import pytest
import torch

from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

from oracle import voxe_oracle as vo

# the small images of this module are meant for the LDS-window (tile) backward: every render call asks for it through
# VoxeRenderCfg::dispatch (tile_min_rays = -1); the shipped thresholds are exercised by the other GPU modules
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("tile_always")]

if torch.cuda.is_available():
    import gpu_helpers as gh


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(1, 40, 3))
    if seed % 7 == 0:
        dims = (int(rng.integers(24, 64)),) * 3
    attn = seed % 5 == 3
    F = 1 if attn else 3
    dens = rng.uniform(-1, 1, dims + (1,)).astype(np.float32)
    feat = rng.uniform(-2, 2, dims + (F,)).astype(np.float32)
    ext = rng.uniform(0.5, 3.0, 3)
    centre = rng.uniform(-0.3, 0.3, 3)
    aabb = [(float(c - e / 2), float(c + e / 2)) for c, e in zip(centre, ext)]
    pre, post, scale = [(abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, 100.0 / 3.0), (abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, 3.0),
                        (abi.ACT_IDENTITY, abi.ACT_RELU, 20.0), (abi.ACT_ABS, abi.ACT_IDENTITY, 1.5),
                        (abi.ACT_IDENTITY, abi.ACT_IDENTITY, 0.7)][seed % 5 if not attn else 1]
    if post == abi.ACT_IDENTITY and pre == abi.ACT_IDENTITY:
        dens = np.abs(dens)  # a raw field must stay non-negative to be a density
    grid = vo.Grid(dens, feat, aabb, scale, pre, post, abi.FEAT_ATTN if attn else abi.FEAT_SH)
    # camera: on a sphere around the box, sometimes inside it, looking roughly at the centre
    h, w = int(rng.integers(3, 40)), int(rng.integers(3, 40))
    radius = rng.uniform(0.2, 4.5)
    eye = centre + radius * _unit(rng.standard_normal(3))
    fwd = _unit(centre + 0.2 * rng.standard_normal(3) - eye)
    up = _unit(np.cross(np.cross(fwd, rng.standard_normal(3)), fwd))
    right = np.cross(fwd, up)
    rot = np.stack([right, up, -fwd], axis=1).astype(np.float32)   # camera looks down -z
    focal = float(rng.uniform(0.4, 2.5) * w)
    o, d = vo.cast_rays(h, w, focal, rot, eye.astype(np.float32))
    S = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 97, 160, 257]))
    near = float(rng.uniform(0.01, 0.5))
    far = float(near + rng.uniform(0.5, 7.0))
    mode = seed % 4
    kw = dict(white_bkgd=bool(rng.integers(0, 2)), linear_disparity=(mode == 1 and not attn), aabb_clip=(mode == 2))
    jitter = rng.uniform(0, 1, (h * w, S)).astype(np.float32) if mode == 3 else None
    if jitter is not None:
        kw["perturb"] = True
    cfg = make_render_cfg(S, near, far, **kw)
    return grid, cfg, o, d, jitter, (h, w), rng


def _close(name, got, ref, far=1.0):
    """1e-4 relative in L2, plus an absolute floor: the density gradient is a difference of O(upstream) terms
    (T dL/dw - suffix / (1 - alpha)) evaluated in float32, so when it nearly cancels (a couple of samples, ReLU field)
    only the absolute error is meaningful.  With an upstream gradient on the DEPTH the cancelling terms carry the sample
    distances (dL/dw_k = ... + g_depth z_k, z up to `far`), so the floor scales with max(1, far) like the forward's depth
    tolerance above (r04 soak, seed 14642 of 16 000: S = 2, ReLU field, far = 5.3: 6.4e-5 on |ref| = 1.7e-3; the plain
    one-atomic-per-corner scatter kernel of r01 has 3.9e-5 there)"""
    err = float(np.linalg.norm(np.asarray(got, np.float64) - np.asarray(ref, np.float64)))
    assert err <= 1e-4 * float(np.linalg.norm(ref)) + 5e-5 * max(1.0, float(far)), (name, err, float(np.linalg.norm(ref)))


def _unit(v):
    return v / max(np.linalg.norm(v), 1e-12)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VOXE_FUZZ_SEEDS", "40"))))
def test_random_configuration(seed):
    grid, cfg, o, d, jitter, (h, w), rng = _case(seed)
    ordered = seed % 3 != 2
    width = w if ordered else 0
    if not ordered:                                        # unordered rays: the scatter backward paths
        perm = rng.permutation(h * w)
        o, d = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
        jitter = None if jitter is None else np.ascontiguousarray(jitter[perm])
    # --- index math: bit for bit
    ref_p = vo.sample_probe(grid, cfg, o, d, jitter)
    got_p = gh.hip_probe(grid, cfg, o, d, jitter, image_width=width)
    assert np.array_equal(got_p["inside"].astype(bool), ref_p["inside"])
    assert np.array_equal(got_p["z"], ref_p["z"])
    m = ref_p["inside"]
    assert np.array_equal(got_p["idx"][m], ref_p["idx"][m])
    # --- forward
    ref = vo.render_fwd(grid, cfg, o, d, jitter)
    got = gh.hip_forward(grid, cfg, o, d, jitter, image_width=width)
    for k in ("colour", "depth", "acc"):
        # depth = sum z_k w_k: errors of the weights (1e-6) are amplified by the sample distances (up to `far`)
        scale = max(1.0, float(np.abs(ref[k]).max())) * (max(1.0, float(cfg.far)) if k == "depth" else 1.0)
        # alpha = 1 - exp(-x) in float32 is quantised to 6e-8 whatever x is, in the reference as in the oracle and the
        # kernels (three different exp's): with S samples of a faint medium the sums may drift apart by ~S * 3e-8
        atol = 5e-6 * scale + 3e-8 * cfg.num_samples * (max(1.0, float(cfg.far)) if k == "depth" else 1.0)
        np.testing.assert_allclose(got[k], ref[k].reshape(got[k].shape), rtol=0, atol=atol, err_msg=k)
    # --- backward (colour + depth + accumulated weight upstream gradients)
    cout = grid.cout
    gc = rng.standard_normal((h * w, cout)).astype(np.float32)
    gdep = (0.2 * rng.standard_normal(h * w)).astype(np.float32)
    gacc = (0.2 * rng.standard_normal(h * w)).astype(np.float32)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, d_acc=gacc, jitter=jitter)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, g_acc=gacc, jitter=jitter, image_width=width)
    for name, got_g, ref_g in (("densities", gd, rd), ("features", gf, rf)):
        # (the floor scales with `far` only where a depth gradient is applied: that is what makes the cancelling terms large)
        _close(name, got_g, ref_g, far=cfg.far if (name == "densities" and np.any(gdep != 0.0)) else 1.0)


@pytest.mark.parametrize("seed", range(max(12, int(os.environ.get("VOXE_FUZZ_SEEDS", "40")) // 4)))
def test_random_configuration_sh_degrees(seed):
    """same sweep with view-dependent colour (SH degree 1..3, 3 * (deg + 1)^2 feature channels), full and diffuse"""
    grid, cfg, o, d, jitter, (h, w), rng = _case(100 + seed)
    if grid.feature_kind == abi.FEAT_ATTN:
        grid.feature_kind = abi.FEAT_SH
    deg = 1 + seed % 3
    dims = grid.densities.shape[:3]
    grid.features = rng.uniform(-1, 1, dims + (3 * (deg + 1) ** 2,)).astype(np.float32)
    cfg.sh_degree = deg
    cfg.render_diffuse = int(seed % 4 == 0 or seed % 7 == 3)   # (diffuse with both ray orders)
    width = w if seed % 2 else 0
    ref = vo.render_fwd(grid, cfg, o, d, jitter)
    got = gh.hip_forward(grid, cfg, o, d, jitter, image_width=width)
    np.testing.assert_allclose(got["colour"], ref["colour"], rtol=0, atol=5e-6)
    gc = rng.standard_normal((h * w, 3)).astype(np.float32)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, jitter=jitter)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, jitter=jitter, image_width=width)
    for name, got_g, ref_g in (("densities", gd, rd), ("features", gf, rf)):
        _close(name, got_g, ref_g)


@pytest.mark.parametrize("seed", range(10))
def test_random_grid_passes(seed):
    """whole-grid passes (density correlation, total variation, Adam, trilinear up-sampling) and the point query on
    random shapes, vs the oracle"""
    from helpers import rel_l2
    from voxe_hip import ops

    rng = np.random.default_rng(500 + seed)
    dims = tuple(int(v) for v in rng.integers(1, 30, 3))
    C = int(rng.integers(1, 5))
    # density-correlation loss (needs non-degenerate variance) and TV (needs every axis to have a difference)
    a = rng.uniform(-1, 1, dims + (1,)).astype(np.float32)
    b = (a + 0.3 * rng.standard_normal(a.shape)).astype(np.float32)
    if a.size > 2:
        ta = gh.t(a, True)
        loss = ops.density_correlation_loss(ta, gh.t(b))
        loss.backward()
        ref_loss, ref_grad = vo.dcl_fwd_bwd(a, b)
        assert abs(float(loss.detach()) - ref_loss) < 5e-6
        assert rel_l2(gh.n(ta.grad), ref_grad) < 2e-5
    if min(dims) >= 2:
        grid = rng.uniform(-1, 1, dims + (C,)).astype(np.float32)
        tg = gh.t(grid, True)
        tv = ops.tv_loss_on_grid(tg)
        tv.backward()
        ref_tv, ref_tvg = vo.tv_fwd_bwd(grid)
        assert abs(float(tv.detach()) - ref_tv) < 5e-6 and rel_l2(gh.n(tg.grad), ref_tvg) < 1e-6
    # Adam, odd lengths and steps
    n = int(rng.integers(1, 5000))
    p = rng.standard_normal(n).astype(np.float32)
    m, v = np.zeros_like(p), np.zeros_like(p)
    tp, tm, tvv = gh.t(p.copy()), gh.t(m.copy()), gh.t(v.copy())
    for step in range(1, 4):
        grad = (rng.standard_normal(n) * 10.0 ** float(rng.integers(-6, 3))).astype(np.float32)
        lr = float(rng.uniform(1e-4, 0.1))
        ops.adam_step_(tp, gh.t(grad), tm, tvv, step, lr=lr)
        vo.adam_step(p, grad, m, v, lr, 0.9, 0.999, 1e-8, step)
        np.testing.assert_allclose(gh.n(tp), p, rtol=2e-6, atol=1e-7)
    # trilinear up-sampling to an arbitrary size
    src = rng.uniform(-1, 1, dims + (C,)).astype(np.float32)
    out_size = tuple(int(v) for v in rng.integers(1, 45, 3))
    np.testing.assert_allclose(gh.n(ops.upsample_trilinear(gh.t(src), out_size)), vo.upsample_trilinear(src, out_size),
                               rtol=0, atol=1e-6)
    # point query, points inside / on the faces / outside the box
    grid, _, _, _, _, _, _ = _case(300 + seed)
    lo = np.array([r[0] for r in grid.aabb]); hi = np.array([r[1] for r in grid.aabb])
    pts = (lo + (hi - lo) * rng.uniform(-0.2, 1.2, (777, 3))).astype(np.float32)
    pts[:8] = np.array([[lo[0], lo[1], lo[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], lo[2]], (lo + hi) / 2,
                        [lo[0], (lo[1] + hi[1]) / 2, hi[2]], lo - 1, hi + 1, [hi[0], lo[1], lo[2]]], np.float32)
    d, f = gh.t(grid.densities, True), gh.t(grid.features, True)
    out = ops.query_points(gh.spec_of(grid), d, f, gh.t(pts))
    ref = vo.query_fwd(grid, pts)
    np.testing.assert_allclose(gh.n(out), ref, rtol=3e-6, atol=3e-6)
    g_out = rng.standard_normal(ref.shape).astype(np.float32)
    (out * gh.t(g_out)).sum().backward()
    rd, rf = vo.query_bwd(grid, pts, g_out)
    _close("query densities", gh.n(d.grad), rd)
    _close("query features", gh.n(f.grad), rf)


@pytest.mark.parametrize("n,count", [(1, 1), (5, 5), (1000, 999), (8 * 400 * 400, 32768), ((1 << 21) + 3, 100000)])
def test_random_subset_bit_exact(n, count):
    from voxe_hip import ops

    got = ops.random_subset(n, count, gh.DEV, rng=(77, 5)).cpu().numpy()
    assert np.array_equal(got, vo.random_subset(n, count, 77, 5))
    assert len(np.unique(got)) == count and got.min() >= 0 and got.max() < n
