"""r06 parity cases (all through the C ABI, against the CPU oracle).

* `VoxeDispatch::precise_grad = 1` where the forward does NOT write the double segment sums -- a render of
  a single depth segment (S <= 32, or S <= 16 / 8 for small launches), `fwd_segments_per_thread > 1`, caller-supplied jitter --
  used to hand the PREC backward uninitialised workspace memory.  The library now derives "the sums exist" from ONE predicate
  for the forward, the re-march and the backward (`precise_sums_apply`, csrc/voxe_api.hip).
  Math: accumulate.py:49-84 (weights, suffix sums), the backward of SURVEY 8(a16)."""
import numpy as np
import pytest
import torch

from helpers import rel_l2
from synth import FAR, NEAR, RADIUS, focal_for, synth_pose_angles
from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

from oracle import voxe_oracle as vo

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import gpu_helpers as gh
    from thre3d_atom.utils.imaging_utils import pose_spherical

AABB = [(-1.5, 1.5)] * 3


def _grid(side, seed=5, post=abi.ACT_SOFTPLUS, scale=8.0):
    rng = np.random.default_rng(seed)
    dens = rng.uniform(-1, 1, (side,) * 3 + (1,)).astype(np.float32)
    feat = rng.uniform(-1, 1, (side,) * 3 + (3,)).astype(np.float32)
    return vo.Grid(dens, feat, AABB, scale, abi.ACT_IDENTITY, post)


def _rays(hw, i):
    yaw, pitch = synth_pose_angles(i, 100)
    pose = pose_spherical(yaw, pitch, RADIUS)
    return vo.cast_rays(hw, hw, focal_for(hw), pose.rotation.numpy(), pose.translation.numpy())


@pytest.mark.parametrize("case", ["S16", "S32", "S8_small", "S96_fseg2", "S96_caller_jitter", "S96_lean_off", "S96"])
def test_precise_grad_where_the_forward_keeps_no_double_sums(case, disp):
    S = {"S16": 16, "S32": 32, "S8_small": 8}.get(case, 96)
    hw = 56 if case == "S8_small" else 150        # 3 136 rays: 16-sample segments; 22 500 rays: 32-sample segments (seg_len_for)
    grid = _grid(48)
    o, d = _rays(hw, 12)
    over = dict(tile_min_rays=-1, precise_grad=1)
    if case == "S96_fseg2":
        over["fwd_segments_per_thread"] = 2
    if case == "S96_lean_off":
        over["tile_lean"] = -1
    disp.set(**over)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=7, rng_offset=1)
    jit = None
    if case == "S96_caller_jitter":
        jit = np.random.default_rng(2).uniform(0, 1, (o.shape[0], S)).astype(np.float32)
    r = np.random.default_rng(9)
    gc = r.standard_normal((o.shape[0], 3)).astype(np.float32)
    gdep = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32)
    # poison what a previous tenant of the caching allocator left behind: the PREC backward must not read sums nobody wrote
    junk = torch.full((64 << 20,), float("nan"), device="cuda")
    del junk
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, jitter=jit, rng=(7, 1), image_width=hw)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, jitter=jit)
    assert np.isfinite(gd).all() and np.isfinite(gf).all()
    assert rel_l2(gd, rd) < 1e-4 and rel_l2(gf, rf) < 1e-4, (case, rel_l2(gd, rd), rel_l2(gf, rf))


# ---- a17 closed: every regulariser of the edit's CLI as HIP (sds_trainer.py:494-505 l2_mode / l1_mode, :526-534 feature correlation) ----
def test_regulariser_modes_vs_the_reference_goldens():
    """the stand-alone entry points (voxe_density_diff_fwd_bwd, voxe_feature_correlation_fwd_bwd) through autograd against the
    reference's own functions (tests/golden/reg_modes.npz): values 2e-6 relative, gradients 1e-5 rel-L2, zeros at ties"""
    from conftest import load_golden
    from voxe_hip import ops

    g = load_golden("reg_modes.npz")
    for t in ("a", "b"):
        for mode in ("l2", "l1"):
            sds = gh.t(g[f"dens_{t}_sds"], True)
            loss = ops.density_diff_loss(sds, gh.t(g[f"dens_{t}_reg"]), l2_mode=mode == "l2")
            (loss * 3.0).backward()
            ref = float(g[f"dens_{t}_{mode}_loss"])
            assert abs(float(loss) - ref) < 2e-6 * max(1.0, abs(ref))
            got = gh.n(sds.grad) / 3.0
            assert rel_l2(got, g[f"dens_{t}_{mode}_grad"]) < 1e-5
            assert np.array_equal(got == 0, g[f"dens_{t}_{mode}_grad"] == 0)
    for t in ("a", "b", "sh1", "attn"):
        sds = gh.t(g[f"feat_{t}_sds"], True)
        loss = ops.feature_correlation_loss(sds, gh.t(g[f"feat_{t}_reg"]))
        (loss * 0.5).backward()
        ref = float(g[f"feat_{t}_loss"])
        assert abs(float(loss) - ref) < 2e-6 * max(1.0, abs(ref))
        assert rel_l2(gh.n(sds.grad) / 0.5, g[f"feat_{t}_grad"]) < 1e-5


@pytest.mark.parametrize("case", ["l2", "l1", "featcorr", "l1+featcorr_slab", "correlation+featcorr", "attn_featcorr"])
def test_regulariser_kinds_inside_the_fused_grid_step_vs_the_oracle(case):
    """VoxeGridRegularisers (ABI v11): the l2 / l1 density terms and the feature-correlation term evaluated INSIDE
    voxe_grid_adam_step.  One step from the zero Adam state with a known workspace gradient: exp_avg / (1 - beta1) is the
    gradient the step saw = workspace gradient + weight x the ORACLE's regulariser gradient (pinned to the reference above);
    the logged loss values against the oracle's; 41 x 9 x 36 voxels: the coalesced kernel + its tail; a slab; an attention grid
    (2-channel texels: the per-voxel kernel)."""
    from voxe_hip import ops

    dev = gh.DEV
    rng = np.random.default_rng(4)
    attn = case == "attn_featcorr"
    dims = (41, 9, 36) if "slab" not in case else (12, 16, 20)
    F = 1 if attn else 3
    C = F + 1
    dens0 = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat0 = rng.uniform(-2, 2, (*dims, F)).astype(np.float32)
    ref_d = (dens0 + 0.3 * rng.standard_normal(dens0.shape)).astype(np.float32)
    ref_d.reshape(-1)[::5] = dens0.reshape(-1)[::5]           # ties: sign(0) = 0
    ref_f = (feat0 + 0.5 * rng.standard_normal(feat0.shape)).astype(np.float32)
    spec = ops.GridSpec(aabb=tuple((-1.5, 1.5) for _ in range(3)), density_scale=3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_RELU, feature_kind=abi.FEAT_ATTN if attn else abi.FEAT_SH)
    d, f = gh.t(dens0), gh.t(feat0)
    ws = ops.Workspace()
    # a workspace with a (linear-layout) gradient: render nothing, write the gradient region by hand
    params = ops.RenderParams(num_samples=8, near=NEAR, far=FAR)
    o = torch.zeros((1, 3), device=dev); dd = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    outs = [torch.empty((1, n), device=dev) for n in ((1 if attn else 3), 1, 1, 1)]
    ops.render_fwd_into(spec, params, d, f, o, dd, None, *outs, ws, (0, 0))
    region = ops.workspace_grad_view(spec, d, f, ws)
    region.zero_()
    gws = rng.standard_normal((int(np.prod(dims)), C)).astype(np.float32)
    region[: gws.size] = gh.t(gws.reshape(-1))
    w_d, w_f = 0.37, 0.011
    kind = {"l2": abi.DREG_L2, "l1": abi.DREG_L1}.get(case.split("+")[0], abi.DREG_CORRELATION)
    use_d = case not in ("featcorr", "attn_featcorr")
    use_f = "featcorr" in case
    x_range = (3, 9) if "slab" in case else None
    x0, x1 = x_range if x_range else (0, dims[0])
    dl, fl = torch.zeros((), device=dev), torch.zeros((), device=dev)
    st_d = (torch.zeros_like(d), torch.zeros_like(d))
    st_f = (torch.zeros_like(f), torch.zeros_like(f))
    ops.grid_adam_step_(spec, d, f, abi.GRAD_LINEAR, ws, 1, 0.01, state_densities=st_d, state_features=st_f, x_range=x_range,
                        dcl_reference=gh.t(ref_d) if use_d else None, dcl_weight=w_d, dcl_loss=dl if use_d else None,
                        density_kind=kind, feat_reference=gh.t(ref_f) if use_f else None, feat_weight=w_f,
                        feat_loss=fl if use_f else None)
    torch.cuda.synchronize()
    # what the step must have seen (identity pre-activation: d packed / d density = scale)
    g_d = gws[:, F].reshape(*dims, 1) * np.float32(3.0)
    g_f = gws[:, :F].reshape(*dims, F).copy()
    n_all = dens0.size
    if use_d:
        if kind == abi.DREG_CORRELATION:
            loss_d, rg = vo.dcl_fwd_bwd(dens0, ref_d, w_d)
        else:
            loss_d, rg = vo.density_diff_fwd_bwd(dens0, ref_d, kind, w_d)
            if x_range:   # the slab's share of the mean (voxe.h)
                loss_d, _ = vo.density_diff_fwd_bwd(dens0[x0:x1], ref_d[x0:x1], kind)
                loss_d *= dens0[x0:x1].size / n_all
        g_d = g_d + rg
        assert abs(float(dl) - loss_d) < 2e-6 * max(1.0, abs(loss_d)), (float(dl), loss_d)
    if use_f:
        loss_f, rg = vo.feature_correlation_fwd_bwd(feat0[x0:x1], ref_f[x0:x1], w_f)
        g_f[x0:x1] += rg
        assert abs(float(fl) - loss_f) < 2e-6 * max(1.0, abs(loss_f)), (float(fl), loss_f)
    got_d, got_f = gh.n(st_d[0]) / np.float32(0.1), gh.n(st_f[0]) / np.float32(0.1)
    assert rel_l2(got_d[x0:x1], g_d[x0:x1]) < 2e-6, rel_l2(got_d[x0:x1], g_d[x0:x1])
    assert rel_l2(got_f[x0:x1], g_f[x0:x1]) < 2e-6, rel_l2(got_f[x0:x1], g_f[x0:x1])
    if x_range:     # nothing outside the slab moved
        assert float(st_d[0][:x0].abs().max()) == 0.0 and float(st_d[0][x1:].abs().max()) == 0.0
        assert np.array_equal(gh.n(d)[:x0], dens0[:x0]) and np.array_equal(gh.n(f)[x1:], feat0[x1:])
    # parameters: the oracle's Adam on the gradient the step saw
    p_ref = dens0.reshape(-1).copy()
    sl = slice(x0 * dims[1] * dims[2], x1 * dims[1] * dims[2])
    ps = p_ref[sl].copy()
    vo.adam_step(ps, np.ascontiguousarray(got_d.reshape(-1)[sl]), np.zeros_like(ps), np.zeros_like(ps), 0.01, 0.9, 0.999, 1e-8, 1)
    assert np.abs(gh.n(d).reshape(-1)[sl] - ps).max() < 1e-6


def test_wide_texels_refuse_the_in_step_feature_term_and_the_trainer_falls_back():
    """SH degree >= 1 grids: voxe_grid_adam_step answers VOXE_ERR_UNSUPPORTED to feat_reference (voxe.h); FusedGridAdam refuses
    set_feature_correlation there, and the stand-alone HIP pass serves the autograd route"""
    from voxe_hip import ops
    from voxe_hip.runtime import VoxeError

    dev = gh.DEV
    dims, F = (8, 8, 8), 12
    d = torch.rand((*dims, 1), device=dev); f = torch.rand((*dims, F), device=dev)
    spec = ops.GridSpec(aabb=tuple((-1.5, 1.5) for _ in range(3)), density_scale=3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS, feature_kind=abi.FEAT_SH)
    ws = ops.Workspace()
    params = ops.RenderParams(num_samples=8, near=NEAR, far=FAR, sh_degree=1)
    o = torch.zeros((1, 3), device=dev); dd = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    outs = [torch.empty((1, n), device=dev) for n in (3, 1, 1, 1)]
    ops.render_fwd_into(spec, params, d, f, o, dd, None, *outs, ws, (0, 0))
    ops.workspace_grad_view(spec, d, f, ws).zero_()
    with pytest.raises(VoxeError):
        ops.grid_adam_step_(spec, d, f, abi.GRAD_LINEAR, ws, 1, 0.01, state_densities=(torch.zeros_like(d), torch.zeros_like(d)),
                            state_features=(torch.zeros_like(f), torch.zeros_like(f)), feat_reference=torch.rand_like(f), feat_weight=0.1)
    fr = f.clone().requires_grad_(True)
    loss = ops.feature_correlation_loss(fr, torch.rand_like(f))
    loss.backward()
    ref_loss, ref_grad = vo.feature_correlation_fwd_bwd(gh.n(fr), gh.n(fr) * 0 + gh.n(fr))   # (shape check only: D = 0)
    assert ref_loss == 0.0 and fr.grad.shape == f.shape and torch.isfinite(fr.grad).all()


# ---- r06: per-lane sample indices in the lean backward (skewed march of oblique tiles, sample phases of split tiles) ----------------
@pytest.mark.parametrize("case", ["oblique_400px_like", "coarse_image_quadrants", "coarse_image_halves", "z_march_pairs", "term_eps",
                                  "depth_and_acc_gradients", "odd_image_side"])
def test_skewed_and_phased_march_vs_oracle_and_the_one_sample_march(case, disp):
    """VoxeDispatch::tile_phases (ABI v11): lanes of a wave at DIFFERENT samples of their rays -- shifted by the layers a ray is
    ahead of the pass's reference ray (oblique views), and, in the parts of a tile that does not fit the window, 2 / 4 consecutive
    samples of one ray in 2 / 4 lanes (transmittance and running sum exchanged between them).  Against the oracle (accumulate.py:
    49-84, the backward of SURVEY 8(a16)) and against the r05 march (tile_phases = -1): same gradients to float summation order.
    (The shipped library is built WITHOUT the phased marches -- profiles/r06_phases_kl8.txt -- so both settings run the one-sample
    march there; a library built with -DVOXE_T4_PHASES_KL8=1 -DVOXE_T4_PHASES_KL10=1 and VOXE_HIP_LIB runs them against each other.)"""
    side, hw, cam, S, kl = {"oblique_400px_like": (64, 160, 12, 128, 8), "coarse_image_quadrants": (96, 56, 3, 96, 0),
                            "coarse_image_halves": (72, 72, 58, 96, 8), "z_march_pairs": (64, 160, 0, 128, 8),
                            "term_eps": (64, 120, 88, 96, 8), "depth_and_acc_gradients": (64, 136, 38, 112, 8),
                            "odd_image_side": (80, 75, 12, 96, 0)}[case]
    grid = _grid(side, seed=21, scale=3.0 * side / 32)
    o, d = _rays(hw, cam)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=9, rng_offset=4,
                          term_eps=1e-3 if case == "term_eps" else 0.0)
    r = np.random.default_rng(5)
    gc = r.standard_normal((o.shape[0], 3)).astype(np.float32)
    gdep = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32) if case == "depth_and_acc_gradients" else None
    gacc = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32) if case == "depth_and_acc_gradients" else None
    disp.set(tile_min_rays=-1, tile_kl=kl)
    new = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, g_acc=gacc, rng=(9, 4), image_width=hw)
    disp.set(tile_min_rays=-1, tile_kl=kl, tile_phases=-1)
    old = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, g_acc=gacc, rng=(9, 4), image_width=hw)
    # (term_eps: a phased pass truncates a ray behind the wave's 2 / 4 samples, the one-sample march behind the sample itself: the
    #  gradients of up to 3 samples at T < 1e-3 differ)
    tol = 2e-3 if case == "term_eps" else 5e-6
    assert rel_l2(new[0], old[0]) < tol and rel_l2(new[1], old[1]) < tol, (rel_l2(new[0], old[0]), rel_l2(new[1], old[1]))
    if case != "term_eps":      # (the truncation is not in the reference: the two marches are compared with each other only)
        rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, d_acc=gacc)
        assert rel_l2(new[0], rd) < 1e-4 and rel_l2(new[1], rf) < 1e-4, (rel_l2(new[0], rd), rel_l2(new[1], rf))


# ---- the forward of view-dependent grids through an LDS window of whole texels -----------------------------
def _sh_grid(side, deg, seed=3, dims=None):
    rng = np.random.default_rng(seed)
    shape = tuple(dims) if dims else (side,) * 3
    nf = 3 * (deg + 1) ** 2
    dens = rng.uniform(-1, 1, shape + (1,)).astype(np.float32)
    feat = (0.4 * rng.uniform(-1, 1, shape + (nf,))).astype(np.float32)
    return vo.Grid(dens, feat, AABB, 8.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)


@pytest.mark.parametrize("deg,case", [(1, "plain"), (2, "plain"), (3, "plain"), (2, "x_march"), (1, "z_march"), (2, "z_march"),
                                      (3, "z_march"), (2, "oblique"), (1, "multi_view"), (2, "clip_jitter_tensor"), (3, "lindisp"),
                                      (2, "tiny_grid"), (1, "coarse_pixels"), (2, "no_jitter_S100")])
def test_wide_window_forward_is_bit_identical_to_the_ray_ordered_forward(deg, case, disp):
    """render_fwd_tilew_kernel against render_fwd_seg_kernel (spherical_harmonics.py:87-116, process.py:45-67): the same contraction
    with the corner texels read from LDS -- every output bit equal, the oracle within the forward tolerance, and the gradients of the
    two-phase backward (it reads the forward's per-sample (rad, v)) equal too"""
    disp.set(region_min_rays=-1, tile_min_rays=-1)
    rng = np.random.default_rng(100 * deg + len(case))
    kw, over, jit, S = dict(white_bkgd=True, sh_degree=deg), {}, None, 96
    side, hw, cam = 64, 168, 3        # ~0.38 voxel per pixel, like 400x400 on 160^3
    if case == "tiny_grid":
        grid, hw, cam = _sh_grid(0, deg, dims=(5, 6, 7)), 40, 5
    else:
        grid = _sh_grid(side, deg)
    if case == "x_march":
        cam = 77
    if case == "z_march":
        cam = 12
        disp.set(fwd_zdom=-1.0)
    if case == "oblique":
        cam = 40
    if case == "coarse_pixels":
        hw = 72
    o, d = _rays(hw, cam)
    if case == "multi_view":
        o2, d2 = _rays(hw, cam + 30)
        o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
        over["image_height"] = hw
    if case in ("plain", "x_march", "z_march", "oblique", "multi_view", "tiny_grid", "coarse_pixels"):
        kw.update(perturb=True, seed=4, rng_offset=2)
    if case == "clip_jitter_tensor":
        kw.update(perturb=True, aabb_clip=True)
        jit = rng.random((o.shape[0], S)).astype(np.float32)
    if case == "lindisp":
        kw.update(linear_disparity=True)
    if case == "no_jitter_S100":
        S = 100
    cfg = make_render_cfg(S, NEAR, FAR, **kw)
    disp.set(fwd_window=-1)
    a = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
    disp.set(fwd_window=0)
    b = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
    for key in ("colour", "depth", "acc"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    np.testing.assert_array_equal(np.isnan(a["disparity"]), np.isnan(b["disparity"]))
    ref = vo.render_fwd(grid, cfg, o, d, jitter=jit)
    np.testing.assert_allclose(b["colour"], ref["colour"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(b["acc"], ref["acc"], rtol=0, atol=1e-5)
    # the window really served the render: in the test-aid mode everything it did not serve is NaN
    if case in ("plain", "x_march", "z_march", "multi_view"):
        disp.set(fwd_window=2)
        s = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
        hit = a["acc"] > 0
        served = hit & ~np.isnan(s["acc"])
        assert served.sum() > 0.6 * hit.sum(), (int(served.sum()), int(hit.sum()))
        np.testing.assert_array_equal(a["colour"][served], s["colour"][served])
    if case in ("plain", "oblique", "z_march", "clip_jitter_tensor"):
        gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
        disp.set(fwd_window=-1)
        g0 = gh.hip_backward(grid, cfg, o, d, gc, jitter=jit, rng=(4, 2), image_width=hw, **over)
        disp.set(fwd_window=0)
        g1 = gh.hip_backward(grid, cfg, o, d, gc, jitter=jit, rng=(4, 2), image_width=hw, **over)
        # (float atomics: the two runs differ by summation order only)
        assert rel_l2(g1[0], g0[0]) < 1e-5 and rel_l2(g1[1], g0[1]) < 1e-5
        rd, rf = vo.render_bwd(grid, cfg, o, d, gc, jitter=jit)
        assert rel_l2(g1[0], rd) < 1e-4 and rel_l2(g1[1], rf) < 1e-4
