"""r05: an OPAQUE scene -- rays whose transmittance underflows to exactly 0.0f behind a dense surface.  Everything behind the
surface contributes exact zeros in the oracle (weights alpha * 0, suffix sums of zeros: accumulate.py:63-84); the kernels march
those samples like any others (an exact early termination was built and measured: profiles/r05_early_exit.txt, not shipped) and
must leave nothing but rounding-sized values there."""
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


def _opaque_grid(side, post_act):
    """a dense ball (sigma * delta ~ 60 per sample: T underflows after two samples) in a faint random medium"""
    rng = np.random.default_rng(11)
    x = np.linspace(-1.5, 1.5, side, dtype=np.float32)
    rr = np.sqrt(x[:, None, None] ** 2 + x[None, :, None] ** 2 + x[None, None, :] ** 2)
    dens = np.where(rr < 0.8, 60.0, 0.02).astype(np.float32)[..., None] * (1.0 + 0.1 * rng.standard_normal((side,) * 3 + (1,)).astype(np.float32))
    feat = rng.uniform(-1, 1, (side,) * 3 + (3,)).astype(np.float32)
    return vo.Grid(dens, feat, AABB, 100.0, abi.ACT_IDENTITY, post_act)


def _rays(hw, i):
    yaw, pitch = synth_pose_angles(i, 100)
    pose = pose_spherical(yaw, pitch, RADIUS)
    return vo.cast_rays(hw, hw, focal_for(hw), pose.rotation.numpy(), pose.translation.numpy())


@pytest.mark.parametrize("route", ["tile_lean", "tile_lean_precise", "region_random_batch"])
@pytest.mark.parametrize("post_act", [abi.ACT_SOFTPLUS, abi.ACT_RELU])
def test_opaque_scene_matches_the_oracle(route, post_act, disp):
    grid = _opaque_grid(96, post_act)
    S = 192
    hw = 120
    o, d = _rays(hw, 12)
    over = dict(image_width=hw)
    if route == "region_random_batch":
        disp.set(region_min_rays=1)
        sel = np.random.default_rng(3).permutation(o.shape[0])[:6000]
        o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
        over = {}
    else:
        disp.set(tile_min_rays=-1, precise_grad=1 if route == "tile_lean_precise" else 0)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=5, rng_offset=3)
    rng = (5, 3)
    out, ref = gh.hip_forward(grid, cfg, o, d, rng=rng, **over), vo.render_fwd(grid, cfg, o, d)
    # the scene IS opaque: most rays through the ball end at T == 0 exactly
    assert (ref["acc"] == 1.0).mean() > 0.08
    np.testing.assert_allclose(out["colour"], ref["colour"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["acc"], ref["acc"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["depth"], ref["depth"], rtol=1e-5, atol=1e-5)
    r = np.random.default_rng(9)
    gc = r.standard_normal((o.shape[0], 3)).astype(np.float32)
    gdep = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32)
    gacc = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, g_acc=gacc, rng=rng, **over)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, d_acc=gacc)
    assert rel_l2(gd, rd) < 1e-4 and rel_l2(gf, rf) < 1e-4, (rel_l2(gd, rd), rel_l2(gf, rf))
    # nothing behind the surface receives a gradient: voxels the oracle leaves at exactly zero hold rounding-sized values only
    # (float suffix sums next to the surface: 1.1e-9 of the largest gradient measured on the tile route)
    untouched = (rd == 0) & (rf == 0).all(axis=-1, keepdims=True)
    assert untouched.mean() > 0.05
    assert np.abs(gd[untouched]).max(initial=0.0) <= 1e-8 * np.abs(rd).max()


@pytest.mark.parametrize("deg,dims", [(1, (21, 19, 23)), (2, (24, 20, 16)), (3, (21, 19, 23))])
def test_lean_deposit_passes_of_view_dependent_grids_vs_oracle_and_the_general_kernel(deg, dims, disp):
    """r05: the deposit passes of the two-phase image-ordered backward run in the lean tile kernel and flush into a group-planar
    staging gradient that one pass adds into the packed gradient (voxel counts with and without a tail behind the 64-voxel
    chunks); against the oracle, against the general kernel's deposit passes (VoxeDispatch::tile_lean = -1), twice on one
    workspace (the staging planes are cleared per call).  Math: spherical_harmonics.py:87-116, accumulate.py:63-84."""
    rng = np.random.default_rng(deg)
    F = 3 * (deg + 1) ** 2
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, F)).astype(np.float32)
    grid = vo.Grid(dens, feat, AABB, 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    hw = 40
    o, d = _rays(hw, 17)
    cfg = make_render_cfg(96, NEAR, FAR, white_bkgd=True, sh_degree=deg, perturb=True, seed=8, rng_offset=2)
    gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    gdep = (0.1 * rng.standard_normal(o.shape[0])).astype(np.float32)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep)
    disp.set(tile_min_rays=-1)
    lean = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, rng=(8, 2), image_width=hw)
    again = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, rng=(8, 2), image_width=hw)
    disp.set(tile_lean=-1)
    general = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, rng=(8, 2), image_width=hw)
    for got in (lean, again, general):
        assert rel_l2(got[0], rd) < 1e-4 and rel_l2(got[1], rf) < 1e-4, (rel_l2(got[0], rd), rel_l2(got[1], rf))
    assert rel_l2(lean[1], general[1]) < 2e-6 and rel_l2(lean[0], general[0]) < 2e-6
    assert rel_l2(lean[1], again[1]) < 2e-6


@pytest.mark.parametrize("hw,cam,kl,side", [(96, 3, 8, 64), (96, 0, 8, 64), (90, 12, 10, 64), (100, 26, 0, 48), (70, 40, 8, 40)])
def test_attention_backward_with_frozen_densities(hw, cam, kl, side, disp):
    """attention grids whose densities are frozen (the refinement loop: modules/attn_grid_trainer.py:243-247; what
    voxe_attn_refine_step runs): the LDS-window backward with want_densities = 0 against the oracle -- x / y / z-march cameras,
    both window widths, image sides that are not multiples of 8, a depth gradient upstream as well -- under both values of
    VoxeDispatch::tile_lean (a lean one-channel variant of the kernel was built on this test, measured and not shipped:
    profiles/r05_attn_lean_null.txt)"""
    from voxe_hip import ops
    rng = np.random.default_rng(side + cam)
    dens = rng.uniform(-1.0, 1.0, (side,) * 3 + (1,)).astype(np.float32)
    attn = (rng.standard_normal((side,) * 3 + (1,)) - 0.5).astype(np.float32)
    grid = vo.Grid(dens, attn, AABB, 20.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_ATTN)
    o, d = _rays(hw, cam)
    cfg = make_render_cfg(96, NEAR, FAR, perturb=True, white_bkgd=True, seed=2, rng_offset=5)
    ga = rng.standard_normal((o.shape[0], 1)).astype(np.float32)
    gdep = (0.1 * rng.standard_normal(o.shape[0])).astype(np.float32)
    _, rf = vo.render_bwd(grid, cfg, o, d, ga, d_depth=gdep, want_densities=False)
    spec = gh.spec_of(grid)
    td, tf, to, tdir = gh.t(dens), gh.t(attn), gh.t(o), gh.t(d)
    got = {}
    for lean in (0, -1):
        disp.set(tile_min_rays=-1, tile_kl=kl, tile_lean=lean)
        params = gh.params_of(cfg, image_width=hw)
        outs = [torch.empty((o.shape[0], n), device="cuda") for n in (1, 1, 1, 1)]
        ws = ops.Workspace()
        ops.render_fwd_into(spec, params, td, tf, to, tdir, None, *outs, ws, (2, 5))
        d_f = torch.zeros_like(tf)
        ops.render_bwd_into(spec, params, td, tf, to, tdir, None, outs[0], outs[1], outs[2], gh.t(ga), gh.t(gdep), None, None, d_f,
                            ws, (2, 5))
        got[lean] = gh.n(d_f)
        assert rel_l2(got[lean], rf) < 1e-4, (lean, rel_l2(got[lean], rf))
    assert rel_l2(got[0], got[-1]) < 2e-6, rel_l2(got[0], got[-1])


@pytest.mark.parametrize("dims,C", [((37, 20, 24), 1), ((38, 12, 16), 3), ((5, 8, 4), 1), ((3, 4, 8), 3), ((17, 9, 7), 3)])
def test_tv_pass_odd_shapes_vs_oracle(dims, C):
    """the TV pass (tv_kernel_v4: 4 elements per vector; rows that do not vectorise -> tv_kernel) on x extents that are not
    multiples of 4, grids smaller than one block and exact ties (sign(0) = 0): loss and gradient against the oracle
    (modules/sds_trainer.py:563-567), accumulate on top of an existing gradient too.  (A variant with 4 consecutive x-planes per
    thread was built on this test and measured slower: profiles/r05_grid_passes.txt)"""
    import ctypes as C_
    from voxe_hip.runtime import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(sum(dims) + C)
    grid = rng.standard_normal((*dims, C)).astype(np.float32)
    grid[1:3, 2:5] = grid[0:1, 2:5]                      # exact ties: sign(0) = 0
    ref_loss, ref_grad = vo.tv_fwd_bwd(grid, 0.7)
    tg = gh.t(grid)
    L = lib()
    sc = torch.empty(L.voxe_tv_scratch_bytes(*dims, C), dtype=torch.uint8, device="cuda")
    loss = torch.zeros((), device="cuda")
    base = rng.standard_normal(grid.shape).astype(np.float32)
    for accumulate in (0, 1):
        d_g = gh.t(base.copy())
        check(L.voxe_tv_fwd_bwd(ptr(tg), *dims, C, 0.7, ptr(loss), ptr(d_g), accumulate, ptr(sc), sc.numel(), stream_ptr(tg.device)), "tv")
        want = ref_grad + base if accumulate else ref_grad
        assert rel_l2(gh.n(d_g), want) < 1e-6
        assert abs(float(loss) - ref_loss) < 2e-6 * max(1.0, abs(ref_loss))


@pytest.mark.parametrize("src_dims,dst_dims,C", [((20, 24, 28), (40, 48, 56), 1), ((20, 24, 28), (40, 48, 56), 3),
                                                   ((16, 16, 16), (23, 23, 24), 3), ((10, 12, 9), (21, 17, 13), 1), ((8, 8, 8), (16, 16, 16), 2)])
def test_upsample_shapes_vs_oracle(src_dims, dst_dims, C):
    """up-sampling by factors other than 2, odd extents and 1 / 2 / 3 channels, bit for bit against the oracle's restatement of
    voxels.py:409-447.  (A variant with four outputs per thread and 16-byte stores was built on this test and measured equal:
    profiles/r05_grid_passes.txt)"""
    from voxe_hip import ops
    rng = np.random.default_rng(sum(src_dims) + C)
    src = rng.standard_normal((*src_dims, C)).astype(np.float32)
    ref = vo.upsample_trilinear(src, dst_dims)
    up = ops.upsample_trilinear(gh.t(src), dst_dims)
    np.testing.assert_array_equal(gh.n(up), ref)
