"""BASELINE.json `configs` at their FULL sizes on the GPU, every one against the CPU oracle (not against another HIP
kernel): the 400x400 backward over all rays (configs[1] / headline), the reconstruction loop's 32768-random-ray batch
over 8 cameras through the unordered-ray backward (configs[1], thre3d_atom/modules/trainers.py:288-351), the 256^3 grid
with an 800x800 camera (configs[4], one rank's share) and the attention render of the refinement loop at 160^3 / 266x266
(configs[3], thre3d_atom/modules/attn_grid_trainer.py:335-378).  Tolerances: forward 1e-5 abs (colour / acc), gradients
1e-4 rel-L2 (north_star: "within a stated float tolerance"; index math is bit exact and checked on a probe)."""
import os
import zlib

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
GRAD_TOL = 1e-4
FWD_ATOL = 1e-5


def _grid(side=160, kind="random"):
    dens, feat = random_grid(side) if kind == "random" else sphere_grid(side)
    return vo.Grid(dens.numpy(), feat.numpy(), AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)


def _pose(i, n=100):
    yaw, pitch = synth_pose_angles(i, n)
    return pose_spherical(yaw, pitch, RADIUS)


def _rays(hw, i=3, n=100):
    pose = _pose(i, n)
    return vo.cast_rays(hw, hw, focal_for(hw), pose.rotation.numpy(), pose.translation.numpy())


def _check_forward(out, ref):
    np.testing.assert_allclose(out["colour"], ref["colour"], rtol=0, atol=FWD_ATOL)
    np.testing.assert_allclose(out["acc"], ref["acc"], rtol=0, atol=FWD_ATOL)
    np.testing.assert_allclose(out["depth"], ref["depth"], rtol=1e-5, atol=FWD_ATOL)


# ---- configs[1] headline: 160^3, one 400x400 camera, every ray ---------------------------------------------------------
@pytest.mark.parametrize("kind,cam,jitter", [("random", 3, True), ("sphere", 11, False), ("random", 58, False),
                                             # camera 26 looks down z with image x along world y (the tile's lanes run down the
                                             # pixel columns), camera 12 is diagonal (lanes outside the window at the far depths)
                                             ("random", 26, True), ("random", 12, True)])
def test_cfg1_400x400_forward_backward_all_rays_vs_oracle(kind, cam, jitter):
    """the bench workload itself (camera 3, in-kernel jitter on) and two more cameras: all 160 000 rays, forward
    outputs and both gradients against the oracle; upstream gradients on colour, depth and accumulated weight"""
    grid = _grid(160, kind)
    o, d = _rays(400, cam)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=jitter, white_bkgd=True, seed=42, rng_offset=7)
    rng = (42, 7)
    _check_forward(gh.hip_forward(grid, cfg, o, d, rng=rng, image_width=400), vo.render_fwd(grid, cfg, o, d))
    r = np.random.default_rng(100 + cam)
    gc = r.standard_normal((o.shape[0], 3)).astype(np.float32)
    gdep = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32)
    gacc = (0.1 * r.standard_normal(o.shape[0])).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, g_acc=gacc, rng=rng, image_width=400)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, d_acc=gacc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))
    # voxels whose oracle gradient is exactly zero (no ray near them, or a float underflow of a vanishing density
    # derivative) hold nothing but such underflow-sized values: nothing leaks out of the LDS window's flush
    untouched = (rd == 0) & (rf == 0).all(axis=-1, keepdims=True)
    assert np.abs(gd[untouched]).max(initial=0.0) <= 1e-9 * np.abs(rd).max()


# ---- configs[1] training batch: 32768 random rays over 8 of 100 cameras, specular + diffuse --------------------------
@pytest.mark.parametrize("order", ["draw", "memory"])
def test_cfg2_random_ray_batch_vs_oracle(order):
    """trainers.py:288-351: cast 8 cameras @ 400x400, collate, random subset of 32768 rays, render specular and
    `render_diffuse=True` (fresh jitter each), L1 of both against the target pixels, one backward.  The batch is an
    unordered ray list (image_width = 0), i.e. the random-batch backward of the product; `memory` = the batch sorted by
    (camera, row, column) as `sample_random_rays_and_pixels_from_cameras(memory_order=True)` hands it over."""
    grid = _grid(160, "random")
    hw, ncam, B = 400, 8, 32768
    cams = np.random.default_rng(0).choice(100, ncam, replace=False)
    poses = np.stack([np.concatenate([_pose(int(i)).rotation.numpy(), _pose(int(i)).translation.numpy()], axis=-1) for i in cams])
    subset = vo.random_subset(ncam * hw * hw, B, 42, 3)
    assert len(np.unique(subset)) == B
    if order == "memory":
        subset = np.sort(subset)
    o, d = vo.cast_rays_indexed(hw, hw, focal_for(hw), poses, subset)
    # the product's indexed ray cast == the oracle's, bit for bit (a2)
    from voxe_hip import ops
    po, pd = ops.cast_rays_indexed(hw, hw, focal_for(hw), gh.t(poses.astype(np.float32)), gh.t(subset))
    np.testing.assert_array_equal(gh.n(po), o)
    np.testing.assert_array_equal(gh.n(pd), d)
    pix = np.random.default_rng(1).random((B, 3)).astype(np.float32)
    gd_sum = gf_sum = rd_sum = rf_sum = 0.0
    for offset, diffuse in ((1, False), (2, True)):
        cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, render_diffuse=diffuse, seed=42, rng_offset=offset)
        ref = vo.render_fwd(grid, cfg, o, d)
        _check_forward(gh.hip_forward(grid, cfg, o, d, rng=(42, offset)), ref)
        # d(L1 mean)/d colour from the oracle's colours, fed to both sides (a sign taken from colours that differ by
        # 1e-7 could flip for a pixel that sits exactly on its target)
        gc = (np.sign(ref["colour"] - pix) / pix.size).astype(np.float32)
        gd, gf = gh.hip_backward(grid, cfg, o, d, gc, rng=(42, offset))
        rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
        assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (diffuse, rel_l2(gd, rd), rel_l2(gf, rf))
        gd_sum, gf_sum, rd_sum, rf_sum = gd_sum + gd, gf_sum + gf, rd_sum + rd, rf_sum + rf
    assert rel_l2(gd_sum, rd_sum) < GRAD_TOL and rel_l2(gf_sum, rf_sum) < GRAD_TOL


def test_cfg2_trainer_batch_through_the_model_api():
    """the same batch through the product's own loop body (VolumetricModel.render_rays twice + torch L1 + autograd):
    the .grad the optimiser sees equals the oracle's gradient of that loss"""
    from thre3d_atom.modules.volumetric_model import VolumetricModel
    from thre3d_atom.rendering.volumetric.render_interface import Rays
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize
    from thre3d_atom.utils.imaging_utils import CameraBounds

    G, hw, B = 160, 400, 32768
    dens, feat = random_grid(G)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    vm = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(S, CameraBounds(NEAR, FAR), white_bkgd=True),
                         device=gh.DEV)
    poses = np.stack([np.concatenate([_pose(i).rotation.numpy(), _pose(i).translation.numpy()], axis=-1) for i in range(8)])
    subset = vo.random_subset(8 * hw * hw, B, 5, 0)
    o, d = vo.cast_rays_indexed(hw, hw, focal_for(hw), poses, subset)
    pix = np.random.default_rng(2).random((B, 3)).astype(np.float32)
    rays = Rays(gh.t(o), gh.t(d))
    # un-jittered so that the oracle can replay the two renders without knowing the product's RNG offsets
    spec = vm.render_rays(rays, perturb_sampled_points=False).colour
    diff = vm.render_rays(rays, perturb_sampled_points=False, render_diffuse=True).colour
    loss = torch.nn.functional.l1_loss(spec, gh.t(pix)) + torch.nn.functional.l1_loss(diff, gh.t(pix))
    loss.backward()
    torch.cuda.synchronize()
    grid = vo.Grid(dens.numpy(), feat.numpy(), AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    rd = rf = 0.0
    ref_loss = 0.0
    for colours, diffuse in ((spec, False), (diff, True)):
        cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True, render_diffuse=diffuse)
        ref = vo.render_fwd(grid, cfg, o, d)
        np.testing.assert_allclose(gh.n(colours), ref["colour"], rtol=0, atol=FWD_ATOL)
        ref_loss += float(np.abs(ref["colour"].astype(np.float64) - pix).mean())
        # the sign pattern the product's autograd used (its own colours)
        gc = (np.sign(gh.n(colours) - pix) / pix.size).astype(np.float32)
        a, b = vo.render_bwd(grid, cfg, o, d, gc)
        rd, rf = rd + a, rf + b
    assert abs(float(loss.detach()) - ref_loss) < 1e-6
    gd = gh.n(vm.thre3d_repr._densities.grad if hasattr(vm.thre3d_repr, "_densities") else vm.thre3d_repr.densities.grad)
    gf = gh.n(vm.thre3d_repr._features.grad if hasattr(vm.thre3d_repr, "_features") else vm.thre3d_repr.features.grad)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


# ---- configs[4]: 256^3 grid, 800x800 cameras (one rank's camera of the 8-way job) ---------------------------------------
@pytest.fixture(scope="module")
def grid256():
    return _grid(256, "random")


def test_cfg5_256_grid_800x800_forward_vs_oracle(grid256):
    o, d = _rays(800, i=17, n=200)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=9, rng_offset=4)
    out = gh.hip_forward(grid256, cfg, o, d, rng=(9, 4), image_width=800)
    _check_forward(out, vo.render_fwd(grid256, cfg, o, d))          # all 640 000 rays
    sel = np.random.default_rng(3).choice(o.shape[0], 400, replace=False)
    pr, po = gh.hip_probe(grid256, cfg, o[sel], d[sel], rng=(9, 4)), vo.sample_probe(grid256, cfg, o[sel], d[sel])
    np.testing.assert_array_equal(pr["idx"], po["idx"])            # voxel indices and inside masks: bit exact
    np.testing.assert_array_equal(pr["inside"], po["inside"])
    np.testing.assert_array_equal(pr["z"], po["z"])                # sample depths incl. the in-kernel jitter stream


def test_cfg5_256_grid_800x800_backward_vs_oracle(grid256):
    o, d = _rays(800, i=101, n=200)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    gc = np.random.default_rng(6).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid256, cfg, o, d, gc, image_width=800)
    rd, rf = vo.render_bwd(grid256, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


def test_cfg5_row_bands_of_8_ranks_sum_to_the_whole_image(grid256):
    """the 8-way job shards an 800x800 camera into row bands (thre3d_atom/modules/parallel.py); the sum of the 8 band
    gradients (what the all-reduce forms) equals the oracle's whole-image gradient"""
    o, d = _rays(800, i=33, n=200)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    gc = np.random.default_rng(8).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd = gf = 0.0
    for rank in range(8):
        lo, hi = rank * 100 * 800, (rank + 1) * 100 * 800
        a, b = gh.hip_backward(grid256, cfg, o[lo:hi], d[lo:hi], gc[lo:hi], image_width=800)
        gd, gf = gd + a.astype(np.float64), gf + b.astype(np.float64)
    rd, rf = vo.render_bwd(grid256, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL


# ---- configs[3]: attention-grid render of the refinement loop, 160^3, 266x266 (800 / 3) ---------------------------------
@pytest.mark.parametrize("hw,cam", [(266, 12), (400, 40)])
def test_cfg4_attention_render_160_vs_oracle(hw, cam, disp):
    dens, _ = sphere_grid(160)
    attn = (np.random.default_rng(4).standard_normal((160, 160, 160, 1)) - 1.0).astype(np.float32)
    grid = vo.Grid(dens.numpy(), attn, AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_ATTN)
    o, d = _rays(hw, cam)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=3, rng_offset=9)
    out, ref = gh.hip_forward(grid, cfg, o, d, rng=(3, 9), image_width=hw), vo.render_fwd(grid, cfg, o, d)
    _check_forward(out, ref)
    # r05: the lean tile-ordered forward renders attention grids too (2-channel texels): bit-identical to the ray-ordered forward
    disp.set(tile_lean=-1)
    out_ro = gh.hip_forward(grid, cfg, o, d, rng=(3, 9), image_width=hw)
    disp.set(tile_lean=0)
    for key in ("colour", "depth", "acc"):
        np.testing.assert_array_equal(out[key], out_ro[key], err_msg=key)
    ga = np.random.default_rng(5).standard_normal((o.shape[0], 1)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, ga, rng=(3, 9), image_width=hw)
    rd, rf = vo.render_bwd(grid, cfg, o, d, ga)
    assert rel_l2(gf, rf) < GRAD_TOL, rel_l2(gf, rf)          # the attention grid: what the refinement loop optimises
    assert rel_l2(gd, rd) < GRAD_TOL, rel_l2(gd, rd)


# ---- the SHIPPED dispatch for small images (explicitly: whatever the surrounding fixtures asked for) -------
@pytest.mark.parametrize("hw", [64, 100])
def test_default_dispatch_small_images_vs_oracle(hw, disp):
    """with the default threshold an image below 8192 rays takes the depth-segmented scatter backward and 100x100 the
    LDS-window backward: run exactly what ships"""
    disp.set(tile_min_rays=0, region_min_rays=0, tile_kl=0)
    grid = _grid(160, "sphere")
    o, d = _rays(hw, 21)
    cfg = make_render_cfg(S, NEAR, FAR, white_bkgd=True)
    gc = np.random.default_rng(hw).standard_normal((o.shape[0], 3)).astype(np.float32)
    _check_forward(gh.hip_forward(grid, cfg, o, d, image_width=hw), vo.render_fwd(grid, cfg, o, d))
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, image_width=hw)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL


# ---- multi-view launches: K cameras of the same size in ONE image-ordered launch (VoxeRenderCfg::image_height) ----------
@pytest.mark.parametrize("hw,K,kind", [(100, 8, "sphere"), (36, 3, "random"), (200, 2, "random")])
def test_multi_view_launch_vs_oracle(hw, K, kind):
    """K cameras back to back, image_width = W, image_height = H (100 and 36 are no multiples of the 8-pixel tiles:
    tiles must not straddle two cameras): forward and gradients equal the oracle on the concatenated rays, and the
    forward equals the same rays rendered as an unordered list"""
    grid = _grid(160, kind)
    rays = [_rays(hw, 5 + 9 * i) for i in range(K)]
    o, d = np.concatenate([r[0] for r in rays]), np.concatenate([r[1] for r in rays])
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=8, rng_offset=2)
    out = gh.hip_forward(grid, cfg, o, d, rng=(8, 2), image_width=hw, image_height=hw)
    _check_forward(out, vo.render_fwd(grid, cfg, o, d))
    flat = gh.hip_forward(grid, cfg, o, d, rng=(8, 2))                       # same rays as an unordered list: the same
    # samples (the jitter stream is keyed by the ray index); the two launches may composite in different segment orders
    np.testing.assert_allclose(out["colour"], flat["colour"], rtol=0, atol=2e-6)
    gc = np.random.default_rng(K).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, rng=(8, 2), image_width=hw, image_height=hw)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


def test_multi_view_bad_shapes_are_rejected():
    from voxe_hip.runtime import VoxeError

    grid = _grid(32, "random")
    o, d = _rays(20, 1)
    cfg = make_render_cfg(16, NEAR, FAR)
    with pytest.raises(VoxeError):
        gh.hip_forward(grid, cfg, o, d, image_width=20, image_height=15)      # 400 rays are no multiple of 15 * 20
    with pytest.raises(VoxeError):
        gh.hip_forward(grid, cfg, o, d, image_width=0, image_height=20)       # a height without a width


def test_collated_cameras_render_as_one_launch():
    """host side: collate_rays of flattened whole cameras keeps the image shape, render_rays renders the multi-view batch
    in one launch and returns what the per-camera renders return"""
    from thre3d_atom.modules.volumetric_model import VolumetricModel
    from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, collate_rays, flatten_rays
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid, _render_params
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize
    from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics

    G, hw = 64, 60
    dens, feat = sphere_grid(G)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
    vm = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(64, CameraBounds(NEAR, FAR), white_bkgd=True,
                                                                        perturb_sampled_points=False), device=gh.DEV)
    intr = CameraIntrinsics(hw, hw, focal_for(hw))
    cams = [flatten_rays(cast_rays(intr, _pose(i), device=gh.DEV)) for i in (2, 30, 71)]
    batch = collate_rays(cams)
    assert batch.image_shape == (hw, hw)
    params = _render_params(vm.thre3d_repr, batch, vm.render_config, attn=False)
    assert (params.image_width, params.image_height) == (hw, hw)
    assert _render_params(vm.thre3d_repr, cams[0], vm.render_config, attn=False).image_height == 0
    with torch.no_grad():
        whole = vm.render_rays(batch).colour
        parts = torch.cat([vm.render_rays(c).colour for c in cams])
    assert torch.equal(whole, parts)


# ---- deterministic (ordered-accumulation) backward: SURVEY 8(b) `deterministic`, section 5 "race detection" -------------
@pytest.mark.parametrize("hw,kind", [(400, "random"), (100, "sphere")])
def test_deterministic_backward_is_bit_reproducible_and_matches_the_atomic_one(hw, kind):
    """VoxeRenderCfg::deterministic = 1 accumulates in 64-bit fixed point (integer adds are associative): two launches
    on the same inputs give IDENTICAL bits; the default float-atomic backward agrees with it to 1e-6 rel-L2 (a race in
    the atomic path -- a lost or doubled deposit -- would show up here), and both agree with the oracle"""
    grid = _grid(160, kind)
    o, d = _rays(hw, 14)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=6, rng_offset=1)
    gc = np.random.default_rng(9).standard_normal((o.shape[0], 3)).astype(np.float32)
    det1 = gh.hip_backward(grid, cfg, o, d, gc, rng=(6, 1), image_width=hw, deterministic=True)
    det2 = gh.hip_backward(grid, cfg, o, d, gc, rng=(6, 1), image_width=hw, deterministic=True)
    np.testing.assert_array_equal(det1[0], det2[0])
    np.testing.assert_array_equal(det1[1], det2[1])
    atomic = gh.hip_backward(grid, cfg, o, d, gc, rng=(6, 1), image_width=hw)
    assert rel_l2(atomic[0], det1[0]) < 1e-6 and rel_l2(atomic[1], det1[1]) < 1e-6, (rel_l2(atomic[0], det1[0]), rel_l2(atomic[1], det1[1]))
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(det1[0], rd) < GRAD_TOL and rel_l2(det1[1], rf) < GRAD_TOL


def test_deterministic_backward_attention_grid_and_unsupported_cases():
    from voxe_hip.runtime import VoxeError

    dens, _ = sphere_grid(64)
    attn = np.random.default_rng(4).standard_normal((64, 64, 64, 1)).astype(np.float32)
    grid = vo.Grid(dens.numpy(), attn, AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_ATTN)
    o, d = _rays(72, 3)
    cfg = make_render_cfg(96, NEAR, FAR, white_bkgd=True)
    ga = np.random.default_rng(5).standard_normal((o.shape[0], 1)).astype(np.float32)
    a = gh.hip_backward(grid, cfg, o, d, ga, image_width=72, deterministic=True)
    b = gh.hip_backward(grid, cfg, o, d, ga, image_width=72, deterministic=True)
    np.testing.assert_array_equal(a[1], b[1])
    rd, rf = vo.render_bwd(grid, cfg, o, d, ga)
    assert rel_l2(a[1], rf) < GRAD_TOL and rel_l2(a[0], rd) < GRAD_TOL
    with pytest.raises(VoxeError):      # unordered rays have no deterministic path
        gh.hip_backward(grid, cfg, o, d, ga, deterministic=True)


# ---- space-binned render (voxe_render_region.hip: forward AND backward) forced onto small cases, every variant vs the oracle --
def _region_env(disp, image_too=False):
    disp.set(region_min_rays=1)
    if image_too:
        disp.set(region_image_ratio=-1.0)


@pytest.mark.parametrize("case", ["sh0_jitter", "sh0_clip_lindisp", "attn", "diffuse_sh1", "tiny_grid", "image_ordered",
                                  "density_only", "features_only", "jitter_tensor", "sh1", "sh2", "sh1_density_only",
                                  "sh2_features_only", "sh3", "sh3_density_only", "sh0_jitter_global_ranks",
                                  "image_ordered_global_ranks"])
def test_region_backward_variants_vs_oracle(case, disp):
    if case.endswith("_global_ranks"):      # the r02 segment pass (one returning atomic per segment): grids above ~200^3 take it
        case = case[:-len("_global_ranks")]
        disp.set(region_lds_ranks=-1)
    _region_env(disp, image_too=(case == "image_ordered"))
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)   # (not hash(): salted per process)
    dims = (5, 6, 7) if case == "tiny_grid" else (40, 33, 48)
    # (sh1 / sh2: view-dependent grids -- whole texels in LDS, two-phase backward; r03.  sh3: 49-channel texels, 151.6 KB; r04)
    nfeat = 12 if case in ("diffuse_sh1", "sh1", "sh1_density_only") else (27 if case in ("sh2", "sh2_features_only") else (1 if case == "attn" else 3))
    if case.startswith("sh3"):
        nfeat = 48
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, nfeat)).astype(np.float32)
    grid = vo.Grid(dens, feat, [(-1.5, 1.5), (-1.2, 1.3), (-1.5, 1.4)], 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS,
                   abi.FEAT_ATTN if case == "attn" else abi.FEAT_SH)
    hw = 44
    o, d = _rays(hw, 7)
    if case != "image_ordered":
        perm = rng.permutation(o.shape[0])[:1500]
        o, d = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
    kw = dict(white_bkgd=True)
    jit = None
    if case in ("sh0_jitter", "attn", "image_ordered"):
        kw.update(perturb=True, seed=5, rng_offset=9)
    if case == "sh0_clip_lindisp":
        kw.update(aabb_clip=True)
    if case == "diffuse_sh1":
        kw.update(sh_degree=1, render_diffuse=True)
    if case in ("sh1", "sh1_density_only"):
        kw.update(sh_degree=1, perturb=True, seed=5, rng_offset=9)
    if case in ("sh2", "sh2_features_only"):
        kw.update(sh_degree=2)
    if case.startswith("sh3"):
        kw.update(sh_degree=3, perturb=True, seed=5, rng_offset=9)
    if case == "jitter_tensor":
        kw.update(perturb=True)
        jit = rng.random((o.shape[0], 70)).astype(np.float32)
    cfg = make_render_cfg(70, NEAR, FAR, **kw)
    cout = 1 if case == "attn" else 3
    gc = rng.standard_normal((o.shape[0], cout)).astype(np.float32)
    gdep = (0.2 * rng.standard_normal(o.shape[0])).astype(np.float32)
    over = dict(image_width=hw) if case == "image_ordered" else {}
    out, ref = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(5, 9), **over), vo.render_fwd(grid, cfg, o, d, jitter=jit)
    _check_forward(out, ref)
    np.testing.assert_array_equal(np.isnan(out["disparity"]), np.isnan(ref["disparity"]))
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, g_depth=gdep, jitter=jit, rng=(5, 9), **over)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc, d_depth=gdep, jitter=jit)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (case, rel_l2(gd, rd), rel_l2(gf, rf))
    if case in ("density_only", "features_only", "sh1_density_only", "sh2_features_only", "sh3_density_only"):
        # one tensor frozen: the other's gradient is unchanged
        from voxe_hip import ops
        dt, ft = gh.t(grid.densities, case.endswith("density_only")), gh.t(grid.features, case.endswith("features_only"))
        c, dep, _, _ = ops.render(gh.spec_of(grid), gh.params_of(cfg), dt, ft, gh.t(o), gh.t(d), None, rng=(5, 9))
        ((c * gh.t(gc)).sum() + (dep[:, 0] * gh.t(gdep)).sum()).backward()
        got, ref = (gh.n(dt.grad), rd) if case.endswith("density_only") else (gh.n(ft.grad), rf)
        assert rel_l2(got, ref) < GRAD_TOL


def test_region_backward_equals_scatter_backward_on_a_random_batch(disp):
    """same rays through the space-binned backward and through the line-dense scatter it replaces for large batches"""
    grid = _grid(96, "random")
    o, d = _rays(300, 9)
    sel = np.random.default_rng(2).permutation(o.shape[0])[:20000]
    o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
    cfg = make_render_cfg(128, NEAR, FAR, perturb=True, white_bkgd=True, seed=1, rng_offset=1)
    gc = np.random.default_rng(3).standard_normal((o.shape[0], 3)).astype(np.float32)
    disp.set(region_min_rays=-1)
    a = gh.hip_backward(grid, cfg, o, d, gc, rng=(1, 1))
    disp.set(region_min_rays=1)
    b = gh.hip_backward(grid, cfg, o, d, gc, rng=(1, 1))
    assert rel_l2(a[0], b[0]) < 5e-5 and rel_l2(a[1], b[1]) < 2e-6, (rel_l2(a[0], b[0]), rel_l2(a[1], b[1]))
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(b[0], rd) < GRAD_TOL and rel_l2(b[1], rf) < GRAD_TOL


def test_region_render_generic_bin_vs_oracle(disp):
    """few samples over a fine grid (a step of ~4 voxels): nearly every sample starts a new region, a (ray, 32-sample
    depth segment) lane runs out of its 16 segment slots and the rest of its samples go through the GENERIC bin (texels
    from global memory, global atomics, shared by many blocks) -- forward and gradients still equal the oracle"""
    _region_env(disp)
    grid = _grid(96, "random")
    o, d = _rays(200, 13)
    sel = np.random.default_rng(4).permutation(o.shape[0])[:21000]      # > 20000 rays: 32-sample depth segments
    o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
    cfg = make_render_cfg(48, NEAR, FAR, perturb=True, white_bkgd=True, seed=2, rng_offset=3)
    _check_forward(gh.hip_forward(grid, cfg, o, d, rng=(2, 3)), vo.render_fwd(grid, cfg, o, d))
    gc = np.random.default_rng(5).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, rng=(2, 3))
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


# ---- LDS-staged forward (render_fwd_tile_kernel) against the ray-ordered forward: bit-identical outputs -------------------
@pytest.mark.parametrize("case", ["400", "266_oblique", "100_sparse", "multi_view", "clip_jitter_tensor", "lindisp", "tiny_grid",
                                  "x_march", "z_march", "z_dominant_default"])
def test_lds_staged_forward_is_bit_identical_to_the_ray_ordered_forward(case, disp):
    """same interpolation arithmetic, texels from the LDS window instead of L1 / L2 (or from the global fallback for
    footprints outside the window: sparse pixels, oblique tiles, tiny grids): every output bit equal, and equal to the
    oracle within the forward tolerance"""
    disp.set(region_min_rays=-1)
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 997)
    kw, over, jit = dict(white_bkgd=True), {}, None
    if case == "tiny_grid":
        dens = rng.uniform(-1, 1, (5, 6, 7, 1)).astype(np.float32)
        feat = rng.uniform(-1, 1, (5, 6, 7, 3)).astype(np.float32)
        grid = vo.Grid(dens, feat, [(-1.5, 1.5), (-1.2, 1.3), (-1.5, 1.4)], 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
        hw, cam = 40, 5
    else:
        grid = _grid(160, "random" if case in ("400", "multi_view") else "sphere")
        hw, cam = {"400": (400, 3), "266_oblique": (266, 40), "100_sparse": (100, 7), "multi_view": (200, 9),
                   "clip_jitter_tensor": (240, 11), "lindisp": (320, 21),
                   # camera 77 looks along x, camera 12 mostly along z (ray by ray by default; VOXE_FWD_TILE_ZDOM < 0 marches z)
                   "x_march": (320, 77), "z_march": (320, 12), "z_dominant_default": (320, 12)}[case]
    if case == "z_march":
        disp.set(fwd_zdom=-1.0)
    o, d = _rays(hw, cam)
    if case == "multi_view":
        o2, d2 = _rays(hw, cam + 30)
        o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
        over["image_height"] = hw
    S_ = 64 if case == "tiny_grid" else S
    if case in ("400", "multi_view", "100_sparse", "x_march", "z_march", "z_dominant_default"):
        kw.update(perturb=True, seed=4, rng_offset=2)
    if case == "clip_jitter_tensor":
        kw.update(perturb=True, aabb_clip=True)
        jit = rng.random((o.shape[0], S_)).astype(np.float32)
    if case == "lindisp":
        kw.update(linear_disparity=True)
    cfg = make_render_cfg(S_, NEAR, FAR, **kw)
    disp.set(fwd_window=-1, tile_lean=-1)                 # the ray-ordered forward (render_fwd_seg_kernel)
    a = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
    disp.set(fwd_window=0, tile_lean=-1)                  # the LDS-window forward (render_fwd_tile_kernel)
    b = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
    disp.set(fwd_window=-1, tile_lean=0)                  # r05: the lean tile-ordered forward where it applies (gathers)
    b5 = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
    disp.set(fwd_window=0, tile_lean=0)                   # r06: ... with the corners from the LDS window (render_fwd_tile4w_kernel; shipped)
    b6 = gh.hip_forward(grid, cfg, o, d, jitter=jit, rng=(4, 2), image_width=hw, **over)
    for other in (b, b5, b6):
        for key in ("colour", "depth", "acc"):
            np.testing.assert_array_equal(a[key], other[key], err_msg=key)
        np.testing.assert_array_equal(np.isnan(a["disparity"]), np.isnan(other["disparity"]))
    _check_forward(b, vo.render_fwd(grid, cfg, o, d, jitter=jit))
    # the backward consumes the forward's depth-segment states: gradients unchanged
    if case in ("400", "266_oblique"):
        gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
        gb = gh.hip_backward(grid, cfg, o, d, gc, rng=(4, 2), image_width=hw)
        rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
        assert rel_l2(gb[0], rd) < GRAD_TOL and rel_l2(gb[1], rf) < GRAD_TOL


# ---- view-dependent field (SH degree 1: 13-channel texels) at the full grid size: both backward routes vs the oracle ----------
@pytest.mark.parametrize("order", ["image", "random"])
def test_sh1_160_full_size_vs_oracle(order):
    """f3 at BASELINE size: 160^3 grid with 12 feature channels (SH-1), S = 256; a 400x400 camera through the two-phase
    LDS-window backward (channel groups) and a 32400-ray random batch through the line-dense scatter"""
    dens, feat = random_grid(160, nfeat=12, seed=7)
    grid = vo.Grid(dens.numpy(), 0.3 * feat.numpy(), AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    o, d = _rays(400, 27)
    over = dict(image_width=400)
    if order == "random":
        sel = np.random.default_rng(1).permutation(o.shape[0])[:32400]
        o, d, over = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]), {}
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, sh_degree=1, seed=12, rng_offset=5)
    _check_forward(gh.hip_forward(grid, cfg, o, d, rng=(12, 5), **over), vo.render_fwd(grid, cfg, o, d))
    gc = np.random.default_rng(2).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, rng=(12, 5), **over)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


# ---- the reference's real-scene configuration: 200^3 grid, 416 samples per ray, linear-disparity sampling --------------------
def test_real_scene_config_200_grid_416_samples_lindisp_vs_oracle():
    """bash_scripts/real_scenes/train_default_relu_field_real.sh:22-28 (grid 200^3, num_samples_per_ray 416,
    linear_disparity_sampling): forward and gradients of a 400x400 camera and of a 32768-ray random batch vs the oracle"""
    dens, feat = random_grid(200, seed=11)
    grid = vo.Grid(dens.numpy(), feat.numpy(), AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    cfg = make_render_cfg(416, NEAR, FAR, perturb=True, linear_disparity=True, white_bkgd=True, seed=21, rng_offset=4)
    o, d = _rays(400, 33)
    _check_forward(gh.hip_forward(grid, cfg, o, d, rng=(21, 4), image_width=400), vo.render_fwd(grid, cfg, o, d))
    gc = np.random.default_rng(3).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, rng=(21, 4), image_width=400)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))
    sel = np.random.default_rng(4).permutation(o.shape[0])[:32768]
    o2, d2, g2 = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]), np.ascontiguousarray(gc[sel])
    _check_forward(gh.hip_forward(grid, cfg, o2, d2, rng=(21, 4)), vo.render_fwd(grid, cfg, o2, d2))
    gd, gf = gh.hip_backward(grid, cfg, o2, d2, g2, rng=(21, 4))
    rd, rf = vo.render_bwd(grid, cfg, o2, d2, g2)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


@pytest.mark.parametrize("deg,dims", [(1, (9, 7, 5)), (2, (13, 11, 9)), (3, (8, 8, 8)), (3, (7, 5, 3))])
def test_wide_texel_conversions_whole_chunks_tail_and_accumulate(deg, dims):
    """view-dependent grids are packed / un-packed 64 voxels at a time with 16-byte accesses (r04): grids of a whole number of
    chunks, with a tail and smaller than one chunk give the oracle's gradients; accumulate = 1 adds to what the tensors hold"""
    from voxe_hip import ops
    rng = np.random.default_rng(70 + deg)
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, 3 * (deg + 1) ** 2)).astype(np.float32)
    grid = vo.Grid(dens, feat, [(-1.5, 1.5)] * 3, 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    o, d = _rays(24, 3)
    cfg = make_render_cfg(48, NEAR, FAR, white_bkgd=True, sh_degree=deg)
    gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    out, ref = gh.hip_forward(grid, cfg, o, d), vo.render_fwd(grid, cfg, o, d)
    _check_forward(out, ref)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL
    # accumulate into tensors that already hold values
    spec, params = gh.spec_of(grid), gh.params_of(cfg)
    td, tf, to, tdir = gh.t(dens), gh.t(feat), gh.t(o), gh.t(d)
    outs = [torch.empty((o.shape[0], n), device="cuda") for n in (3, 1, 1, 1)]
    ws = ops.Workspace()
    ops.render_fwd_into(spec, params, td, tf, to, tdir, None, *outs, ws, (0, 0))
    d_d, d_f = torch.full_like(td, 0.25), torch.full_like(tf, -0.5)
    ops.render_bwd_into(spec, params, td, tf, to, tdir, None, outs[0], outs[1], outs[2], gh.t(gc), None, None, d_d, d_f, ws,
                        (0, 0), accumulate=True)
    np.testing.assert_allclose(gh.n(d_d) - 0.25, gd, rtol=0, atol=2e-6 * max(1.0, float(np.abs(gd).max())))
    np.testing.assert_allclose(gh.n(d_f) + 0.5, gf, rtol=0, atol=2e-6 * max(1.0, float(np.abs(gf).max())))


def test_region_route_beyond_the_strata_table(disp):
    """S = 600 samples per ray: more than the space-binned kernels tabulate per block (VOXE_REGION_STRATA = 512), so they take
    DepthGen's per-sample path -- same results as the oracle, jitter on"""
    _region_env(disp)
    rng = np.random.default_rng(5)
    dims = (24, 20, 28)
    grid = vo.Grid(rng.uniform(-1, 1, (*dims, 1)).astype(np.float32), rng.uniform(-1, 1, (*dims, 3)).astype(np.float32),
                   [(-1.5, 1.5), (-1.2, 1.3), (-1.5, 1.4)], 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    o, d = _rays(36, 11)
    sel = rng.permutation(o.shape[0])[:700]
    o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
    cfg = make_render_cfg(600, NEAR, FAR, white_bkgd=True, perturb=True, seed=3, rng_offset=4)
    gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    _check_forward(gh.hip_forward(grid, cfg, o, d, rng=(3, 4)), vo.render_fwd(grid, cfg, o, d))
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc, rng=(3, 4))
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    assert rel_l2(gd, rd) < GRAD_TOL and rel_l2(gf, rf) < GRAD_TOL, (rel_l2(gd, rd), rel_l2(gf, rf))


@pytest.mark.parametrize("order", ["image", "random"])
def test_sh_source_pass_reads_the_forwards_sample_values_or_regathers(order):
    """view-dependent grids (r04): the forward leaves (rad, v) of every sample for the two-phase backward's source pass -- image
    order: tile kernels; random order: space-binned route.  Three ways to the same gradient: forward + backward (values from the forward), an inference forward
    (keep_for_backward=False: nothing kept) followed by a backward (which re-marches), and a backward alone"""
    from voxe_hip import ops
    from voxe_hip.dispatch import TILE_ALWAYS, Dispatch
    rng = np.random.default_rng(21)
    dims = (40, 36, 44)
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, 27)).astype(np.float32)
    grid = vo.Grid(dens, feat, [(-1.5, 1.5)] * 3, 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    hw = 48
    o, d = _rays(hw, 17)
    cfg = make_render_cfg(96, NEAR, FAR, white_bkgd=True, sh_degree=2, perturb=True, seed=8, rng_offset=2)
    gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    if order == "random":
        perm = rng.permutation(o.shape[0])
        o, d, gc = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm]), np.ascontiguousarray(gc[perm])
        rd, rf = vo.render_bwd(grid, cfg, o, d, gc)     # (the in-kernel jitter is keyed by the ray index)
        spec, params = gh.spec_of(grid), gh.params_of(cfg, dispatch=Dispatch(region_min_rays=1))
    else:
        spec, params = gh.spec_of(grid), gh.params_of(cfg, image_width=hw, dispatch=TILE_ALWAYS)
    td, tf, to, tdir, tg = gh.t(dens), gh.t(feat), gh.t(o), gh.t(d), gh.t(gc)
    outs = [torch.empty((o.shape[0], n), device="cuda") for n in (3, 1, 1, 1)]
    got = []
    for mode in ("forward_then_backward", "inference_forward_then_backward", "backward_alone"):
        ws = ops.Workspace()
        if mode != "backward_alone":
            ops.render_fwd_into(spec, params, td, tf, to, tdir, None, *outs, ws, (8, 2),
                                keep_for_backward=(mode == "forward_then_backward"))
        else:
            ws.invalidate()
        assert (ws.state_key is not None) == (mode == "forward_then_backward")
        d_d, d_f = torch.zeros_like(td), torch.zeros_like(tf)
        ops.render_bwd_into(spec, params, td, tf, to, tdir, None, outs[0], outs[1], outs[2], tg, None, None, d_d, d_f, ws, (8, 2))
        got.append((gh.n(d_d), gh.n(d_f)))
        assert rel_l2(got[-1][0], rd) < GRAD_TOL and rel_l2(got[-1][1], rf) < GRAD_TOL, (mode, rel_l2(got[-1][0], rd), rel_l2(got[-1][1], rf))
    for a, b in zip(got[0], got[1]):
        assert rel_l2(a, b) < 2e-6


@pytest.mark.parametrize("order", ["image", "random"])
def test_a_false_ray_state_valid_claim_is_served_by_a_re_march(order):
    """`ray_state_valid = 1` is a claim of the C caller.  An inference forward (ray_state_valid = -1: per-sample values of
    a view-dependent grid NOT kept) followed by a backward that claims 1 used to read stale per-sample values; the library now
    checks the claim against its record of what the last forward left in that workspace and re-marches.  Same for a forward of
    OTHER rays in between.  The gradients equal those of an honest forward + backward."""
    from voxe_hip import ops
    from voxe_hip.dispatch import TILE_ALWAYS, Dispatch
    from voxe_hip.runtime import lib
    import ctypes as C
    rng = np.random.default_rng(5)
    dims = (36, 40, 32)
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, 12)).astype(np.float32)
    grid = vo.Grid(dens, feat, [(-1.5, 1.5)] * 3, 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    hw = 48
    o, d = _rays(hw, 31)
    cfg = make_render_cfg(96, NEAR, FAR, white_bkgd=True, sh_degree=1, perturb=True, seed=4, rng_offset=6)
    gc = rng.standard_normal((o.shape[0], 3)).astype(np.float32)
    if order == "random":
        perm = rng.permutation(o.shape[0])
        o, d, gc = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm]), np.ascontiguousarray(gc[perm])
        spec, params = gh.spec_of(grid), gh.params_of(cfg, dispatch=Dispatch(region_min_rays=1))
    else:
        spec, params = gh.spec_of(grid), gh.params_of(cfg, image_width=hw, dispatch=TILE_ALWAYS)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc)
    td, tf, to, tdir, tg = gh.t(dens), gh.t(feat), gh.t(o), gh.t(d), gh.t(gc)
    to2, tdir2 = to.flip(0).contiguous(), tdir.flip(0).contiguous()
    outs = [torch.empty((o.shape[0], n), device="cuda") for n in (3, 1, 1, 1)]
    junk = [torch.empty((o.shape[0], n), device="cuda") for n in (3, 1, 1, 1)]
    # the size a training caller allocates (a C caller's one workspace for inference and training alike)
    g_, c_ = ops._descs(spec, params, td, tf, 4, 6, False)
    nbytes = lib().voxe_workspace_bytes(C.byref(g_), C.byref(c_), o.shape[0])
    c_.ray_state_valid = -1
    assert lib().voxe_workspace_bytes(C.byref(g_), C.byref(c_), o.shape[0]) < nbytes     # inference needs none of the backward's scratch
    got = {}
    for mode in ("honest", "inference_forward", "other_rays_in_between"):
        ws = ops.Workspace()
        ws.ensure(nbytes, td.device).fill_(0xFF)        # (NaN bit patterns wherever a kernel reads what nobody wrote)
        ops.render_fwd_into(spec, params, td, tf, to, tdir, None, *outs, ws, (4, 6), keep_for_backward=(mode != "inference_forward"))
        claim = ops._state_key(ops._pack_key(spec, td, tf), params, to, tdir, None, (4, 6), ops._route(g_, c_, o.shape[0]))
        if mode == "other_rays_in_between":
            ops.render_fwd_into(spec, params, td, tf, to2, tdir2, None, *junk, ws, (4, 6))
        ws.state_key = claim                            # what a careless C caller passes: ray_state_valid = 1
        d_d, d_f = torch.zeros_like(td), torch.zeros_like(tf)
        ops.render_bwd_into(spec, params, td, tf, to, tdir, None, outs[0], outs[1], outs[2], tg, None, None, d_d, d_f, ws, (4, 6))
        got[mode] = (gh.n(d_d), gh.n(d_f))
        assert np.isfinite(got[mode][0]).all() and np.isfinite(got[mode][1]).all(), mode
        assert rel_l2(got[mode][0], rd) < GRAD_TOL and rel_l2(got[mode][1], rf) < GRAD_TOL, (mode, rel_l2(got[mode][0], rd), rel_l2(got[mode][1], rf))
    for mode in ("inference_forward", "other_rays_in_between"):
        for a, b in zip(got["honest"], got[mode]):
            assert rel_l2(a, b) < 2e-6, mode


def test_a_grid_step_between_forward_and_backward_invalidates_the_claim(tile_always):
    """ray_state_valid = 1 after voxe_grid_adam_step moved the parameters: the states in the workspace describe the OLD grid; the
    library forgets its record of the forward with the step, so the backward re-marches on the NEW grid and equals a fresh
    forward + backward there"""
    from voxe_hip import ops
    rng = np.random.default_rng(8)
    dims = (32, 32, 32)
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, 3)).astype(np.float32)
    grid = vo.Grid(dens, feat, [(-1.5, 1.5)] * 3, 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, abi.FEAT_SH)
    hw = 64
    o, d = _rays(hw, 5)
    cfg = make_render_cfg(64, NEAR, FAR, white_bkgd=True, perturb=True, seed=2, rng_offset=3)
    spec, params = gh.spec_of(grid), gh.params_of(cfg, image_width=hw)
    td, tf, to, tdir = gh.t(dens), gh.t(feat), gh.t(o), gh.t(d)
    gc = gh.t(rng.standard_normal((o.shape[0], 3)).astype(np.float32))
    outs = [torch.empty((o.shape[0], n), device="cuda") for n in (3, 1, 1, 1)]
    ws = ops.Workspace()
    ops.render_fwd_into(spec, params, td, tf, to, tdir, None, *outs, ws, (2, 3))
    claim = ws.state_key
    layout = ops.render_bwd_acc(spec, params, td, tf, to, tdir, None, outs[0], outs[1], outs[2], gc, None, None, ws, (2, 3), zero_first=True)
    st_d, st_f = (torch.zeros_like(td), torch.zeros_like(td)), (torch.zeros_like(tf), torch.zeros_like(tf))
    ops.grid_adam_step_(spec, td, tf, layout, ws, 1, 0.05, state_densities=st_d, state_features=st_f)      # the grid moves by ~lr
    # a careless caller: backward of the OLD forward's outputs with the claim that the workspace still holds its states
    fresh = [torch.empty_like(t_) for t_ in outs]
    ws2 = ops.Workspace()
    ops.render_fwd_into(spec, params, td, tf, to, tdir, None, *fresh, ws2, (2, 3))
    want_d, want_f = torch.zeros_like(td), torch.zeros_like(tf)
    ops.render_bwd_into(spec, params, td, tf, to, tdir, None, fresh[0], fresh[1], fresh[2], gc, None, None, want_d, want_f, ws2, (2, 3))
    assert claim is not None and ws.state_key is None                # (the binding itself drops its claim with the step)
    g_, c_ = ops._descs(spec, params, td, tf, 2, 3, False)
    ws.state_key = ops._state_key(ops._pack_key(spec, td, tf), params, to, tdir, None, (2, 3), ops._route(g_, c_, o.shape[0]))
    got_d, got_f = torch.zeros_like(td), torch.zeros_like(tf)
    ops.render_bwd_into(spec, params, td, tf, to, tdir, None, fresh[0], fresh[1], fresh[2], gc, None, None, got_d, got_f, ws, (2, 3))
    assert rel_l2(gh.n(got_d), gh.n(want_d)) < 2e-6 and rel_l2(gh.n(got_f), gh.n(want_f)) < 2e-6
