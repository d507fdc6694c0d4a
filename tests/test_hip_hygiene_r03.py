"""r03 parity hygiene (VERDICT r02 "what's weak" 1 + ADVICE r02):

* gradient accuracy PER MAGNITUDE BAND at the bench size (160^3, 400x400, S = 256) for the three backward routes -- a global
  rel-L2 (helpers.rel_l2) cannot see a wrong deposit confined to faint voxels;
* ownership of every oracle-inside sample by exactly one segment of the space-binned route, read from the tables the
  PRODUCTION forward left in the workspace (voxe_region_debug_layout);
* the disparity chain rule kernel against autograd through the reference's three tensor ops;
* the scope of FusedGridAdam's deferred-gradient mode and its per-parameter step counters.
"""
import gc

import numpy as np
import pytest
import torch

from helpers import band_errors, rel_l2
from synth import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles
from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

from oracle import voxe_oracle as vo

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import gpu_helpers as gh
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import ops

AABB = [(-1.5, 1.5)] * 3
S = 256


def _grid(side=160):
    dens, feat = random_grid(side)
    return vo.Grid(dens.numpy(), feat.numpy(), AABB, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)


def _rays(hw, i):
    yaw, pitch = synth_pose_angles(i, 100)
    pose = pose_spherical(yaw, pitch, RADIUS)
    return vo.cast_rays(hw, hw, focal_for(hw), pose.rotation.numpy(), pose.translation.numpy())


# Measured on MI355X (r03), median relative error per band of |oracle gradient| / max, bands (1e-3, 1] / (1e-6, 1e-3] /
# (1e-9, 1e-6]:  features 2e-7 / 4e-7 / 7e-7 on every route;  densities 1e-5 / 8e-6 / 9e-6 (tile, scatter).  Before the
# forward saved SUFFIX sums at the depth-segment boundaries (r03: render_fwd_combine_kernel) the densities read 3.6e-5 /
# 4.6e-3 / 0.23 -- the suffix sum_{j>k} dL/dw_j w_j taken as (whole ray) - (prefix) cancels for samples deep inside a
# dense medium -- while the REFERENCE's float32 autograd (reverse cumsum) holds 4e-7 in every band against the same oracle
# (tests/golden/render_sh0.npz, 16^3).  What remains is the cancellation inside one 32-sample segment.  The bars sit ~5x
# above the measured medians.
BAND_BARS = {"features": {(1e-3, 1.0): 2e-6, (1e-6, 1e-3): 3e-6, (1e-9, 1e-6): 5e-6},
             # measured (profiles/r03_band_probe.txt, re-measured r04): 1e-5 / 8e-6 / 9e-6 on the tile and scatter routes, 3e-6 / 2e-6 /
             # 2e-6 on the space-binned route; the bars sit at 2x the worst of them (r03: 5x).  What it would take to go lower,
             # measured: profiles/r04_suffix_accuracy.txt (everything in double: 1.7e-6 / 1.2e-6 / 1.5e-6 at +5.8 % of the step)
             "densities": {(1e-3, 1.0): 2e-5, (1e-6, 1e-3): 2e-5, (1e-9, 1e-6): 2e-5}}


@pytest.mark.parametrize("route", ["tile", "region", "scatter"])
def test_gradient_error_per_magnitude_band_at_bench_size(route, disp):
    """median relative error of the voxel gradients per decade band of |oracle gradient| / max: a deposit that goes wrong
    only for faint voxels (lost corners, a quantised window) fails here and passes a global rel-L2"""
    grid = _grid()
    o, d = _rays(400, 3)
    over = {}
    if route == "tile":
        over = dict(image_width=400)
    else:
        perm = np.random.default_rng(11).permutation(o.shape[0])
        o, d = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
        disp.set(region_min_rays=16384 if route == "region" else -1)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=42, rng_offset=7)
    gc_ = np.random.default_rng(43).standard_normal((o.shape[0], 3)).astype(np.float32)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc_, rng=(42, 7), **over)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc_)
    assert rel_l2(gd, rd) < 1e-4 and rel_l2(gf, rf) < 1e-4
    for name, got, ref in (("densities", gd, rd), ("features", gf, rf)):
        bands = band_errors(got, ref, list(BAND_BARS[name]))
        assert len(bands) == 3, (route, name, bands)          # >= 3 decades populated
        for (lo, hi), (count, median, p99) in bands.items():
            assert count > 1000, (route, name, lo, hi, count)
            assert median < BAND_BARS[name][(lo, hi)], (route, name, (lo, hi), count, median, p99)
        # nothing deposited where the oracle has no gradient at all (rays never reached those voxels)
        stray = np.abs(got[ref == 0.0])
        assert stray.size == 0 or float(stray.max()) <= 1e-12 * float(np.abs(ref).max()), (route, name, float(stray.max()))


# VoxeDispatch::precise_grad (r05, lean tile kernels): the density bars of the reference-accuracy mode -- <= 3e-6 median per band
# (measured 2.0e-6 / 1.4e-6 / 1.6e-6, profiles/r05_band_probe.txt; default mode 9.6e-6 / 8.0e-6 / 9.3e-6), 99th percentile
# per band <= 1e-4 / 3e-4 / 1e-3 (measured 4.7e-5 / 1.1e-4 / 5.6e-4; default 2.9e-4 / 1.0e-3 / 5.2e-3).  What is left comes from
# the float states at the segment boundaries; the reference's own float32 autograd holds 4e-7 against the same oracle.
PRECISE_DENSITY_MEDIAN = 3e-6
PRECISE_DENSITY_P99 = {(1e-3, 1.0): 1e-4, (1e-6, 1e-3): 3e-4, (1e-9, 1e-6): 1e-3}


def test_precise_grad_density_bands_at_bench_size(disp):
    """VoxeDispatch::precise_grad = 1 on the image-ordered SH-0 render: density gradients per magnitude band at the tighter
    bars, forward outputs bit-identical to the default mode, features unchanged"""
    grid = _grid()
    o, d = _rays(400, 3)
    cfg = make_render_cfg(S, NEAR, FAR, perturb=True, white_bkgd=True, seed=42, rng_offset=7)
    gc_ = np.random.default_rng(43).standard_normal((o.shape[0], 3)).astype(np.float32)
    f0 = gh.hip_forward(grid, cfg, o, d, rng=(42, 7), image_width=400)
    disp.set(precise_grad=1)
    f1 = gh.hip_forward(grid, cfg, o, d, rng=(42, 7), image_width=400)
    for key in ("colour", "depth", "acc"):
        np.testing.assert_array_equal(f0[key], f1[key], err_msg=key)
    gd, gf = gh.hip_backward(grid, cfg, o, d, gc_, rng=(42, 7), image_width=400)
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc_)
    assert rel_l2(gd, rd) < 1e-5 and rel_l2(gf, rf) < 1e-5
    bands = band_errors(gd, rd, list(BAND_BARS["densities"]))
    assert len(bands) == 3, bands
    for (lo, hi), (count, median, p99) in bands.items():
        assert count > 1000 and median < PRECISE_DENSITY_MEDIAN and p99 < PRECISE_DENSITY_P99[(lo, hi)], ((lo, hi), count, median, p99)
    for (lo, hi), (count, median, p99) in band_errors(gf, rf, list(BAND_BARS["features"])).items():
        assert median < BAND_BARS["features"][(lo, hi)], ("features", (lo, hi), median)


@pytest.mark.parametrize("ranks", ["lds_ranks", "global_ranks"])
@pytest.mark.parametrize("case", ["random_batch", "sparse_image", "generic_bin"])
def test_region_route_every_inside_sample_owned_by_exactly_one_segment(case, ranks, disp):
    """the segment tables the production forward writes: (a) every sample the ORACLE's probe calls inside belongs to
    exactly one segment of its ray; (b) a segment's samples all start in the segment's region (generic bin excepted);
    (c) the counting sort is a permutation: `sorted` lists every used slot once, inside its region's range -- with the
    segments ranked per block in LDS (r05, shipped) and with one returning global atomic per segment (grids above ~200^3)"""
    disp.set(region_min_rays=1, region_lds_ranks=0 if ranks == "lds_ranks" else -1)
    grid = _grid(96)
    Sn = 128
    if case == "random_batch":
        o, d = _rays(300, 9)
        sel = np.random.default_rng(2).permutation(o.shape[0])[:6000]
        o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
        over = {}
    elif case == "sparse_image":
        disp.set(region_image_ratio=-1.0)
        o, d = _rays(56, 21)
        over = dict(image_width=56)
    else:   # few samples over a fine grid: lanes run out of slots, the rest goes to the generic bin
        Sn = 48                 # (> 20000 rays: 32-sample depth segments, i.e. up to 32 one-sample segments per lane of 16 slots)
        o, d = _rays(200, 13)
        sel = np.random.default_rng(4).permutation(o.shape[0])[:21000]
        o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
        over = {}
    cfg = make_render_cfg(Sn, NEAR, FAR, perturb=True, white_bkgd=True, seed=3, rng_offset=5)
    spec, params = gh.spec_of(grid), gh.params_of(cfg, **over)
    dens, feat = gh.t(grid.densities), gh.t(grid.features)
    ws = ops.Workspace()
    R = o.shape[0]
    outs = [torch.empty((R, n), dtype=torch.float32, device=gh.DEV) for n in (3, 1, 1, 1)]
    ops.render_fwd_into(spec, params, dens, feat, gh.t(o), gh.t(d), None, *outs, ws, (3, 5))
    torch.cuda.synchronize()
    tab = ops.region_debug_tables(spec, params, dens, feat, R, ws)
    probe = vo.sample_probe(grid, cfg, o, d)
    inside = probe["inside"].astype(bool)                # [R, S] oracle's strict inside test
    idx = probe["idx"]                                   # [R, S, 3] low-corner voxel index
    per_lane = tab["slots_per_lane"]
    lane_n = tab["lane_n"].cpu().numpy()
    nlanes = lane_n.shape[0]
    nseg = nlanes // R
    # slot j of lane L is slot j * nlanes + L (j-major)
    seg = tab["slot_seg"].cpu().numpy().view(np.uint32).reshape(per_lane, nlanes, 2).transpose(1, 0, 2)
    reg = tab["slot_region"].cpu().numpy().view(np.uint32).reshape(per_lane, nlanes).T
    bx, by, bz = tab["region_cells"]
    X, Y, Z = grid.densities.shape[:3]
    nry, nrz = ((max(Y - 1, 1)) + by - 1) // by, ((max(Z - 1, 1)) + bz - 1) // bz
    nreg = tab["nreg"]
    owned = np.zeros(inside.shape, dtype=np.int32)
    used_slots, generic_samples = 0, 0
    for lane in np.nonzero(lane_n)[0]:
        r = lane % R          # lane = depth segment * R + ray
        for j in range(int(lane_n[lane])):
            ray, kk = int(seg[lane, j, 0]), int(seg[lane, j, 1])
            k0, k1 = kk & 0xFFFF, kk >> 16
            assert ray == r and k0 <= k1 < Sn
            region = int(reg[lane, j]) & 0x00FFFFFF
            ks = np.arange(k0, k1 + 1)
            ks = ks[inside[r, k0:k1 + 1]]
            owned[r, ks] += 1
            used_slots += 1
            if region == nreg:
                generic_samples += ks.size
                continue
            assert k1 - k0 + 1 <= tab["chunk"]
            # cells clamped like make_cell(): low corner in [0, N - 2]
            cx = np.clip(idx[r, ks, 0], 0, max(X - 2, 0)) // bx
            cy = np.clip(idx[r, ks, 1], 0, max(Y - 2, 0)) // by
            cz = np.clip(idx[r, ks, 2], 0, max(Z - 2, 0)) // bz
            assert np.all((cx * nry + cy) * nrz + cz == region), (case, lane, j)
    assert np.array_equal(owned, inside.astype(np.int32)), (case, int((owned != inside).sum()))
    if case == "generic_bin":
        assert generic_samples > 0
    # the counting sort: every used slot appears exactly once, between its region's start offsets
    start = tab["start"].cpu().numpy().view(np.uint32)
    srt = tab["sorted"].cpu().numpy().view(np.uint32)[:used_slots]
    assert int(start[-1]) == used_slots
    slots = srt[:, 2].astype(np.int64)
    assert np.unique(slots).size == used_slots
    flat_reg = tab["slot_region"].cpu().numpy().view(np.uint32).reshape(-1)
    ncls = tab["len_classes"]
    pos = np.arange(used_slots)
    r_of = (flat_reg[slots] & 0x00FFFFFF).astype(np.int64)
    assert np.all(pos >= start[r_of * ncls]) and np.all(pos < start[(r_of + 1) * ncls])


def test_disparity_chain_rule_kernel_matches_autograd():
    g = torch.Generator().manual_seed(5)
    R = 5000
    depth = (torch.rand(R, 1, generator=g) * 4.0).to(gh.DEV)
    acc = torch.rand(R, 1, generator=g).to(gh.DEV)
    acc[::17] = 0.0            # rays that miss: disparity NaN in the reference, no gradient
    depth[::17] = 0.0
    depth[5::23] = 1e-14       # quotient below the clamp: no gradient
    g_disp = torch.randn(R, 1, generator=g).to(gh.DEV)
    g_dep0 = torch.randn(R, 1, generator=g).to(gh.DEV)
    d, a = depth.clone().requires_grad_(True), acc.clone().requires_grad_(True)
    disp = 1.0 / torch.maximum(torch.full_like(d, 1e-10), d / a)       # accumulate.py:85-88
    gd_ref, ga_ref = torch.autograd.grad(disp, (d, a), g_disp)
    gd_ref = torch.nan_to_num(gd_ref, nan=0.0, posinf=0.0, neginf=0.0) + g_dep0
    ga_ref = torch.nan_to_num(ga_ref, nan=0.0, posinf=0.0, neginf=0.0)
    gd, ga = ops.disparity_bwd(depth, acc, g_disp, g_dep0, None)
    # (same real-number formula, another rounding order than autograd's maximum / div / reciprocal chain)
    torch.testing.assert_close(gd, gd_ref, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(ga, ga_ref, rtol=1e-4, atol=1e-6)


def _small_model():
    from thre3d_atom.modules.volumetric_model import VolumetricModel
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig, render_sh_voxel_grid
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize
    from thre3d_atom.utils.imaging_utils import CameraBounds

    side = 16
    dens, feat = random_grid(side)
    vg = VoxelGrid(dens, feat, VoxelSize(3.0 / side, 3.0 / side, 3.0 / side), density_preactivation=torch.nn.Identity(),
                   density_postactivation=torch.nn.Softplus(), expected_density_scale=3.0, tunable=True)
    cfg = SHVoxGridRenderConfig(32, CameraBounds(NEAR, FAR), white_bkgd=True, render_num_samples_per_ray=32)
    return VolumetricModel(vg, render_sh_voxel_grid, cfg, device=gh.DEV)


def _render_loss(model, seed=7):
    from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays
    from thre3d_atom.utils.imaging_utils import CameraIntrinsics

    torch.manual_seed(seed)      # (the render draws its jitter stream from torch's generator: same draws for every call)
    yaw, pitch = synth_pose_angles(3, 100)
    rays = flatten_rays(cast_rays(CameraIntrinsics(24, 24, focal_for(24)), pose_spherical(yaw, pitch, RADIUS), device=gh.DEV))
    return model.render_rays(rays).colour.square().mean()


def test_fused_grid_adam_mode_is_scoped():
    """the deferred-gradient mode must not outlive its optimiser"""
    from thre3d_atom.modules.optim import FusedGridAdam

    model = _small_model()
    grid = model.thre3d_repr
    ws = grid.voxe_workspace("sh")
    # a loop that raises inside `with`: the mode is gone afterwards, ordinary gradients flow again
    with pytest.raises(ZeroDivisionError):
        with FusedGridAdam(grid, lr=1e-2) as opt:
            _render_loss(model).backward()
            assert grid.densities.grad is None and ws.deferred.dirty
            1 / 0
    assert ws.deferred is None
    _render_loss(model).backward()
    assert grid.densities.grad is not None and float(grid.densities.grad.abs().sum()) > 0
    ref = grid.densities.grad.clone()
    grid.densities.grad = grid.features.grad = None
    # a second optimiser on a grid whose mode is on raises instead of taking over the accumulated gradient
    opt = FusedGridAdam(grid, lr=1e-2)
    with pytest.raises(RuntimeError):
        FusedGridAdam(grid, lr=1e-2)
    # dropped without detach(): the finalizer leaves the mode; the unconsumed gradient does not leak into later steps
    _render_loss(model).backward()
    assert ws.deferred.dirty
    del opt
    gc.collect()
    assert ws.deferred is None
    _render_loss(model).backward()
    torch.testing.assert_close(grid.densities.grad, ref, rtol=1e-4, atol=1e-9)
    grid.densities.grad = grid.features.grad = None
    with FusedGridAdam(grid, lr=1e-2) as opt2:
        before = grid.densities.detach().clone()
        _render_loss(model).backward()
        opt2.step()
        moved = (grid.densities.detach() - before).abs().max()
        assert 0 < float(moved) <= 1.0001e-2          # ONE Adam step of one gradient (not two summed, not none)


def test_fused_grid_adam_counts_steps_per_parameter():
    """after a step in which only the densities had a gradient, the two Adam counters differ; the fused
    step applies each tensor's own bias correction, like torch.optim.Adam"""
    from thre3d_atom.modules.optim import FusedGridAdam

    torch.manual_seed(0)
    fused_m, ref_m = _small_model(), _small_model()
    fg, rg = fused_m.thre3d_repr, ref_m.thre3d_repr
    ref_opt = torch.optim.Adam([{"params": [rg.densities, rg.features], "lr": 2e-2}], betas=(0.9, 0.999))
    with FusedGridAdam(fg, lr=2e-2) as opt:
        for it in range(4):
            for m, o in ((fused_m, opt), (ref_m, ref_opt)):
                o.zero_grad()
                if it == 1:      # regulariser-only step on the densities (no render): features keep their counter
                    (m.thre3d_repr.densities ** 2).sum().backward()
                else:
                    _render_loss(m, seed=100 + it).backward()
                o.step()
            assert opt.state[fg.features]["step"] == ref_opt.state[rg.features]["step"]
        assert opt.state[fg.densities]["step"] == 4 and opt.state[fg.features]["step"] == 3
    torch.testing.assert_close(fg.densities.detach(), rg.densities.detach(), rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(fg.features.detach(), rg.features.detach(), rtol=1e-4, atol=2e-6)


def test_render_route_and_clock_probe(disp):
    """voxe_render_route names the kernels a render resolves to (it is part of the ray-state key: forward and backward must
    agree), also when a tuning switch flips between the two calls; voxe_clock_probe returns a plausible shader clock"""
    import ctypes as C

    from voxe_hip.runtime import lib

    grid = _grid(32)
    spec = gh.spec_of(grid)
    dens, feat = gh.t(grid.densities), gh.t(grid.features)

    def route(R, **over):
        cfg = make_render_cfg(64, NEAR, FAR, white_bkgd=True)
        g, c = ops._descs(spec, gh.params_of(cfg, **over), dens, feat, 0, 0, False)
        return lib().voxe_render_route(C.byref(g), C.byref(c), R)

    assert route(96 * 96, image_width=96) == abi.ROUTE_TILE
    assert route(20000) == abi.ROUTE_REGION
    assert route(3000) == abi.ROUTE_PACKED_SCATTER
    assert route(96 * 96, image_width=96, deterministic=True) == abi.ROUTE_DETERMINISTIC
    assert route(0) == abi.ROUTE_NONE
    disp.set(region_min_rays=-1)
    assert route(20000) == abi.ROUTE_PACKED_SCATTER
    # the switch flipped between a forward and its backward: the states of the other route are NOT taken for valid -- the
    # backward re-marches and the gradients still equal the oracle's
    disp.set(region_min_rays=0)
    o, d = _rays(150, 5)
    sel = np.random.default_rng(1).permutation(o.shape[0])[:17000]
    o, d = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
    cfg = make_render_cfg(64, NEAR, FAR, white_bkgd=True)
    dt, ft = gh.t(grid.densities, True), gh.t(grid.features, True)
    colour = ops.render(spec, gh.params_of(cfg), dt, ft, gh.t(o), gh.t(d))[0]          # forward: space-binned route
    gc_ = np.random.default_rng(2).standard_normal((o.shape[0], 3)).astype(np.float32)
    disp.set(region_min_rays=-1)                                  # backward: line-dense scatter
    (colour * gh.t(gc_)).sum().backward()
    torch.cuda.synchronize()
    rd, rf = vo.render_bwd(grid, cfg, o, d, gc_)
    assert rel_l2(gh.n(dt.grad), rd) < 1e-4 and rel_l2(gh.n(ft.grad), rf) < 1e-4
    hz = ops.clock_probe(gh.DEV)
    assert 1.0e9 < hz < 3.0e9, hz
