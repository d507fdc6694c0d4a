"""Pins the CPU oracle (oracle/voxe_cpu.c) to the reference: every check compares the oracle with
vectors produced by importing TAU-VAILab/Vox-E in the build container (tools/gen_golden.py).
CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import KINDS, cfg_from_bounds, grid_from_golden, nan_equal, psnr, rel_l2
from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

from oracle import voxe_oracle as vo

# float32 tolerances.  The oracle follows the reference's op order in float32; differences left are
# libm exp/log vs torch's vectorised versions and the order of the per-ray sums (<= a few ulp).
FWD_ATOL = 2e-6
GRAD_REL_L2 = 2e-5


def test_linspace_matches_torch():
    """t_val() in voxe_cpu.c vs torch.linspace (thre3d_atom/rendering/volumetric/sample.py:44):
    with near=0, far=1 the sample depths are exactly the linspace values."""
    grid = vo.Grid(np.zeros((2, 2, 2, 1)), np.zeros((2, 2, 2, 3)), [(-1, 1)] * 3)
    o = np.zeros((1, 3), np.float32)
    d = np.array([[0, 0, 1]], np.float32)
    for S in (1, 2, 3, 16, 17, 64, 100, 256, 416, 512, 1024):
        cfg = make_render_cfg(S, 0.0, 1.0)
        z = vo.sample_probe(grid, cfg, o, d)["z"][0]
        ref = torch.linspace(0.0, 1.0, S, dtype=torch.float32).numpy()
        assert np.array_equal(z, ref), S


def test_cast_rays():
    g = load_golden("cast_rays.npz")
    tags = sorted({k.split("_")[0] for k in g.files})
    assert len(tags) == 9
    for t in tags:
        h, w, f = g[t + "_hwf"]
        o, d = vo.cast_rays(int(h), int(w), f, g[t + "_rot"], g[t + "_trans"])
        np.testing.assert_array_equal(o.reshape(int(h), int(w), 3), g[t + "_origins"])
        # torch evaluates R @ dirs with a BLAS-ordered dot product: float tolerance
        np.testing.assert_allclose(d.reshape(int(h), int(w), 3), g[t + "_directions"], rtol=0, atol=3e-7)


def _unit_grid():
    return vo.Grid(np.zeros((2, 2, 2, 1)), np.zeros((2, 2, 2, 3)), [(-1, 1)] * 3)


@pytest.mark.parametrize("S", [2, 16, 65, 256])
@pytest.mark.parametrize("mode", ["uniform", "lindisp"])
def test_sample_depths_bit_exact(S, mode):
    g = load_golden("sampling.npz")
    o, d, b = g["rays_o"], g["rays_d"], g["bounds"]
    cfg = cfg_from_bounds(b, S, linear_disparity=(mode == "lindisp"))
    z = vo.sample_probe(_unit_grid(), cfg, o, d)["z"]
    np.testing.assert_array_equal(z, g[f"{mode}_S{S}_depths"])
    cfg = cfg_from_bounds(b, S, linear_disparity=(mode == "lindisp"), perturb=True)
    z = vo.sample_probe(_unit_grid(), cfg, o, d, jitter=g[f"{mode}_S{S}_jitter"])["z"]
    np.testing.assert_array_equal(z, g[f"{mode}_S{S}_jdepths"])


def test_ray_aabb_bounds():
    g = load_golden("sampling.npz")
    aabb = [tuple(r) for r in g["aabb"]]
    grid = vo.Grid(np.zeros((5, 6, 7, 1)), np.zeros((5, 6, 7, 3)), aabb)
    o, d, b = g["aabb_rays_o"], g["aabb_rays_d"], g["bounds"]
    cfg = cfg_from_bounds(b, 2, aabb_clip=True)
    z = vo.sample_probe(grid, cfg, o, d)["z"]  # S=2: z = (near', far') of each ray
    ref = g["aabb_bounds"]
    assert (g["aabb_hit"] == 0).sum() > 5 and (g["aabb_hit"] == 1).sum() > 5
    np.testing.assert_array_equal(z, ref)


@pytest.mark.parametrize("tag,kind", [("aniso", "softplus"), ("cube", "softplus"), ("abs", "abs"), ("relu", "relu")])
def test_voxel_forward_index_math(tag, kind):
    """VoxelGrid.forward / test_inside_volume (voxels.py:263-342): voxel indices and inside mask are
    bit-exact, interpolated values agree to float32 rounding."""
    g = load_golden("voxel_forward.npz")
    grid = grid_from_golden(g, tag + "_", kind)
    pts = g[tag + "_points"]
    # probe the points as 1-sample rays: o = p, d = (0,0,1), near = far = 0  ->  p = o + d*0
    cfg = make_render_cfg(1, 0.0, 0.0)
    d = np.tile(np.array([[0, 0, 1]], np.float32), (len(pts), 1))
    pr = vo.sample_probe(grid, cfg, pts, d)
    np.testing.assert_array_equal(pr["idx"][:, 0], g[tag + "_i0"])
    np.testing.assert_array_equal(pr["inside"][:, 0], g[tag + "_inside"])
    val = g[tag + "_values"]  # [N, 3+1] = cat(features, density) un-masked
    ins = g[tag + "_inside"]
    assert 0.2 < ins.mean() < 0.8
    np.testing.assert_allclose(pr["sigma"][ins, 0], val[ins, 3], rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(pr["sigma"][~ins, 0], 0.0)
    c0 = np.float32(0.28209479177387814)
    np.testing.assert_allclose(pr["rad"][ins, 0], c0 * val[ins, :3], rtol=2e-6, atol=2e-6)
    assert np.all(pr["rad"][~ins] == np.float32(-1e10))


def _render_cases():
    g = load_golden("render_sh0.npz")
    tags = sorted({k[: -len("colour")] for k in g.files if k.endswith("_colour") and not k.endswith("g_colour")})
    return tags


@pytest.mark.parametrize("tag", _render_cases())
def test_render_forward_and_grads(tag):
    g = load_golden("render_sh0.npz")
    kind = next(k for k in sorted(KINDS, key=len, reverse=True) if tag.startswith(k + "_"))
    grid = grid_from_golden(g, kind + "_", kind)
    rest = tag[len(kind) + 1:]
    kw = {}
    if rest.startswith("S"):
        S = int(rest.split("_")[0][1:])
        kw["white_bkgd"] = rest.split("_")[1] == "w1"
    else:
        S = 64
        kw["white_bkgd"] = True
        if rest.startswith("jit"):
            kw["perturb"] = True
        elif rest.startswith("lindisp"):
            kw["linear_disparity"] = True
        elif rest.startswith("clipjit"):
            kw["aabb_clip"] = kw["perturb"] = True
        elif rest.startswith("clip"):
            kw["aabb_clip"] = True
    cfg = cfg_from_bounds(g["bounds"], S, **kw)
    jit = g[tag + "jitter"] if tag + "jitter" in g.files else None
    o, d = g["rays_o"], g["rays_d"]
    out = vo.render_fwd(grid, cfg, o, d, jitter=jit)
    np.testing.assert_allclose(out["colour"], g[tag + "colour"], rtol=0, atol=FWD_ATOL)
    np.testing.assert_allclose(out["acc"], g[tag + "acc"], rtol=0, atol=FWD_ATOL)
    np.testing.assert_allclose(out["depth"], g[tag + "depth"], rtol=2e-6, atol=FWD_ATOL)
    nan_equal(out["disparity"], g[tag + "disparity"], rtol=1e-5, atol=1e-6)
    assert np.isnan(g[tag + "disparity"]).sum() >= 1  # the rays that miss (accumulate.py:85-88)
    if tag + "grad_densities" in g.files:
        gd, gf = vo.render_bwd(grid, cfg, o, d, g[tag + "g_colour"], jitter=jit)
        assert rel_l2(gd, g[tag + "grad_densities"]) < GRAD_REL_L2
        assert rel_l2(gf, g[tag + "grad_features"]) < GRAD_REL_L2
        gd, gf = vo.render_bwd(grid, cfg, o, d, g[tag + "g_colour"], g[tag + "g_depth"], g[tag + "g_acc"], jitter=jit)
        assert rel_l2(gd, g[tag + "grad2_densities"]) < GRAD_REL_L2
        assert rel_l2(gf, g[tag + "grad2_features"]) < GRAD_REL_L2


@pytest.mark.parametrize("kind", ["softplus", "softplus_soft"])
@pytest.mark.parametrize("white", [0, 1])
def test_render_attn(kind, white):
    g = load_golden("render_attn.npz")
    grid = grid_from_golden(g, kind + "_", kind, attn=True)
    tag = f"{kind}_w{white}_"
    cfg = cfg_from_bounds(g["bounds"], 48, white_bkgd=bool(white))
    o, d = g["rays_o"], g["rays_d"]
    out = vo.render_fwd(grid, cfg, o, d)
    np.testing.assert_allclose(out["colour"], g[tag + "colour"], rtol=0, atol=FWD_ATOL)
    np.testing.assert_allclose(out["depth"], g[tag + "depth"], rtol=2e-6, atol=FWD_ATOL)
    gd, gf = vo.render_bwd(grid, cfg, o, d, g[tag + "g_colour"])
    assert rel_l2(gd, g[tag + "grad_densities"]) < GRAD_REL_L2
    assert rel_l2(gf, g[tag + "grad_features"]) < GRAD_REL_L2  # grad w.r.t. VoxelGrid.attn


@pytest.mark.parametrize("deg", [1, 2, 3])
@pytest.mark.parametrize("mode", ["full", "diffuse"])
def test_render_sh_degrees(deg, mode):
    g = load_golden("render_shdeg.npz")
    grid = grid_from_golden(g, f"deg{deg}_", "softplus_soft")
    tag = f"deg{deg}_{mode}_"
    cfg = cfg_from_bounds(g["bounds"], 32, white_bkgd=True, sh_degree=deg, render_diffuse=(mode == "diffuse"))
    o, d = g["rays_o"], g["rays_d"]
    out = vo.render_fwd(grid, cfg, o, d)
    np.testing.assert_allclose(out["colour"], g[tag + "colour"], rtol=0, atol=3e-6)
    gd, gf = vo.render_bwd(grid, cfg, o, d, g[tag + "g_colour"])
    assert rel_l2(gd, g[tag + "grad_densities"]) < GRAD_REL_L2
    assert rel_l2(gf, g[tag + "grad_features"]) < GRAD_REL_L2


def test_regulariser_modes():
    """l2_mode / l1_mode of density_correlation_loss_fn and _feature_correlation_loss (sds_trainer.py:494-505, 526-534) against
    the reference's own functions (tests/golden/reg_modes.npz, tools/gen_golden.py g17): values <= 2e-6 relative, gradients
    <= 1e-5 rel-L2, exact zeros of the l1 gradient at ties"""
    from voxe_hip import abi

    g = load_golden("reg_modes.npz")
    for t in ("a", "b"):
        sds, reg = g[f"dens_{t}_sds"], g[f"dens_{t}_reg"]
        for mode, kind in (("l2", abi.DREG_L2), ("l1", abi.DREG_L1)):
            loss, grad = vo.density_diff_fwd_bwd(sds, reg, kind)
            assert abs(loss - float(g[f"dens_{t}_{mode}_loss"])) < 2e-6 * max(1.0, abs(loss))
            assert rel_l2(grad, g[f"dens_{t}_{mode}_grad"]) < 1e-5
            assert np.array_equal(grad == 0, g[f"dens_{t}_{mode}_grad"] == 0)
        assert float(g[f"dens_{t}_both_loss"]) == float(g[f"dens_{t}_l2_loss"])     # (l2 wins when both flags are set)
    for t in ("a", "b", "sh1", "attn"):
        loss, grad = vo.feature_correlation_fwd_bwd(g[f"feat_{t}_sds"], g[f"feat_{t}_reg"])
        assert abs(loss - float(g[f"feat_{t}_loss"])) < 2e-6 * max(1.0, abs(loss))
        assert rel_l2(grad, g[f"feat_{t}_grad"]) < 1e-5


def test_grid_ops():
    g = load_golden("grid_ops.npz")
    for t in ("a", "b"):
        loss, grad = vo.dcl_fwd_bwd(g[f"dcl_{t}_sds"], g[f"dcl_{t}_reg"])
        assert abs(loss - float(g[f"dcl_{t}_loss"])) < 2e-6
        assert rel_l2(grad, g[f"dcl_{t}_grad"]) < 1e-5
        loss, grad = vo.tv_fwd_bwd(g[f"tv_{t}_grid"])
        assert abs(loss - float(g[f"tv_{t}_loss"])) < 2e-6
        assert rel_l2(grad, g[f"tv_{t}_grad"]) < 1e-6
    p = g["adam_p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for step in range(5):
        vo.adam_step(p, np.ascontiguousarray(g["adam_grads"][step]), m, v, 0.03, 0.9, 0.999, 1e-8, step + 1)
        np.testing.assert_allclose(p, g["adam_traj"][step], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("t", ["a", "b", "c", "d"])
def test_upsample(t):
    g = load_golden("upsample.npz")
    src = np.concatenate([g[f"{t}_features"], g[f"{t}_densities"]], axis=-1)
    ref = np.concatenate([g[f"{t}_up_features"], g[f"{t}_up_densities"]], axis=-1)
    up = vo.upsample_trilinear(src, ref.shape[:3])
    np.testing.assert_allclose(up, ref, rtol=0, atol=5e-7)


def test_frames_psnr():
    """cfg1-style image check: 8 cameras @ 64x64 on the 32^3 grid, PSNR >= 40 dB (north_star)."""
    g = load_golden("frames32.npz")
    grid = grid_from_golden(g, "", "softplus")
    h, w, f = g["hwf"]
    cfg = cfg_from_bounds(g["bounds"], 128, white_bkgd=True)
    for i in range(8):
        o, d = vo.cast_rays(int(h), int(w), f, g["rot"][i], g["trans"][i])
        img = vo.render_fwd(grid, cfg, o, d)["colour"].reshape(int(h), int(w), 3)
        assert psnr(img, g["frames"][i]) > 100.0  # i.e. identical to ~1e-6
        assert np.linalg.norm(img - g["frames"][i]) / np.linalg.norm(g["frames"][i]) < 1e-5


@pytest.mark.parametrize("tag,kind", [("aniso", "softplus"), ("cube", "softplus"), ("abs", "abs"), ("relu", "relu")])
def test_point_query_forward_and_grads(tag, kind):
    """VoxelGrid.forward on all probe points (inside AND outside the AABB: the query is not masked) + autograd."""
    g = load_golden("voxel_forward.npz")
    grid = grid_from_golden(g, tag + "_", kind)
    out = vo.query_fwd(grid, g[tag + "_points"])
    np.testing.assert_allclose(out, g[tag + "_values"], rtol=2e-6, atol=2e-6)
    gd, gf = vo.query_bwd(grid, g[tag + "_points"], g[tag + "_g_out"])
    assert rel_l2(gd, g[tag + "_grad_densities"]) < 1e-5
    assert rel_l2(gf, g[tag + "_grad_features"]) < 1e-6


def test_edit_trajectory_oracle_vs_reference():
    """12 Adam steps of a tint edit run by the reference (tests/golden/edit_trajectory.npz) replayed with the oracle's
    forward, analytic backward and Adam: same losses, same edited frames (BASELINE: within 1e-3 L2)."""
    z = load_golden("edit_trajectory.npz")
    aabb = [tuple(r) for r in z["start_aabb"]]
    grid = vo.Grid(z["start_densities"].copy(), z["start_features"].copy(), aabb, 100.0 / 3.0,
                   abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    H, W, focal = int(z["hwf"][0]), int(z["hwf"][1]), float(z["hwf"][2])
    cfg = make_render_cfg(int(z["samples"]), float(z["bounds"][0]), float(z["bounds"][1]), white_bkgd=True)
    cams = [vo.cast_rays(H, W, focal, z["rot"][i], z["trans"][i]) for i in range(3)]
    tint = z["tint"].astype(np.float32)
    state = {n: (np.zeros_like(p), np.zeros_like(p)) for n, p in (("d", grid.densities), ("f", grid.features))}
    for step in range(int(z["steps"])):
        o, d = cams[step % 3]
        col = vo.render_fwd(grid, cfg, o, d)["colour"]
        diff = col - tint
        loss = float((diff.astype(np.float64) ** 2).mean())
        assert abs(loss - z["losses"][step]) < 2e-7, step
        gd, gf = vo.render_bwd(grid, cfg, o, d, (2.0 / diff.size) * diff)
        vo.adam_step(grid.densities, gd, *state["d"], 0.03, 0.9, 0.999, 1e-8, step + 1)
        vo.adam_step(grid.features, gf, *state["f"], 0.03, 0.9, 0.999, 1e-8, step + 1)
    for i, (o, d) in enumerate(cams):
        frame = vo.render_fwd(grid, cfg, o, d)["colour"].reshape(H, W, 3)
        l2 = float(np.sqrt(((frame - z["final_frames"][i]) ** 2).mean()))
        assert l2 < 1e-5, (i, l2)          # BASELINE bar: 1e-3
        assert psnr(frame, z["final_frames"][i]) > 90.0


def test_cast_rays_indexed_equals_full_cast():
    """the oracle's selected-pixel ray generator == its full-image cast_rays (pinned to the reference above) + index"""
    g = load_golden("cast_rays.npz")
    H, W, focal = 23, 31, 40.5
    rng = np.random.default_rng(3)
    poses = []
    for k in range(3):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        poses.append(np.concatenate([q, rng.standard_normal((3, 1))], axis=1).astype(np.float32))
    poses = np.stack(poses)
    full_o, full_d = [], []
    for p in poses:
        o, d = vo.cast_rays(H, W, focal, p[:, :3], p[:, 3])
        full_o.append(o), full_d.append(d)
    full_o, full_d = np.concatenate(full_o), np.concatenate(full_d)
    idx = rng.permutation(3 * H * W)[:500]
    o, d = vo.cast_rays_indexed(H, W, focal, poses, idx)
    assert np.array_equal(o, full_o[idx]) and np.array_equal(d, full_d[idx])
    assert len(g.files) > 0


@pytest.mark.parametrize("deg", [1, 2, 3])
def test_sh_evaluation_host_helper_matches_reference(deg):
    """thre3d_atom...spherical_harmonics.evaluate_spherical_harmonics (host helper, same name as the reference's) on
    the reference's own evaluations (golden G8)"""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vox-e_amd"))
    from thre3d_atom.rendering.volumetric.utils.spherical_harmonics import evaluate_spherical_harmonics

    g = load_golden("render_shdeg.npz")
    out = evaluate_spherical_harmonics(deg, torch.from_numpy(g[f"eval_deg{deg}_coeffs"]), torch.from_numpy(g[f"eval_deg{deg}_dirs"]))
    np.testing.assert_allclose(out.numpy(), g[f"eval_deg{deg}_out"], rtol=0, atol=2e-6)
