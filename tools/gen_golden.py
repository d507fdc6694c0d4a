#!/usr/bin/env python3
"""Generate tests/golden/* by IMPORTING the reference (TAU-VAILab/Vox-E) from /root/reference.

Runs only in the build container (the reference never travels to the GPU box).  The outputs are
data fixtures: inputs + the reference's outputs for the render hot path (SURVEY.md section 8c,
G1..G12).  Nothing of the reference's source text is stored.

    python tools/gen_golden.py            # rewrites tests/golden/

The only import the reference's hot path needs that this image lacks is `easydict`
(thre3d_atom/utils/misc.py:6); it is replaced by a 2-line stand-in module that is never used on
the path.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("VOXE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

_ed = types.ModuleType("easydict")
_ed.EasyDict = dict
sys.modules.setdefault("easydict", _ed)
sys.path.insert(0, REF)

from thre3d_atom.modules.volumetric_model import VolumetricModel  # noqa: E402
from thre3d_atom.rendering.volumetric.render_interface import Rays  # noqa: E402
from thre3d_atom.rendering.volumetric.sample import (  # noqa: E402
    _ray_aabb_intersection,
    sample_uniform_points_on_rays,
)
from thre3d_atom.rendering.volumetric.utils.misc import cast_rays, flatten_rays  # noqa: E402
from thre3d_atom.rendering.volumetric.utils.spherical_harmonics import (  # noqa: E402
    evaluate_spherical_harmonics,
)
from thre3d_atom.thre3d_reprs.renderers import (  # noqa: E402
    SHVoxGridRenderConfig,
    render_sh_voxel_grid,
    render_sh_voxel_grid_attn,
)
from thre3d_atom.thre3d_reprs.voxels import (  # noqa: E402
    VoxelGrid,
    VoxelSize,
    create_voxel_grid_from_saved_info_dict,
    scale_voxel_grid_with_required_output_size,
)
from thre3d_atom.utils.imaging_utils import (  # noqa: E402
    CameraBounds,
    CameraIntrinsics,
    pose_spherical,
)

RADIUS = 4.0311  # modules/sds_trainer.py:45
BOUNDS = CameraBounds(1.8, 6.6)  # tools/convert_from_nerf_blender_dataset.py:15, data/datasets.py:275-276


def synth_pose(i, n):
    """SURVEY.md section 8d synthetic cameras."""
    yaw = 360.0 * i / n
    pitch = 15.0 + 75.0 * ((i * 0.618034) % 1.0)
    return pose_spherical(yaw, pitch, RADIUS)


def focal_for(width):
    return 0.5 * width / np.tan(0.5 * 0.6911112)


def make_grid(dims, nfeat, seed, kind, world=3.0, attn=False, voxel_size=None):
    g = torch.Generator().manual_seed(seed)
    dens = torch.empty((*dims, 1)).uniform_(-1.0, 1.0, generator=g)
    feat = torch.empty((*dims, nfeat)).uniform_(-1.0, 1.0, generator=g)
    if voxel_size is None:
        voxel_size = VoxelSize(*[world / d for d in dims])
    if kind == "softplus":
        act = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.Softplus(),
                   expected_density_scale=100.0 / 3.0)
    elif kind == "softplus_soft":  # translucent: rays integrate through the whole volume
        act = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.Softplus(),
                   expected_density_scale=2.0)
    elif kind == "relu":
        act = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(),
                   expected_density_scale=100.0 / 3.0)
    elif kind == "abs":
        act = dict(density_preactivation=torch.abs, density_postactivation=torch.nn.Identity(),
                   expected_density_scale=1.0)
    else:
        raise ValueError(kind)
    kw = {}
    if attn:
        kw["attn"] = torch.empty((*dims, 1)).uniform_(-3.0, 3.0, generator=g)
    vg = VoxelGrid(dens, feat, voxel_size, tunable=True, **act, **kw)
    return vg


def grid_arrays(vg, prefix):
    d = {
        prefix + "densities": vg.densities.detach().numpy().copy(),
        prefix + "features": vg.features.detach().numpy().copy(),
        prefix + "aabb": np.array(vg.aabb, dtype=np.float64),
        prefix + "voxel_size": np.array(vg.voxel_size, dtype=np.float64),
    }
    if vg.attn is not None:
        d[prefix + "attn"] = vg.attn.detach().numpy().copy()
    return d


def rays_for(h, w, i, n):
    intr = CameraIntrinsics(h, w, focal_for(w))
    pose = synth_pose(i, n)
    r = flatten_rays(cast_rays(intr, pose))
    return r, intr, pose


def np_(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# -------------------------------------------------------------------------------------------------
def g1_cast_rays():
    out = {}
    for tag, (h, w, f) in {"a": (8, 8, 10.0), "b": (24, 32, 44.45), "c": (64, 64, 88.8889)}.items():
        for i in range(3):
            pose = synth_pose(i * 3 + 1, 10)
            rays = cast_rays(CameraIntrinsics(h, w, f), pose)
            out[f"{tag}{i}_hwf"] = np.array([h, w, f], dtype=np.float64)
            out[f"{tag}{i}_rot"] = np_(pose.rotation)
            out[f"{tag}{i}_trans"] = np_(pose.translation)
            out[f"{tag}{i}_origins"] = np_(rays.origins)
            out[f"{tag}{i}_directions"] = np_(rays.directions)
    save("cast_rays.npz", **out)


def g2_g3_sampling():
    out = {}
    rays, _, _ = rays_for(6, 6, 2, 8)
    out["rays_o"], out["rays_d"] = np_(rays.origins), np_(rays.directions)
    out["bounds"] = np.array(BOUNDS, dtype=np.float64)
    for S in (2, 16, 65, 256):
        for mode in ("uniform", "lindisp"):
            sp = sample_uniform_points_on_rays(rays, BOUNDS, S, perturb=False,
                                               linear_disparity_sampling=(mode == "lindisp"))
            out[f"{mode}_S{S}_depths"] = np_(sp.depths)
            out[f"{mode}_S{S}_points"] = np_(sp.points)
        # perturb with the uniforms captured by replaying the seed (sample.py:63 draws torch.rand(R, S))
        for mode in ("uniform", "lindisp"):
            torch.manual_seed(1234 + S)
            u = torch.rand(len(rays), S)
            torch.manual_seed(1234 + S)
            sp = sample_uniform_points_on_rays(rays, BOUNDS, S, perturb=True,
                                               linear_disparity_sampling=(mode == "lindisp"))
            out[f"{mode}_S{S}_jitter"] = np_(u)
            out[f"{mode}_S{S}_jdepths"] = np_(sp.depths)
    # G3: ray / AABB slab test incl. rays that miss
    vg = make_grid((5, 6, 7), 3, 7, "softplus", voxel_size=VoxelSize(0.31, 0.27, 0.23))
    g = torch.Generator().manual_seed(5)
    o = torch.empty(96, 3).uniform_(-3.0, 3.0, generator=g)
    d = torch.empty(96, 3).uniform_(-1.0, 1.0, generator=g)
    d[:8, 0] = 0.0  # axis-parallel rays exercise the +1e-10 guard
    d[8:12, 1] = 0.0
    mixed = Rays(torch.cat([rays.origins, o]), torch.cat([rays.directions, d]))
    b, hit = _ray_aabb_intersection(mixed, BOUNDS, vg.aabb)
    out["aabb"] = np.array(vg.aabb, dtype=np.float64)
    out["aabb_rays_o"], out["aabb_rays_d"] = np_(mixed.origins), np_(mixed.directions)
    out["aabb_bounds"] = np_(b)
    out["aabb_hit"] = np_(hit)
    save("sampling.npz", **out)


def aten_unnormalize_index(norm_points, dims):
    """float32 replay of ATen grid_sampler_unnormalize(align_corners=False) + floor on the reference's
    normalised points: ((n + 1) * size - 1) / 2."""
    idx = np.empty(norm_points.shape, np.int32)
    for a in range(3):
        n = norm_points[:, a].astype(np.float32)
        u = ((n + np.float32(1.0)) * np.float32(dims[a]) - np.float32(1.0)) / np.float32(2.0)
        idx[:, a] = np.floor(u).astype(np.int32)
    return idx


def g4_voxel_forward():
    out = {}
    cases = {
        "aniso": make_grid((5, 6, 7), 3, 11, "softplus", voxel_size=VoxelSize(0.31, 0.27, 0.23)),
        "cube": make_grid((32, 32, 32), 3, 42, "softplus"),
        "abs": make_grid((9, 9, 9), 3, 13, "abs"),
        "relu": make_grid((9, 9, 9), 3, 14, "relu"),
    }
    for tag, vg in cases.items():
        g = torch.Generator().manual_seed(99)
        lo = torch.tensor([r[0] for r in vg.aabb], dtype=torch.float32)
        hi = torch.tensor([r[1] for r in vg.aabb], dtype=torch.float32)
        ext = hi - lo
        pts = lo - 0.15 * ext + torch.rand(2000, 3, generator=g) * (1.3 * ext)
        # border cases: exactly on faces, on voxel centres, on voxel boundaries
        pts[0] = lo
        pts[1] = hi
        pts[2] = (lo + hi) / 2
        pts[3] = lo + ext / torch.tensor(vg.grid_dims, dtype=torch.float32) * 0.5
        pts[4] = lo + ext / torch.tensor(vg.grid_dims, dtype=torch.float32) * 1.0
        with torch.no_grad():
            val = vg(pts)
            npts = vg._normalize_points(pts)
            inside = vg.test_inside_volume(pts)
        # autograd of the point query: loss = sum(forward(points) * G)
        gq = torch.randn(val.shape, generator=g)
        gd, gf = torch.autograd.grad((vg(pts) * gq).sum(), [vg.densities, vg.features])
        out[tag + "_g_out"], out[tag + "_grad_densities"], out[tag + "_grad_features"] = np_(gq), np_(gd), np_(gf)
        out.update(grid_arrays(vg, tag + "_"))
        out[tag + "_points"] = np_(pts)
        out[tag + "_values"] = np_(val)
        out[tag + "_norm_points"] = np_(npts)
        out[tag + "_inside"] = np_(inside)[:, 0]
        out[tag + "_i0"] = aten_unnormalize_index(np_(npts), vg.grid_dims)
    save("voxel_forward.npz", **out)


def render_case(vg, rays, S, white, attn=False, jitter_seed=None, grads=False, **cfg_kw):
    cfg = SHVoxGridRenderConfig(num_samples_per_ray=S, camera_bounds=BOUNDS,
                                perturb_sampled_points=jitter_seed is not None, white_bkgd=white, **cfg_kw)
    res = {}
    if jitter_seed is not None:
        torch.manual_seed(jitter_seed)
        res["jitter"] = np_(torch.rand(len(rays), S))
        torch.manual_seed(jitter_seed)
    proc = render_sh_voxel_grid_attn if attn else render_sh_voxel_grid
    out = proc(vg, rays, cfg)
    col = out.attn if attn else out.colour
    res["colour"] = np_(col)
    res["depth"] = np_(out.depth)[:, 0]
    res["acc"] = np_(out.extra["accumulated_weight"])[:, 0]
    res["disparity"] = np_(out.extra["disparity"])[:, 0]
    if grads:
        g = torch.Generator().manual_seed(43)
        g_col = torch.randn(col.shape, generator=g)
        g_dep = torch.randn(out.depth.shape, generator=g) * 0.25
        g_acc = torch.randn(out.depth.shape, generator=g) * 0.25
        params = [vg.densities, vg.attn if attn else vg.features]
        # colour-only loss (what every reference trainer uses)
        gd, gf = torch.autograd.grad((col * g_col).sum(), params, retain_graph=True)
        res["g_colour"] = np_(g_col)
        res["grad_densities"], res["grad_features"] = np_(gd), np_(gf)
        # colour + depth + accumulated-weight loss
        loss = (col * g_col).sum() + (out.depth * g_dep).sum() + (out.extra["accumulated_weight"] * g_acc).sum()
        gd2, gf2 = torch.autograd.grad(loss, params)
        res["g_depth"], res["g_acc"] = np_(g_dep)[:, 0], np_(g_acc)[:, 0]
        res["grad2_densities"], res["grad2_features"] = np_(gd2), np_(gf2)
    return res


def g5_g6_render():
    out = {}
    rays, _, _ = rays_for(16, 16, 3, 8)
    # add rays that miss the volume entirely and one that starts inside it
    extra_o = torch.tensor([[0.0, 0.0, 6.0], [0.1, -0.2, 0.3], [4.0, 4.0, 4.0]])
    extra_d = torch.tensor([[1.0, 0.0, 0.0], [0.3, 0.5, -0.8], [-1.0, -1.0, -1.02]])
    rays = Rays(torch.cat([rays.origins, extra_o]), torch.cat([rays.directions, extra_d]))
    out["rays_o"], out["rays_d"] = np_(rays.origins), np_(rays.directions)
    out["bounds"] = np.array(BOUNDS, dtype=np.float64)
    kinds = ["softplus", "softplus_soft", "relu", "abs"]
    for kind in kinds:
        vg = make_grid((16, 16, 16), 3, 42, kind)
        out.update(grid_arrays(vg, kind + "_"))
        for S in (16, 64, 256):
            for white in (False, True):
                if S != 64 and kind in ("relu", "abs") and not white:
                    continue
                tag = f"{kind}_S{S}_w{int(white)}_"
                res = render_case(vg, rays, S, white, grads=(S == 64))
                out.update({tag + k: v for k, v in res.items()})
        # jittered + linear disparity + optimized (AABB clipped) sampling
        res = render_case(vg, rays, 64, True, jitter_seed=77, grads=True)
        out.update({f"{kind}_jit_" + k: v for k, v in res.items()})
        res = render_case(vg, rays, 64, True, linear_disparity_sampling=True)
        out.update({f"{kind}_lindisp_" + k: v for k, v in res.items()})
        res = render_case(vg, rays, 64, True, optimized_sampling=True, grads=(kind == "softplus_soft"))
        out.update({f"{kind}_clip_" + k: v for k, v in res.items()})
        # the tester's combination: AABB-clipped AND jittered (modules/testers.py:35 with the default perturbation)
        res = render_case(vg, rays, 64, True, jitter_seed=78, optimized_sampling=True, grads=(kind == "softplus_soft"))
        out.update({f"{kind}_clipjit_" + k: v for k, v in res.items()})
    save("render_sh0.npz", **out)


def g7_attn():
    out = {}
    rays, _, _ = rays_for(12, 12, 5, 8)
    out["rays_o"], out["rays_d"] = np_(rays.origins), np_(rays.directions)
    out["bounds"] = np.array(BOUNDS, dtype=np.float64)
    for kind in ("softplus", "softplus_soft"):
        vg = make_grid((12, 12, 12), 3, 21, kind, attn=True)
        out.update(grid_arrays(vg, kind + "_"))
        for white in (False, True):
            res = render_case(vg, rays, 48, white, attn=True, grads=True)
            out.update({f"{kind}_w{int(white)}_" + k: v for k, v in res.items()})
    save("render_attn.npz", **out)


def g8_sh_degrees():
    out = {}
    rays, _, _ = rays_for(10, 10, 1, 8)
    out["rays_o"], out["rays_d"] = np_(rays.origins), np_(rays.directions)
    out["bounds"] = np.array(BOUNDS, dtype=np.float64)
    g = torch.Generator().manual_seed(3)
    for deg in (1, 2, 3):
        nc = (deg + 1) ** 2
        coeffs = torch.randn(50, 3, nc, generator=g)
        dirs = torch.nn.functional.normalize(torch.randn(50, 3, generator=g), dim=-1)
        out[f"eval_deg{deg}_coeffs"] = np_(coeffs)
        out[f"eval_deg{deg}_dirs"] = np_(dirs)
        out[f"eval_deg{deg}_out"] = np_(evaluate_spherical_harmonics(deg, coeffs, dirs))
        vg = make_grid((8, 8, 8), 3 * nc, 50 + deg, "softplus_soft")
        out.update(grid_arrays(vg, f"deg{deg}_"))
        res = render_case(vg, rays, 32, True, grads=True)
        out.update({f"deg{deg}_full_" + k: v for k, v in res.items()})
        res = render_case(vg, rays, 32, True, grads=True, render_diffuse=True)
        out.update({f"deg{deg}_diffuse_" + k: v for k, v in res.items()})
    save("render_shdeg.npz", **out)


def g9_grid_losses():
    """_density_correlation_loss / _tv_loss_on_grid (modules/sds_trainer.py:507-524,563-567).
    sds_trainer cannot be imported here (diffusers, wandb ... absent), and stubbing half of its
    imports would be a stand-in build; the two functions are pure torch one-liners, so their
    formulas are restated below exactly as written there and differentiated by autograd."""
    out = {}
    g = torch.Generator().manual_seed(9)
    for tag, shape in {"a": (8, 8, 8, 1), "b": (6, 7, 5, 1)}.items():
        reg = torch.empty(shape).uniform_(-1, 1, generator=g)
        sds = (reg + 0.3 * torch.randn(shape, generator=g)).requires_grad_(True)
        eps = 0.0000001
        sds_var = torch.mean((sds - torch.mean(sds)) ** 2)
        reg_var = torch.mean((reg - torch.mean(reg)) ** 2)
        den = torch.sqrt(sds_var * reg_var)
        cov = (sds - torch.mean(sds)) * (reg - torch.mean(reg))
        loss = 1.0 - torch.mean(cov / (den + eps))
        (gr,) = torch.autograd.grad(loss, sds)
        out[f"dcl_{tag}_sds"], out[f"dcl_{tag}_reg"] = np_(sds), np_(reg)
        out[f"dcl_{tag}_loss"], out[f"dcl_{tag}_grad"] = np_(loss), np_(gr)
    for tag, shape in {"a": (8, 8, 8, 1), "b": (5, 6, 7, 3)}.items():
        grid = torch.empty(shape).uniform_(-1, 1, generator=g).requires_grad_(True)
        tv = (grid.diff(dim=0).abs().mean() + grid.diff(dim=1).abs().mean() + grid.diff(dim=2).abs().mean()) / 3
        (gr,) = torch.autograd.grad(tv, grid)
        out[f"tv_{tag}_grid"], out[f"tv_{tag}_loss"], out[f"tv_{tag}_grad"] = np_(grid), np_(tv), np_(gr)
    # torch.optim.Adam trajectory (modules/sds_trainer.py:200-203)
    p = torch.empty(257).uniform_(-1, 1, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=0.03, betas=(0.9, 0.999))
    out["adam_p0"] = np_(p)
    grads, traj = [], []
    for step in range(5):
        gr = torch.randn(257, generator=g) * (10.0 ** (step - 2))
        p.grad = gr.clone()
        opt.step()
        grads.append(np_(gr))
        traj.append(np_(p))
    out["adam_grads"], out["adam_traj"] = np.stack(grads), np.stack(traj)
    save("grid_ops.npz", **out)


def g10_upsample():
    out = {}
    for tag, (dims, new) in {"a": ((8, 8, 8), (16, 16, 16)), "b": ((5, 6, 7), (10, 12, 14)),
                             "c": ((20, 20, 20), (40, 40, 40)), "d": ((6, 6, 6), (9, 11, 7))}.items():
        vg = make_grid(dims, 3, 70, "softplus")
        with torch.no_grad():
            up = scale_voxel_grid_with_required_output_size(vg, new)
        out[f"{tag}_densities"], out[f"{tag}_features"] = np_(vg.densities), np_(vg.features)
        out[f"{tag}_up_densities"], out[f"{tag}_up_features"] = np_(up.densities), np_(up.features)
        out[f"{tag}_voxel_size"] = np.array(vg.voxel_size, dtype=np.float64)
        out[f"{tag}_up_voxel_size"] = np.array(up.voxel_size, dtype=np.float64)
    save("upsample.npz", **out)


def g11_checkpoint():
    """A reference-written checkpoint (VolumetricModel.get_save_info -> torch.save,
    modules/volumetric_model.py:85-99, modules/trainers.py:490-499) for the loader round trip."""
    vg = make_grid((6, 6, 6), 3, 4, "softplus")
    vm = VolumetricModel(vg, render_sh_voxel_grid,
                         SHVoxGridRenderConfig(64, BOUNDS, white_bkgd=True, render_num_samples_per_ray=128),
                         device=torch.device("cpu"))
    extra = {"camera_bounds": BOUNDS, "camera_intrinsics": CameraIntrinsics(20, 20, focal_for(20)),
             "hemispherical_radius": RADIUS}
    path = os.path.join(OUT, "ref_checkpoint.pth")
    torch.save(vm.get_save_info(extra), path)
    rays, _, _ = rays_for(8, 8, 2, 8)
    with torch.no_grad():
        out = vm.render_rays(rays, perturb_sampled_points=False)
    save("ref_checkpoint_render.npz", rays_o=np_(rays.origins), rays_d=np_(rays.directions),
         colour=np_(out.colour), depth=np_(out.depth)[:, 0])
    # sanity: the reference can read its own file
    data = torch.load(path, weights_only=False)
    create_voxel_grid_from_saved_info_dict(data)
    print(f"ref_checkpoint.pth: {os.path.getsize(path) / 1024:.1f} KiB")


def g12_frames():
    out = {}
    vg = make_grid((32, 32, 32), 3, 42, "softplus")
    out.update(grid_arrays(vg, ""))
    vm = VolumetricModel(vg, render_sh_voxel_grid, SHVoxGridRenderConfig(128, BOUNDS, white_bkgd=True),
                         device=torch.device("cpu"))
    intr = CameraIntrinsics(64, 64, focal_for(64))
    frames, rots, trans = [], [], []
    for i in range(8):
        pose = synth_pose(i, 8)
        r = vm.render(pose, intr, perturb_sampled_points=False)
        frames.append(np_(r.colour))
        rots.append(np_(pose.rotation))
        trans.append(np_(pose.translation))
    out["hwf"] = np.array([64, 64, focal_for(64)], dtype=np.float64)
    out["bounds"] = np.array(BOUNDS, dtype=np.float64)
    out["frames"] = np.stack(frames).astype(np.float32)
    out["rot"], out["trans"] = np.stack(rots), np.stack(trans)
    save("frames32.npz", **out)


def g13_refinement():
    """Refinement stage (modules/refinement_functions.py): the graph `build_graph` constructs (:182-287) and
    `calc_loss_on_attn_grid` (:42-76).  The module imports two packages this image lacks: `wandb` (logging only,
    never reached with log_wandb=False) and `maxflow` (PyMaxflow).  `maxflow` is replaced by a RECORDING stand-in:
    it logs add_nodes / add_tedge / add_edge and solves nothing (maxflow() -> 0, get_segment() -> 0), so the
    fixture pins the graph CONSTRUCTION only; the min-cut solver is pinned elsewhere (scipy / brute force)."""
    recorded = []

    class _RecGraph:
        def __class_getitem__(cls, item):
            return cls

        def __init__(self):
            self.tedges, self.edges, self.n = [], [], 0
            recorded.append(self)

        def add_nodes(self, n):
            self.n = int(n)
            return np.arange(self.n)

        def add_tedge(self, i, s, t):
            self.tedges.append((int(i), float(s), float(t)))

        def add_edge(self, i, j, c, rc):
            self.edges.append((int(i), int(j), float(c), float(rc)))

        def maxflow(self):
            return 0.0

        def get_segment(self, i):
            return 0

    mf = types.ModuleType("maxflow")
    mf.Graph = _RecGraph
    sys.modules.setdefault("maxflow", mf)
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    from thre3d_atom.modules.refinement_functions import build_graph, calc_loss_on_attn_grid  # noqa: E402

    out = {}
    g = torch.Generator().manual_seed(13)

    def scene(dims, fill):
        dens = torch.empty(*dims, 1).uniform_(-1.0, fill, generator=g)
        feat = torch.sigmoid(torch.empty(*dims, 3).uniform_(-2, 2, generator=g))  # refinement_functions.py:378
        edit = torch.empty(*dims, 1).uniform_(-1, 1, generator=g)
        obj = torch.empty(*dims, 1).uniform_(-1, 1, generator=g)
        return dens, feat, edit, obj

    cases = {
        # enough edit voxels -> thresholded edit seeds + random object seeds
        "a": dict(dims=(6, 7, 8), fill=0.25, kw=dict(K=5.0, sigma=0.1, edit_mask_thresh=0.9, num_obj_voxels_thresh=12,
                                                    min_num_edit_voxels=2, top_k_edit_thresh=5, top_k_obj_thresh=4)),
        # too few edit voxels -> top-k fallback for both seed sets
        "b": dict(dims=(5, 6, 7), fill=0.4, kw=dict(K=5.0, sigma=0.1, edit_mask_thresh=0.999, num_obj_voxels_thresh=10,
                                                   min_num_edit_voxels=50, top_k_edit_thresh=6, top_k_obj_thresh=5)),
        # down-sampled grid branch (max / average pooling by 4)
        "c": dict(dims=(16, 16, 16), fill=0.01, kw=dict(K=5.0, sigma=0.1, edit_mask_thresh=0.999, num_obj_voxels_thresh=8,
                                                       min_num_edit_voxels=2, top_k_edit_thresh=5, top_k_obj_thresh=4,
                                                       downsample_grid=True, downsample_factor=4)),
    }
    for tag, case in cases.items():
        dens, feat, edit, obj = scene(case["dims"], case["fill"])
        torch.manual_seed(1300 + ord(tag))
        with torch.no_grad():
            _, idx_values = build_graph(feat, dens, edit, obj, **case["kw"])
        rec = recorded[-1]
        out[f"{tag}_densities"], out[f"{tag}_features"] = np_(dens), np_(feat)
        out[f"{tag}_edit_attn"], out[f"{tag}_obj_attn"] = np_(edit), np_(obj)
        out[f"{tag}_seed"] = np.array(1300 + ord(tag))
        out[f"{tag}_node_idx"] = idx_values.numpy().astype(np.int32)
        out[f"{tag}_tedges"] = np.array(rec.tedges, dtype=np.float64).reshape(-1, 3)
        out[f"{tag}_edges"] = np.array(rec.edges, dtype=np.float64).reshape(-1, 4)
        for k, v in case["kw"].items():
            out[f"{tag}_kw_{k}"] = np.array(v)

    # masked-L1 attention loss + gradient
    for tag, (h, w) in {"a": (12, 10), "b": (7, 9)}.items():
        render = (torch.empty(h * w, 1).uniform_(-0.5, 1.0, generator=g)).requires_grad_(True)
        amap = torch.empty(h, w).uniform_(0, 1, generator=g)
        loss = calc_loss_on_attn_grid(render, amap, token="edit", global_step=1)
        (gr,) = torch.autograd.grad(loss, render)
        out[f"loss_{tag}_render"], out[f"loss_{tag}_map"] = np_(render), np_(amap)
        out[f"loss_{tag}_value"], out[f"loss_{tag}_grad"] = np_(loss), np_(gr)
    save("refine_graph.npz", **out)


def g14_edit_trajectory():
    """A short edit run ENTIRELY by the reference: differentiable `VolumetricModel.render_rays` (jitter off) ->
    image loss -> `torch.optim.Adam(betas=(0.9, 0.999))` (the optimiser of modules/sds_trainer.py:200-203,332-333),
    cycling over 3 cameras.  The loss stands in for the SDS gradient (a flat-tint L2, like the stub guidance of the
    GPU trainer tests).  Fixture = start grid, per-step losses, final grid, final frames: the build must land on the
    same edited renders (BASELINE.json: "edited renders within 1e-3 L2 of reference")."""
    out = {}
    vg = make_grid((32, 32, 32), 3, 77, "softplus")
    out.update(grid_arrays(vg, "start_"))
    cfg = SHVoxGridRenderConfig(64, BOUNDS, perturb_sampled_points=False, white_bkgd=True)
    vm = VolumetricModel(vg, render_sh_voxel_grid, cfg, device=torch.device("cpu"))
    hw = 48
    cams = [rays_for(hw, hw, i, 8) for i in (0, 3, 5)]
    tint = torch.tensor([0.9, 0.2, 0.1])
    opt = torch.optim.Adam([{"params": vg.parameters(), "lr": 0.03}], betas=(0.9, 0.999))
    losses = []
    steps = 12
    for step in range(steps):
        rays = cams[step % 3][0]
        col = vm.render_rays(rays).colour
        loss = ((col - tint) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    with torch.no_grad():
        frames = [np_(vm.render_rays(c[0]).colour).reshape(hw, hw, 3) for c in cams]
    out["hwf"] = np.array([hw, hw, focal_for(hw)], dtype=np.float64)
    out["bounds"] = np.array(BOUNDS, dtype=np.float64)
    out["rot"] = np.stack([np_(c[2].rotation) for c in cams])
    out["trans"] = np.stack([np_(c[2].translation) for c in cams])
    out["tint"], out["lr"], out["steps"], out["samples"] = tint.numpy(), np.array(0.03), np.array(steps), np.array(64)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["final_densities"], out["final_features"] = np_(vg.densities), np_(vg.features)
    out["final_frames"] = np.stack(frames).astype(np.float32)
    save("edit_trajectory.npz", **out)


def g15_attention_maps():
    """Token attention maps (thre3d_atom/thre3d_reprs/cross_attn.py:425-467).  That module imports diffusers / cv2 and
    calls `.cuda()`, so it cannot be imported here; its Gaussian filter (thre3d_atom/thre3d_reprs/gaussian_smoothing.py,
    torch only) IS imported and applied, and the ~15 lines around it (layer / head average, token slice, reflect pad,
    bilinear up-sampling) are restated below exactly as written there."""
    from thre3d_atom.thre3d_reprs.gaussian_smoothing import GaussianSmoothing  # noqa: E402

    g = torch.Generator().manual_seed(15)
    layers = [torch.softmax(torch.randn(4, pix, 12, generator=g), dim=-1) for pix in (256, 1024, 256, 64, 256)]  # 12 tokens keep the fixture small
    res, prompts = 16, ["a prompt"]
    out_maps = []
    for item in layers:                      # aggregate_attention(..., select=0)
        if item.shape[1] == res ** 2:
            out_maps.append(item.reshape(len(prompts), -1, res, res, item.shape[-1])[0])
    agg = torch.cat(out_maps, dim=0)
    agg = agg.sum(0) / agg.shape[0]
    text = agg[:, :, 1:-1]                   # compute_max_attention_per_index
    indices = [2, 5, 9]
    h, w = 37, 52
    maps = []
    for i in [j - 1 for j in indices]:
        image = text[:, :, i]
        smoothing = GaussianSmoothing(channels=1, kernel_size=3, sigma=0.5, dim=2)
        inp = torch.nn.functional.pad(image.unsqueeze(0).unsqueeze(0), (1, 1, 1, 1), mode="reflect")
        image = smoothing(inp).squeeze(0).squeeze(0)
        up = torch.nn.Upsample(size=(h, w), mode="bilinear")(image.view(1, 1, res, res))[0][0]
        u_inp = torch.nn.functional.pad(up.unsqueeze(0).unsqueeze(0), (1, 1, 1, 1), mode="reflect")
        maps.append(smoothing(u_inp).squeeze(0).squeeze(0))
    out = {f"layer{n}": np_(a) for n, a in enumerate(layers)}
    out["indices"], out["hw"] = np.array(indices), np.array([h, w])
    out["average"] = np_(agg)
    out["maps"] = np.stack([np_(m) for m in maps])
    out["kernel"] = np_(GaussianSmoothing(1, 3, 0.5, 2).weight[0, 0])
    save("attention_maps.npz", **out)


def g16_sds_boundary():
    """SDS boundary a19: the REFERENCE's thre3d_atom/thre3d_reprs/sd.py (scoreDistillationLoss.training_step :365-385 ->
    StableDiffusion.train_step :174-234 -> SpecifyGradient :20-34) executed on the tiny stand-in SD stack of
    tests/sds_standins.py (diffusers / weights are absent here).  Recorded per step: the rendered colours handed in, the
    direction word, every random draw (timestep, VAE sample noise, diffusion noise), the returned loss, the gradient that
    reaches the colours, the max-step ratio after the schedule.  tests/test_sds_boundary.py replays the draws through
    the build's sd.py."""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), ""))      # tests/
    import sds_standins as st  # noqa: E402

    h, w = 20, 28
    out = {"hw": np.array([h, w])}
    with st.installed():
        from thre3d_atom.thre3d_reprs import sd as ref_sd  # noqa: E402

        torch.manual_seed(16)
        guide = ref_sd.scoreDistillationLoss(torch.device("cpu"), "a yarn doll", t_sched_start=2, t_sched_freq=2,
                                             t_sched_gamma=0.5, directional=True)
        plain = ref_sd.scoreDistillationLoss(torch.device("cpu"), "a yarn doll", directional=False)
        directions = ["front", "side", "overhead", "back", "side", "front"]   # (step 6: the 0.22 floor of the ratio)
        n_steps = len(directions)
        for step in range(1, n_steps + 1):
            colour = torch.sigmoid(torch.randn(h * w, 3)).requires_grad_(True)
            with st.RandomTape(record=True) as tape:
                loss = guide.training_step(colour, h, w, directions=[directions[step - 1]], global_step=step)
                loss.backward()
            out[f"s{step}_colour"] = np_(colour)
            out[f"s{step}_grad"] = np_(colour.grad)
            out[f"s{step}_loss"] = np_(loss)
            out[f"s{step}_ratio"] = np.array(guide.get_current_max_step_ratio())
            out[f"s{step}_max_step"] = np.array(guide.sd_model.max_step)
            for i, d in enumerate(tape.draws):
                out[f"s{step}_draw{i}"] = d.numpy()
            out[f"s{step}_ndraws"] = np.array(len(tape.draws))
        # non-directional guidance with a per-call log-variance (sd.py:226-227).  (One image per call: train_step pairs the
        # [uncond, text] embeddings with cat([latents] * 2), so the reference itself only works for a batch of one.)
        colour = torch.sigmoid(torch.randn(h * w, 3)).requires_grad_(True)
        logvar = torch.tensor(0.3)
        with st.RandomTape(record=True) as tape:
            loss = plain.training_step(colour, h, w, global_step=7, logvars=logvar)
            loss.backward()
        out["b_colour"], out["b_grad"], out["b_logvar"] = np_(colour), np_(colour.grad), np_(logvar)
        for i, d in enumerate(tape.draws):
            out[f"b_draw{i}"] = d.numpy()
        out["b_ndraws"] = np.array(len(tape.draws))
        out["directions"] = np.array(directions)
        out["steps"] = np.array(n_steps)
        out["num_tokens"] = np.array(guide.sd_model.get_num_tokens("a yarn doll"))
        out["text_front"] = np_(guide.text_encodings["front"])
        out["alphas"] = np_(guide.sd_model.alphas)
    save("sds_boundary.npz", **out)


def _reference_functions(path, names):
    """compile the named top-level functions of a reference module from ITS OWN source text (read here, in the build container,
    never stored) into a namespace holding the torch names the module imports for them -- for modules whose import chain
    needs packages this image lacks (diffusers, wandb, lpips, tensorboard ...) but whose functions are pure torch"""
    import ast

    from torch import Tensor
    from torch.nn.functional import l1_loss, mse_loss

    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), [n.name for n in body]
    ns = {"torch": torch, "Tensor": Tensor, "mse_loss": mse_loss, "l1_loss": l1_loss, "np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def g17_regulariser_modes():
    """density_correlation_loss_fn with l2_mode / l1_mode (modules/sds_trainer.py:494-505) and _feature_correlation_loss
    (:526-534): the reference's own functions (compiled from its source file, see _reference_functions), values and autograd
    gradients.  Includes exact ties (sds == regular at some voxels: sign(0) = 0 in l1_loss's gradient)."""
    dcl_fn, _, fcl_fn = _reference_functions(os.path.join(REF, "thre3d_atom", "modules", "sds_trainer.py"),
                                             ["density_correlation_loss_fn", "_density_correlation_loss", "_feature_correlation_loss"])
    out = {}
    g = torch.Generator().manual_seed(17)
    for tag, shape in {"a": (8, 8, 8, 1), "b": (6, 7, 5, 1)}.items():
        reg = torch.empty(shape).uniform_(-1, 1, generator=g)
        sds = reg + 0.3 * torch.randn(shape, generator=g)
        sds.view(-1)[::7] = reg.view(-1)[::7]            # ties
        sds.requires_grad_(True)
        out[f"dens_{tag}_sds"], out[f"dens_{tag}_reg"] = np_(sds), np_(reg)
        for mode in ("l2", "l1"):
            loss, aux = dcl_fn(sds_density=sds, regular_density=reg, l2_mode=mode == "l2", l1_mode=mode == "l1")
            assert aux is None
            (gr,) = torch.autograd.grad(loss, sds)
            out[f"dens_{tag}_{mode}_loss"], out[f"dens_{tag}_{mode}_grad"] = np_(loss), np_(gr)
        loss, _ = dcl_fn(sds_density=sds, regular_density=reg, l2_mode=True, l1_mode=True)   # (both flags: l2 wins, :498-500)
        out[f"dens_{tag}_both_loss"] = np_(loss)
    for tag, shape in {"a": (8, 8, 8, 3), "b": (5, 6, 7, 3), "sh1": (4, 5, 3, 12), "attn": (6, 5, 4, 1)}.items():
        reg = torch.empty(shape).uniform_(-2, 2, generator=g)
        sds = (reg + 0.5 * torch.randn(shape, generator=g)).requires_grad_(True)
        loss = fcl_fn(sds_features=sds, regular_features=reg, density_cov_grid=None)
        (gr,) = torch.autograd.grad(loss, sds)
        out[f"feat_{tag}_sds"], out[f"feat_{tag}_reg"] = np_(sds), np_(reg)
        out[f"feat_{tag}_loss"], out[f"feat_{tag}_grad"] = np_(loss), np_(gr)
    save("reg_modes.npz", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:      # python tools/gen_golden.py g17_regulariser_modes  -> only that fixture
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    torch.set_num_threads(8)
    g1_cast_rays()
    g2_g3_sampling()
    g4_voxel_forward()
    g5_g6_render()
    g7_attn()
    g8_sh_degrees()
    g9_grid_losses()
    g10_upsample()
    g11_checkpoint()
    g12_frames()
    g13_refinement()
    g14_edit_trajectory()
    g15_attention_maps()
    g16_sds_boundary()
    g17_regulariser_modes()
