"""Helpers of the `-m gpu` parity tests: run the same case through the HIP C ABI (via voxe_hip.ops)
and through the CPU oracle."""
import numpy as np
import torch

from voxe_hip import abi, ops

from oracle import voxe_oracle as vo

DEV = torch.device("cuda:0")


def spec_of(grid: vo.Grid) -> ops.GridSpec:
    return ops.GridSpec(
        aabb=tuple((float(a), float(b)) for a, b in grid.aabb),
        density_scale=float(grid.density_scale),
        density_pre_act=grid.density_pre_act,
        density_post_act=grid.density_post_act,
        feature_kind=grid.feature_kind,
    )


def params_of(cfg, **over) -> ops.RenderParams:
    p = ops.RenderParams(
        num_samples=cfg.num_samples, near=float(cfg.near), far=float(cfg.far), perturb=bool(cfg.perturb),
        linear_disparity=bool(cfg.linear_disparity), aabb_clip=bool(cfg.aabb_clip),
        white_bkgd=bool(cfg.white_bkgd), sh_degree=cfg.sh_degree, render_diffuse=bool(cfg.render_diffuse),
        term_eps=float(cfg.term_eps), image_width=cfg.image_width, image_height=cfg.image_height, deterministic=bool(cfg.deterministic),
    )
    for k, v in over.items():
        setattr(p, k, v)
    return p


def t(a, requires_grad=False):
    if a is None:
        return None
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x.requires_grad_(True) if requires_grad else x


def n(x):
    return x.detach().cpu().numpy()


def hip_forward(grid: vo.Grid, cfg, rays_o, rays_d, jitter=None, rng=(0, 0), **over):
    with torch.no_grad():
        c, d, a, disp = ops.render(spec_of(grid), params_of(cfg, **over), t(grid.densities), t(grid.features),
                                   t(rays_o), t(rays_d), t(jitter), rng=rng)
    torch.cuda.synchronize()
    return {"colour": n(c), "depth": n(d)[:, 0], "acc": n(a)[:, 0], "disparity": n(disp)[:, 0]}


def hip_backward(grid: vo.Grid, cfg, rays_o, rays_d, g_colour, g_depth=None, g_acc=None, jitter=None,
                 rng=(0, 0), **over):
    d = t(grid.densities, True)
    f = t(grid.features, True)
    c, dep, acc, _ = ops.render(spec_of(grid), params_of(cfg, **over), d, f, t(rays_o), t(rays_d), t(jitter), rng=rng)
    loss = (c * t(g_colour)).sum()
    if g_depth is not None:
        loss = loss + (dep[:, 0] * t(g_depth)).sum()
    if g_acc is not None:
        loss = loss + (acc[:, 0] * t(g_acc)).sum()
    loss.backward()
    torch.cuda.synchronize()
    return n(d.grad), n(f.grad)


def hip_probe(grid: vo.Grid, cfg, rays_o, rays_d, jitter=None, rng=(0, 0), **over):
    out = ops.sample_probe(spec_of(grid), params_of(cfg, **over), t(grid.densities), t(grid.features),
                           t(rays_o), t(rays_d), t(jitter), rng=rng)
    torch.cuda.synchronize()
    return {k: n(v) for k, v in out.items()}
