"""A plain PyTorch restatement of the render path (SH-0 grid): the "reference-equivalent" baseline that bench.py times on
the SAME GPU under PyTorch-ROCm (`gpu_baseline`) and on the host cores.

NOT part of the product path and not imported by it: nothing in voxe_hip.ops / thre3d_atom reaches this module; it exists
so that north_star's "x times the reference on one GPU" has a denominator that can be measured on the GPU box, where the
reference itself cannot travel.  It is written from the oracle's arithmetic (oracle/voxe_cpu.c), which restates
  sample   rendering/volumetric/sample.py:15-68        uniform depths + stratified jitter
  process  rendering/volumetric/process.py:20-174      VoxelGrid.forward (2 x F.grid_sample, thre3d_reprs/voxels.py:287-332),
                                                       SH degree 0 colour, inside-AABB mask
  accumulate rendering/volumetric/accumulate.py:31-113 alpha, exclusive cumprod transmittance, weights, white background
as the same ~40 tensor ops with [rays x samples] temporaries, differentiated by autograd -- i.e. the execution model of
the reference (one ATen kernel per op, grid_sampler_3d_backward for the grid gradient), not its source.
tests/test_torch_baseline.py pins it to the reference's own outputs and gradients (tests/golden/render_sh0.npz)."""
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

C0 = 0.28209479177387814          # SH degree-0 basis
ZERO_PLUS = 1e-10


def render(densities: torch.Tensor, features: torch.Tensor, aabb: Sequence[Tuple[float, float]], density_scale: float,
           rays_o: torch.Tensor, rays_d: torch.Tensor, num_samples: int, near: float, far: float,
           jitter: Optional[torch.Tensor] = None, perturb: bool = False, white_bkgd: bool = True,
           post_act: str = "softplus"):
    """densities [X,Y,Z,1], features [X,Y,Z,3] (SH-0), rays [R,3] -> colour [R,3], depth [R], acc [R].
    jitter: uniforms [R,S] (None with perturb=True: torch.rand)."""
    dev = densities.device
    R, S = rays_o.shape[0], num_samples
    t = torch.linspace(0.0, 1.0, S, device=dev)
    z = (near * (1.0 - t) + far * t).expand(R, S)
    if perturb or jitter is not None:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], dim=-1)
        lower = torch.cat([z[:, :1], mids], dim=-1)
        u = jitter if jitter is not None else torch.rand(R, S, device=dev)
        z = lower + (upper - lower) * u
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]                      # [R,S,3]
    lo = torch.tensor([a[0] for a in aabb], dtype=torch.float32, device=dev)
    hi = torch.tensor([a[1] for a in aabb], dtype=torch.float32, device=dev)
    scale = 2.0 / (hi - lo)
    bias = -1.0 - lo * scale
    n = pts * scale + bias                                                            # [-1, 1] over the box
    inside = ((pts > lo) & (pts < hi)).all(dim=-1)                                    # strict test (voxels.py:263-285)
    # grid_sample wants (W, H, D) = (z, y, x) coordinates for a [1, C, X, Y, Z] volume
    coords = n[..., [2, 1, 0]].reshape(1, R, S, 1, 3)
    dvol = (densities * density_scale).permute(3, 0, 1, 2)[None]                      # pre-activation = identity
    fvol = features.permute(3, 0, 1, 2)[None]
    raw = F.grid_sample(dvol, coords, mode="bilinear", padding_mode="zeros", align_corners=False)[0, 0, :, :, 0]
    feat = F.grid_sample(fvol, coords, mode="bilinear", padding_mode="zeros", align_corners=False)[0, :, :, :, 0]
    sigma = {"softplus": F.softplus, "relu": torch.relu, "identity": lambda x: x}[post_act](raw)
    sigma = sigma * inside
    rgb = torch.sigmoid(C0 * feat.permute(1, 2, 0)) * inside[..., None]               # [R,S,3]
    deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10, device=dev)], dim=-1)
    deltas = deltas * rays_d.norm(dim=-1, keepdim=True)
    alpha = 1.0 - torch.exp(-sigma * deltas)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1, device=dev), 1.0 - alpha + ZERO_PLUS], dim=-1), dim=-1)[:, :-1]
    w = alpha * trans
    colour = (w[..., None] * rgb).sum(dim=1)
    acc = w.sum(dim=-1)
    depth = (w * z).sum(dim=-1)
    if white_bkgd:
        colour = colour + (1.0 - acc[:, None])
    return colour, depth, acc


def time_step(densities, features, aabb, density_scale, rays_o, rays_d, g_colour, num_samples, near, far, chunk: int,
              steps: int = 3, warmup: int = 1, lr: float = 1e-4, median: bool = False):
    """seconds per step of (render fwd + autograd bwd over ray chunks of `chunk` rays + torch.optim.Adam): the reference's
    own way of processing an image (parallel_rays_chunk_size, modules/volumetric_model.py:152-176)"""
    import time

    d = densities.clone().requires_grad_(True)
    f = features.clone().requires_grad_(True)
    opt = torch.optim.Adam([d, f], lr=lr, betas=(0.9, 0.999))
    R = rays_o.shape[0]
    sync = torch.cuda.synchronize if d.is_cuda else (lambda: None)

    def step():
        opt.zero_grad(set_to_none=True)
        for a in range(0, R, chunk):
            b = min(R, a + chunk)
            colour, _, _ = render(d, f, aabb, density_scale, rays_o[a:b], rays_d[a:b], num_samples, near, far, perturb=True)
            (colour * g_colour[a:b]).sum().backward()
        opt.step()

    for _ in range(warmup):
        step()
    sync()
    if median:      # every step timed on its own, the median reported (SURVEY 8(d) protocol)
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step()
            sync()
            times.append(time.perf_counter() - t0)
        return sorted(times)[len(times) // 2]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps
