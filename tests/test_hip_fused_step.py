"""Fused grid optimiser step (voxe_render_bwd_acc + voxe_grid_adam_step) against the split path it replaces
(voxe_render_bwd + voxe_adam_step on each tensor): parameters, Adam moments and the next render must be identical
BIT FOR BIT over several steps when both consume the same gradient bits (the backward itself sums float atomics in
a launch-dependent order, so two backward LAUNCHES agree only to rounding) -- for both gradient layouts (LDS-window backward: linear; scatter backward: 2x2x2
bricks), odd grid sizes, extra (regulariser) gradients, accumulation of two renders and a frozen tensor."""
import pytest
import torch

from synth import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import abi, dispatch, ops

AABB = ((-1.5, 1.5),) * 3
LR = 3e-3


def _scene(side, nfeat, hw, ordered, dims=None, cam=3):
    dev = torch.device("cuda", 0)
    dens, feat = random_grid(side, nfeat)
    if dims is not None:
        dens, feat = dens[: dims[0], : dims[1], : dims[2]].contiguous(), feat[: dims[0], : dims[1], : dims[2]].contiguous()
    yaw, pitch = synth_pose_angles(cam, 100)
    pose = pose_spherical(yaw, pitch, RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
    if not ordered:
        perm = torch.randperm(ro.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
        ro, rd = ro[perm].contiguous(), rd[perm].contiguous()
    return dens.to(dev), feat.to(dev), ro, rd


def _region_to_gradients(region, layout, dens, spec, C):
    """the workspace gradient region -> (d_densities, d_features) exactly like unpack_grad_kernel: de-brick, split the
    channels, chain rule of the density pre-activation (sign(raw * scale) * scale for abs, scale otherwise)"""
    X, Y, Z = dens.shape[:3]
    dev = dens.device
    if layout == abi.GRAD_BRICKED:
        bx, by, bz = (X + 1) // 2, (Y + 1) // 2, (Z + 1) // 2
        x, y, z = torch.meshgrid(torch.arange(X, device=dev), torch.arange(Y, device=dev), torch.arange(Z, device=dev), indexing="ij")
        slot = ((((x // 2) * by + (y // 2)) * bz + (z // 2)) * 8) + (x % 2) * 4 + (y % 2) * 2 + (z % 2)
        g = region[: bx * by * bz * 8 * C].view(-1, C)[slot.reshape(-1)].view(X, Y, Z, C)
    else:
        g = region[: X * Y * Z * C].view(X, Y, Z, C)
    scale = torch.tensor(spec.density_scale, dtype=torch.float32, device=dev)
    chain = torch.sign(dens * scale) * scale if spec.density_pre_act == abi.ACT_ABS else scale
    return (g[..., C - 1:] * chain).contiguous(), g[..., : C - 1].contiguous()


class _Run:
    """one optimisation run.  mode "fused": voxe_render_bwd_acc + voxe_grid_adam_step.  mode "split": voxe_render_bwd +
    voxe_adam_step per tensor (its own backward launches: float atomics make it equal only to rounding).  mode "shadow":
    the split optimiser fed with the gradient decoded from a fused run's workspace (bit-exact comparison)."""

    def __init__(self, mode, dens, feat, spec, params, ro, rd, cout, freeze_density=False, extras=False, renders=1):
        self.mode, self.spec, self.params, self.ro, self.rd = mode, spec, params, ro, rd
        self.dens, self.feat = dens.clone(), feat.clone()
        self.freeze_density, self.extras, self.renders = freeze_density, extras, renders
        dev, R = dens.device, ro.shape[0]
        self.out = [torch.empty((R, n), dtype=torch.float32, device=dev) for n in (cout, 1, 1, 1)]
        gen = torch.Generator().manual_seed(11)
        self.g_colour = torch.randn((R, cout), generator=gen).to(dev)
        self.g_depth = torch.randn((R, 1), generator=gen).to(dev) * 0.1
        self.g_acc = torch.randn((R, 1), generator=gen).to(dev) * 0.1
        self.m = [torch.zeros_like(self.dens), torch.zeros_like(self.feat)]
        self.v = [torch.zeros_like(self.dens), torch.zeros_like(self.feat)]
        self.ws = ops.Workspace()
        self.n = 0
        self.layouts = set()
        self.egen = torch.Generator().manual_seed(5)
        self.decoded = None   # fused mode: (d_densities, d_features) decoded from the workspace before the step

    def _extras(self):
        if not self.extras:
            return None, None
        return ((torch.randn(self.dens.shape, generator=self.egen) * 1e-3).to(self.dens.device),
                (torch.randn(self.feat.shape, generator=self.egen) * 1e-3).to(self.dens.device))

    def _split_update(self, d_dens, d_feat, ex_d, ex_f):
        if self.extras:   # autograd would add the regulariser gradient onto the render gradient
            d_dens, d_feat = d_dens + ex_d, d_feat + ex_f
        if not self.freeze_density:
            ops.adam_step_(self.dens, d_dens, self.m[0], self.v[0], self.n, LR)
        ops.adam_step_(self.feat, d_feat, self.m[1], self.v[1], self.n, LR)

    def shadow_step(self, decoded):
        self.n += 1
        self._split_update(*decoded, *self._extras())

    def step(self):
        self.n += 1
        ex_d, ex_f = self._extras()
        d_dens, d_feat = torch.zeros_like(self.dens), torch.zeros_like(self.feat)
        layout = abi.GRAD_ANY
        for r in range(self.renders):
            rng = (9, 100 * self.n + r)
            ops.render_fwd_into(self.spec, self.params, self.dens, self.feat, self.ro, self.rd, None, *self.out, self.ws, rng)
            args = (self.spec, self.params, self.dens, self.feat, self.ro, self.rd, None, self.out[0], self.out[1],
                    self.out[2], self.g_colour, self.g_depth, self.g_acc)
            if self.mode == "fused":
                layout = ops.render_bwd_acc(*args, self.ws, rng, zero_first=(self.n == 1 and r == 0))
                self.layouts.add(layout)
            else:
                ops.render_bwd_into(*args, d_dens, d_feat, self.ws, rng, accumulate=(r > 0))
        if self.mode == "fused":
            region = ops.workspace_grad_view(self.spec, self.dens, self.feat, self.ws)
            self.decoded = _region_to_gradients(region.clone(), layout, self.dens, self.spec, self.feat.shape[-1] + 1)
            sd = None if self.freeze_density else (self.m[0], self.v[0])
            ops.grid_adam_step_(self.spec, self.dens, self.feat, layout, self.ws, self.n, LR, sd, (self.m[1], self.v[1]),
                                ex_d, ex_f)
            assert float(region.abs().max()) == 0.0      # the step leaves the gradient region cleared
        else:
            self._split_update(d_dens, d_feat, ex_d, ex_f)

    def render(self, ws=None):
        ws = ws if ws is not None else self.ws
        ops.render_fwd_into(self.spec, self.params, self.dens, self.feat, self.ro, self.rd, None, *self.out, ws, (9, 7))
        return [t.clone() for t in self.out]


def _state(r: _Run):
    return (("densities", r.dens), ("features", r.feat), ("exp_avg d", r.m[0]), ("exp_avg f", r.m[1]),
            ("exp_avg_sq d", r.v[0]), ("exp_avg_sq f", r.v[1]))


def _compare(a: _Run, b: _Run):
    for (name, x), (_, y) in zip(_state(a), _state(b)):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} differ, max {float((x - y).abs().max()):.3e}"


def _rel(x, y):
    return float((x - y).norm() / y.norm().clamp_min(1e-30))


CASES = {
    # name: side, F, hw, ordered rays, dims, kind, pre_act, post_act, kwargs
    "sh0_tile_linear": (48, 3, 96, True, None, "sh", "identity", "softplus", {}),
    "sh0_scatter_bricked": (40, 3, 64, False, None, "sh", "identity", "softplus", {}),
    "sh0_odd_dims_bricked": (40, 3, 64, False, (37, 40, 33), "sh", "identity", "softplus", {}),
    "sh0_odd_dims_tile": (40, 3, 96, True, (37, 40, 33), "sh", "identity", "softplus", {}),
    # 33^3 = 35 937 voxels: an odd count -- the coalesced grid step takes 140 chunks of 256, the per-voxel kernel the last 97
    "sh0_odd_voxel_count_tile": (40, 3, 96, True, (33, 33, 33), "sh", "identity", "softplus", {}),
    "sh0_frozen_density_tile": (40, 3, 96, True, None, "sh", "identity", "softplus", {"freeze_density": True}),
    # 33^3 = 35 937 voxels: an odd count -- the coalesced grid step takes 140 chunks of 256, the per-voxel kernel the last 97
    "sh0_odd_voxel_count_tile": (33, 3, 96, True, (33, 33, 33), "sh", "identity", "softplus", {}),
    "sh0_frozen_density_tile": (40, 3, 96, True, None, "sh", "identity", "softplus", {"freeze_density": True}),
    "sh0_preact_abs_relu": (40, 3, 96, True, None, "sh", "abs", "relu", {}),
    "sh0_extras_two_renders": (40, 3, 96, True, None, "sh", "identity", "softplus", {"extras": True, "renders": 2}),
    "sh0_extras_scatter": (40, 3, 64, False, None, "sh", "identity", "softplus", {"extras": True, "renders": 2}),
    "attn_frozen_density": (40, 1, 96, True, None, "attn", "identity", "relu", {"freeze_density": True}),
    "sh1_linear": (24, 12, 48, True, None, "sh1", "identity", "softplus", {}),
    "sh1_scatter_bricked_odd": (24, 12, 48, False, (23, 24, 21), "sh1", "identity", "softplus", {"extras": True}),
    "sh1_frozen_density": (24, 12, 48, True, None, "sh1", "abs", "relu", {"freeze_density": True}),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_fused_step_equals_split_step(case):
    side, nfeat, hw, ordered, dims, kind, pre, post, kw = CASES[case]
    dens, feat, ro, rd = _scene(side, nfeat, hw, ordered, dims)
    acts = {"identity": abi.ACT_IDENTITY, "abs": abi.ACT_ABS, "relu": abi.ACT_RELU, "softplus": abi.ACT_SOFTPLUS}
    spec = ops.GridSpec(aabb=AABB, density_scale=100.0 / 3.0 if post == "softplus" else 1.0, density_pre_act=acts[pre],
                        density_post_act=acts[post], feature_kind=abi.FEAT_ATTN if kind == "attn" else abi.FEAT_SH)
    params = ops.RenderParams(num_samples=96, near=NEAR, far=FAR, perturb=True, white_bkgd=kind != "attn",
                              sh_degree=1 if kind == "sh1" else 0, image_width=hw if ordered else 0,
                              # (48x48 is below the shipped tile threshold: the ordered cases ask for the LDS-window backward)
                              dispatch=dispatch.TILE_ALWAYS if ordered else None)
    cout = 1 if kind == "attn" else 3
    shadow = _Run("shadow", dens, feat, spec, params, ro, rd, cout, **kw)
    split = _Run("split", dens, feat, spec, params, ro, rd, cout, **kw)
    fused = _Run("fused", dens, feat, spec, params, ro, rd, cout, **kw)
    for _ in range(4):
        fused.step()
        shadow.shadow_step(fused.decoded)      # same gradient bits through voxe_adam_step: must match exactly
        _compare(shadow, fused)
        split.step()                           # own backward launches: equal up to the order of the float atomics
    for (name, x), (_, y) in zip(_state(fused), _state(split)):
        if name.startswith("exp_avg_sq") or not float(y.abs().max()):
            continue
        assert _rel(x, y) < 5e-4, f"{name}: rel-L2 {_rel(x, y):.3e}"
    expect = abi.GRAD_LINEAR if ordered else abi.GRAD_BRICKED   # LDS-window backward / line-dense scatter
    assert fused.layouts == {expect}
    assert not torch.equal(fused.feat, feat)             # the run really moved the parameters
    if kw.get("freeze_density"):
        assert torch.equal(fused.dens, dens) and float(fused.m[0].abs().max()) == 0.0
    # the packed grid the fused step left in the workspace == a fresh pack of the updated tensors
    reused = fused.render()
    fresh = fused.render(ops.Workspace())
    for x, y in zip(reused, fresh):
        assert torch.equal(x, y)


def test_gradient_region_view_matches_split_gradient():
    """what a data-parallel job all-reduces: the workspace gradient region, linear layout = [X,Y,Z,(features, density)]"""
    dens, feat, ro, rd = _scene(40, 3, 96, True)
    spec = ops.GridSpec(aabb=AABB, density_scale=100.0 / 3.0)
    params = ops.RenderParams(num_samples=96, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=96)
    run = _Run("fused", dens, feat, spec, params, ro, rd, 3)
    rng = (9, 1)
    ops.render_fwd_into(spec, params, run.dens, run.feat, ro, rd, None, *run.out, run.ws, rng)
    args = (spec, params, run.dens, run.feat, ro, rd, None, run.out[0], run.out[1], run.out[2], run.g_colour, run.g_depth, run.g_acc)
    layout = ops.render_bwd_acc(*args, run.ws, rng)
    assert layout == abi.GRAD_LINEAR
    region = ops.workspace_grad_view(spec, run.dens, run.feat, run.ws)
    assert region.numel() * 4 == 40 ** 3 * 4 * 4       # even dims: the bricked layout needs no padding
    d_dens, d_feat = _region_to_gradients(region.clone(), layout, run.dens, spec, 4)
    s_dens, s_feat = torch.zeros_like(dens), torch.zeros_like(feat)
    ops.render_bwd_into(*args, s_dens, s_feat, run.ws, rng)       # (re-runs the backward: equal up to atomics order)
    assert _rel(d_feat, s_feat) < 1e-5 and _rel(d_dens, s_dens) < 1e-5
    # doubling the region (what an all-reduce over two identical ranks does) doubles the step's gradient
    ops.render_bwd_acc(*args, run.ws, rng)
    region = ops.workspace_grad_view(spec, run.dens, run.feat, run.ws)
    d_dens, d_feat = _region_to_gradients(region.clone(), layout, run.dens, spec, 4)
    region.mul_(2.0)
    ref = _Run("shadow", dens, feat, spec, params, ro, rd, 3)
    ops.adam_step_(ref.dens, d_dens * 2.0, ref.m[0], ref.v[0], 1, LR)
    ops.adam_step_(ref.feat, d_feat * 2.0, ref.m[1], ref.v[1], 1, LR)
    ops.grid_adam_step_(spec, run.dens, run.feat, layout, run.ws, 1, LR, (run.m[0], run.v[0]), (run.m[1], run.v[1]))
    _compare(ref, run)
    assert float(ops.workspace_grad_view(spec, run.dens, run.feat, run.ws).abs().max()) == 0.0


def test_fused_step_argument_errors():
    dens, feat, ro, rd = _scene(16, 3, 16, True)
    spec = ops.GridSpec(aabb=AABB)
    with pytest.raises(ops.VoxeError):
        ops.grid_adam_step_(spec, dens, feat, abi.GRAD_LINEAR, ops.Workspace(), 1, LR)   # no gradient in the workspace
    ws = ops.Workspace()
    params = ops.RenderParams(num_samples=16, near=NEAR, far=FAR, image_width=16)
    out = [torch.empty((ro.shape[0], n), device=dens.device) for n in (3, 1, 1, 1)]
    ops.render_fwd_into(spec, params, dens, feat, ro, rd, None, *out, ws)
    ops.render_bwd_acc(spec, params, dens, feat, ro, rd, None, out[0], out[1], out[2], torch.ones_like(out[0]), None, None, ws)
    m = torch.zeros_like(dens)
    with pytest.raises(ops.VoxeError):   # moments of the wrong size
        ops.grid_adam_step_(spec, dens, feat, abi.GRAD_LINEAR, ws, 1, LR, (m, m), (m, m))
    with pytest.raises(ops.VoxeError):   # unknown layout
        ops.grid_adam_step_(spec, dens, feat, 7, ws, 1, LR, (m, m.clone()))
    with pytest.raises(ops.VoxeError):   # step < 1
        ops.grid_adam_step_(spec, dens, feat, abi.GRAD_LINEAR, ws, 0, LR, (m, m.clone()))


@pytest.mark.parametrize("ordered,dims,slabs", [(True, (40, 40, 40), [(0, 20), (20, 40)]),
                                                (True, (37, 40, 33), [(0, 11), (11, 12), (12, 37)]),
                                                (False, (40, 36, 33), [(0, 10), (10, 20), (20, 30), (30, 40)]),
                                                (False, (37, 40, 33), [(0, 36), (36, 37)])])
def test_slab_steps_equal_full_step(ordered, dims, slabs):
    """x_range: the step applied slab by slab (what the ranks of a sharded optimiser do, each on its own slab) leaves
    exactly the parameters, moments, packed grid and cleared gradient of one full step"""
    hw = 96 if ordered else 64
    dens, feat, ro, rd = _scene(40, 3, hw, ordered, dims)
    spec = ops.GridSpec(aabb=AABB, density_scale=100.0 / 3.0)
    params = ops.RenderParams(num_samples=96, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw if ordered else 0)
    part = _Run("fused", dens, feat, spec, params, ro, rd, 3)
    rng = (9, 100)
    ops.render_fwd_into(spec, params, part.dens, part.feat, ro, rd, None, *part.out, part.ws, rng)
    args = (spec, params, part.dens, part.feat, ro, rd, None, part.out[0], part.out[1], part.out[2], part.g_colour,
            part.g_depth, part.g_acc)
    layout = ops.render_bwd_acc(*args, part.ws, rng)
    region = ops.workspace_grad_view(spec, part.dens, part.feat, part.ws)
    d_dens, d_feat = _region_to_gradients(region.clone(), layout, part.dens, spec, 4)
    ref = _Run("shadow", dens, feat, spec, params, ro, rd, 3)     # the full step through voxe_adam_step, same gradient bits
    ref.shadow_step((d_dens, d_feat))
    for x0, x1 in slabs:
        ops.grid_adam_step_(spec, part.dens, part.feat, layout, part.ws, 1, LR, (part.m[0], part.v[0]), (part.m[1], part.v[1]),
                            x_range=(x0, x1))
    _compare(ref, part)
    assert float(region.abs().max()) == 0.0
    got = ops.workspace_packed_view(spec, part.dens, part.feat, part.ws).view(*dims, 4)
    assert torch.equal(got[..., :3], part.feat) and torch.equal(got[..., 3:], part.dens * spec.density_scale)
    if layout == abi.GRAD_BRICKED:
        with pytest.raises(ops.VoxeError):      # bricks pair x-planes: a slab cannot start on an odd plane
            ops.grid_adam_step_(spec, part.dens, part.feat, layout, part.ws, 2, LR, (part.m[0], part.v[0]),
                                (part.m[1], part.v[1]), x_range=(1, 3))
    with pytest.raises(ops.VoxeError):
        ops.grid_adam_step_(spec, part.dens, part.feat, layout, part.ws, 2, LR, (part.m[0], part.v[0]),
                            (part.m[1], part.v[1]), x_range=(0, dims[0] + 1))


def test_density_correlation_inside_the_grid_step_equals_the_separate_pass():
    """VoxeGridRegularisers (r04): the SDS edit's density-correlation regulariser evaluated inside voxe_grid_adam_step ==
    voxe_dcl_fwd_bwd into a gradient tensor + the same step with it as extra_d_densities (weight folded into the two gradient
    constants in double instead of multiplying the float gradient: equal to float rounding), loss value included; slabs and
    frozen densities are rejected"""
    side, hw = 40, 96
    dens, feat, ro, rd = _scene(side, 3, hw, True, None)
    ref = (dens * 0.8 + 0.1 * torch.randn_like(dens)).contiguous()
    spec = ops.GridSpec(aabb=AABB, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY, density_post_act=abi.ACT_SOFTPLUS)
    params = ops.RenderParams(num_samples=96, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw)
    gcol = torch.randn((ro.shape[0], 3), generator=torch.Generator().manual_seed(5)).to(ro.device)
    weight = 200.0
    runs = {}
    for mode in ("separate", "in_step"):
        d, f = dens.clone(), feat.clone()
        st_d, st_f = (torch.zeros_like(d), torch.zeros_like(d)), (torch.zeros_like(f), torch.zeros_like(f))
        ws = ops.Workspace()
        outs = [torch.empty((ro.shape[0], k), device=ro.device) for k in (3, 1, 1, 1)]
        loss_val = torch.zeros((), device=ro.device)
        vals = []
        for it in range(1, 4):
            ops.render_fwd_into(spec, params, d, f, ro, rd, None, *outs, ws, (9, it))
            layout = ops.render_bwd_acc(spec, params, d, f, ro, rd, None, outs[0], outs[1], outs[2], gcol, None, None, ws, (9, it),
                                        zero_first=(it == 1))
            if mode == "separate":
                dd = d.clone().requires_grad_(True)
                l = ops.density_correlation_loss(dd, ref)
                (l * weight).backward()
                vals.append(float(l.detach()))
                ops.grid_adam_step_(spec, d, f, layout, ws, it, 1e-2, state_densities=st_d, state_features=st_f,
                                    extra_d_densities=dd.grad.contiguous())
            else:
                ops.grid_adam_step_(spec, d, f, layout, ws, it, 1e-2, state_densities=st_d, state_features=st_f,
                                    dcl_reference=ref, dcl_weight=weight, dcl_loss=loss_val)
                vals.append(float(loss_val))
        runs[mode] = (d, f, st_d[0], vals)
    a, b = runs["separate"], runs["in_step"]
    assert all(abs(x - y) < 2e-6 for x, y in zip(a[3], b[3])), (a[3], b[3])
    # (two runs of the render backward differ by the order of their float atomics, and Adam's first steps turn the rounding
    #  noise of near-zero gradients into +-lr moves: first moments to 5e-4, parameters against their movement)
    assert _rel(a[2], b[2]) < 5e-4, _rel(a[2], b[2])
    assert _rel(a[0] - dens, b[0] - dens) < 0.02 and _rel(a[1] - feat, b[1] - feat) < 0.02, (_rel(a[0] - dens, b[0] - dens), _rel(a[1] - feat, b[1] - feat))
    assert float((b[0] - dens).abs().max()) > 1e-3
    d, f = dens.clone(), feat.clone()
    st_d, st_f = (torch.zeros_like(d), torch.zeros_like(d)), (torch.zeros_like(f), torch.zeros_like(f))
    with pytest.raises(ops.VoxeError):       # the moments are over the whole grid: no slabs
        ops.grid_adam_step_(spec, d, f, 0, ws, 1, 1e-2, state_densities=st_d, state_features=st_f, x_range=(0, side // 2),
                            dcl_reference=ref, dcl_weight=weight)
    with pytest.raises(ops.VoxeError):       # frozen densities have nothing to regularise
        ops.grid_adam_step_(spec, d, f, 0, ws, 1, 1e-2, state_densities=None, state_features=st_f, dcl_reference=ref, dcl_weight=weight)


@pytest.mark.parametrize("case", ["whole", "slab", "features_frozen", "extras"])
def test_wide_texel_grid_step_chunks_tail_and_slabs(case):
    """view-dependent grids take the fused grid step 64 voxels at a time with 16-byte accesses (r04); the tail, x-slabs that are
    not 16-byte aligned and bricked buffers take the element kernel: a hand-made gradient region -> the same parameters, moments
    and packed grid as voxe_adam_step per tensor on the decoded gradient, bit for bit"""
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(31)
    dims, F = (8, 6, 6), 12                      # 36 voxels per x plane: slab [2, 6) = voxels 72 .. 215 = 2 chunks + a tail of 16
    C = F + 1
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=3.0, density_pre_act=abi.ACT_ABS, density_post_act=abi.ACT_SOFTPLUS)
    dens = torch.empty((*dims, 1)).uniform_(-1, 1, generator=gen).to(dev)
    feat = torch.empty((*dims, F)).uniform_(-1, 1, generator=gen).to(dev)
    ws = ops.Workspace()
    ws.ensure(4 * 2 * (dens.numel() + feat.numel()) * 2 + (1 << 16), dev)
    ws.buf.zero_()
    m = [torch.zeros_like(dens), torch.zeros_like(feat)]
    v = [torch.zeros_like(dens), torch.zeros_like(feat)]
    ref = [dens.clone(), feat.clone(), [t.clone() for t in m], [t.clone() for t in v]]
    x_range = (2, 6) if case == "slab" else None
    x0, x1 = x_range if x_range else (0, dims[0])
    for step in (1, 2, 3):
        region = ops.workspace_grad_view(spec, dens, feat, ws)
        g = torch.empty(dens.numel() * C).normal_(generator=gen).to(dev)
        region[: g.numel()] = g
        ex_d = torch.empty(dens.shape).normal_(generator=gen).to(dev) if case == "extras" else None
        ex_f = torch.empty(feat.shape).normal_(generator=gen).to(dev) if case == "extras" else None
        d_d, d_f = _region_to_gradients(region.clone(), abi.GRAD_LINEAR, ref[0], spec, C)
        if ex_d is not None:
            d_d, d_f = d_d + ex_d, d_f + ex_f
        ops.grid_adam_step_(spec, dens, feat, abi.GRAD_LINEAR, ws, step, LR, (m[0], v[0]),
                            None if case == "features_frozen" else (m[1], v[1]), ex_d, ex_f, x_range=x_range)
        sl = slice(x0, x1)
        for i, (p, grad) in enumerate(((ref[0], d_d), (ref[1], d_f))):
            if i == 1 and case == "features_frozen":
                continue
            ps, gs, ms, vs = (t[sl].contiguous() for t in (p, grad, ref[2][i], ref[3][i]))
            ops.adam_step_(ps, gs, ms, vs, step, LR)
            p[sl], ref[2][i][sl], ref[3][i][sl] = ps, ms, vs
        for nm, a, b in (("densities", dens, ref[0]), ("features", feat, ref[1]), ("exp_avg d", m[0], ref[2][0]),
                         ("exp_avg f", m[1], ref[2][1]), ("exp_avg_sq d", v[0], ref[3][0]), ("exp_avg_sq f", v[1], ref[3][1])):
            assert torch.equal(a, b), (case, step, nm, float((a - b).abs().max()))
        packed = ops.workspace_packed_view(spec, dens, feat, ws).view(*dims, C)[sl]
        assert torch.equal(packed[..., :F], feat[sl]) and torch.equal(packed[..., F:], (dens[sl] * 3.0).abs())
        assert float(ops.workspace_grad_view(spec, dens, feat, ws)[x0 * 36 * C: x1 * 36 * C].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["sh0_tile", "sh0_scatter_bricked_odd", "sh0_preact_abs_relu", "attn_frozen_density", "sh1_linear"])
def test_fused_step_first_step_against_the_oracle(case):
    """the equivalence test above compares product paths with each other.  Here ONE fused step (voxe_render_fwd ->
    voxe_render_bwd_acc -> voxe_grid_adam_step from the zero Adam state) against the CPU oracle: its render gradient w.r.t. the
    API tensors (chain rule of the density pre-activation included) and its restatement of torch.optim.Adam -- exp_avg =
    (1 - beta1) * gradient in the rel-L2 sense, parameters on every voxel whose gradient is far above the float atomics' noise"""
    import numpy as np

    from oracle import voxe_oracle as vo
    from voxe_hip.desc import make_render_cfg

    side, nfeat, hw, ordered, dims, kind, pre, post, freeze = {
        "sh0_tile": (40, 3, 96, True, None, "sh", "identity", "softplus", False),
        "sh0_scatter_bricked_odd": (40, 3, 64, False, (37, 40, 33), "sh", "identity", "softplus", False),
        "sh0_preact_abs_relu": (40, 3, 96, True, None, "sh", "abs", "relu", False),
        "attn_frozen_density": (40, 1, 96, True, None, "attn", "identity", "relu", True),
        "sh1_linear": (24, 12, 48, True, None, "sh1", "identity", "softplus", False),
    }[case]
    dens, feat, ro, rd = _scene(side, nfeat, hw, ordered, dims)
    acts = {"identity": abi.ACT_IDENTITY, "abs": abi.ACT_ABS, "relu": abi.ACT_RELU, "softplus": abi.ACT_SOFTPLUS}
    scale = 100.0 / 3.0 if post == "softplus" else 1.0
    fk = abi.FEAT_ATTN if kind == "attn" else abi.FEAT_SH
    deg = 1 if kind == "sh1" else 0
    spec = ops.GridSpec(aabb=AABB, density_scale=scale, density_pre_act=acts[pre], density_post_act=acts[post], feature_kind=fk)
    params = ops.RenderParams(num_samples=96, near=NEAR, far=FAR, perturb=True, white_bkgd=kind != "attn", sh_degree=deg,
                              image_width=hw if ordered else 0, dispatch=dispatch.TILE_ALWAYS if ordered else None)
    cout = 1 if kind == "attn" else 3
    R = ro.shape[0]
    g_col = torch.randn((R, cout), generator=torch.Generator().manual_seed(9)).to(dens.device)
    # ---- oracle: gradient of sum(colour * g_col) w.r.t. the API tensors, then Adam
    grid = vo.Grid(dens.cpu().numpy(), feat.cpu().numpy(), [(-1.5, 1.5)] * 3, scale, acts[pre], acts[post], fk)
    cfg = make_render_cfg(96, NEAR, FAR, perturb=True, white_bkgd=kind != "attn", sh_degree=deg, seed=21, rng_offset=5)
    gd, gf = vo.render_bwd(grid, cfg, ro.cpu().numpy(), rd.cpu().numpy(), g_col.cpu().numpy())
    # ---- one fused step
    d, f = dens.clone(), feat.clone()
    st_d = None if freeze else (torch.zeros_like(d), torch.zeros_like(d))
    st_f = (torch.zeros_like(f), torch.zeros_like(f))
    outs = [torch.empty((R, n), device=d.device) for n in (cout, 1, 1, 1)]
    ws = ops.Workspace()
    ops.render_fwd_into(spec, params, d, f, ro, rd, None, *outs, ws, (21, 5))
    layout = ops.render_bwd_acc(spec, params, d, f, ro, rd, None, outs[0], outs[1], outs[2], g_col, None, None, ws, (21, 5),
                                zero_first=True, want_densities=not freeze)
    ops.grid_adam_step_(spec, d, f, layout, ws, 1, LR, state_densities=st_d, state_features=st_f)
    checks = [(st_f[0], gf, f, feat)] + ([] if freeze else [(st_d[0], gd, d, dens)])
    for m1, g_ref, p_new, p_old in checks:
        g_ref = np.asarray(g_ref, dtype=np.float32).reshape(-1)
        g_lib = m1.cpu().numpy().reshape(-1) / np.float32(0.1)
        assert np.linalg.norm(g_lib - g_ref) / np.linalg.norm(g_ref) < 1e-4
        p_ref = p_old.cpu().numpy().reshape(-1).copy()
        vo.adam_step(p_ref, g_ref, np.zeros_like(p_ref), np.zeros_like(p_ref), LR, 0.9, 0.999, 1e-8, 1)
        big = np.abs(g_ref) > 1e-3 * np.abs(g_ref).max()
        assert big.sum() > 100
        assert np.abs(p_new.cpu().numpy().reshape(-1) - p_ref)[big].max() < 1e-3 * LR
    if freeze:
        assert torch.equal(d, dens)
