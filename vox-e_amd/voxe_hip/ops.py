"""torch-level operators over libvoxe_hip.so: the fused render (autograd.Function), ray casting and
the whole-grid passes.  Tensors are plumbing (device memory + streams); all compute is in the HIP
library.  Nothing here falls back to torch ops or to the CPU oracle."""
import ctypes as C
import dataclasses
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import abi
from . import dispatch as _dispatch
from .desc import make_grid_desc, make_render_cfg
from .runtime import VoxeError, check, ensure_gfx950, f32c, lib, ptr, require_device, stream_ptr


@dataclass(frozen=True)
class GridSpec:
    """Static (non-tensor) description of a voxel grid: what VoxeGridDesc needs besides pointers."""
    aabb: Tuple[Tuple[float, float], Tuple[float, float], Tuple[float, float]]
    density_scale: float = 1.0
    density_pre_act: int = abi.ACT_IDENTITY
    density_post_act: int = abi.ACT_SOFTPLUS
    feature_kind: int = abi.FEAT_SH


@dataclass
class RenderParams:
    """Everything VoxeRenderCfg holds except the RNG stream and the packed-grid reuse flag."""
    num_samples: int
    near: float
    far: float
    perturb: bool = False
    linear_disparity: bool = False
    aabb_clip: bool = False
    white_bkgd: bool = False
    sh_degree: int = 0
    render_diffuse: bool = False
    term_eps: float = 0.0
    image_width: int = 0
    image_height: int = 0     # > 0 (with image_width): the rays are K = R / (H * W) images, one after the other
    deterministic: bool = False   # backward in 64-bit fixed point: bit-reproducible (test / race-check mode)
    linear_grad: bool = False     # render_bwd_acc: write VOXE_GRAD_LINEAR whatever kernel runs (deferred-gradient mode)
    dispatch: Optional[_dispatch.Dispatch] = None   # kernel routes / tuning of THIS call (VoxeDispatch); None = dispatch.current()


@dataclass
class DeferredGrad:
    """state of the deferred-gradient mode of one grid: which layout the accumulated gradient has and whether the region
    holds anything since the last optimiser step"""
    layout: int = abi.GRAD_ANY
    dirty: bool = False
    want_densities: bool = True
    want_features: bool = True
    clean_ptr: int = 0        # data_ptr of the workspace buffer whose gradient region is known to be cleared / in use


class Workspace:
    """Caller-owned scratch of the render entry points: [packed grid | packed gradient].
    Remembers which grid values it holds packed so consecutive calls can skip the pack pass."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.key = None        # which grid values are packed in the buffer
        self.state_key = None  # which forward call's per-ray depth-segment states it holds
        # A differentiable forward leaves its per-ray states here for its backward.  When a second differentiable
        # forward arrives before that backward (two renders in one loss: specular + diffuse), it runs in `sibling`
        # (own buffers) instead of overwriting the states -- otherwise the first backward must re-march its rays.
        self.pending = False
        self.pending_version = None   # (densities._version, features._version) of the forward that set `pending`
        self.sibling: Optional["Workspace"] = None
        # deferred-gradient mode (FusedGridAdam): backward passes of renders through this workspace (or its sibling)
        # LEAVE the grid gradient in this workspace's gradient region instead of returning .grad tensors
        self.deferred: Optional["DeferredGrad"] = None
        self.recon_scratch: dict = {}   # device scratch of recon_step_ (rays, targets, outputs of one fused iteration)
        # recon_prefetch_: the library's side stream may be writing into this buffer (and reading `prefetch_keepalive`) until the
        # next recon_step_ of the owning workspace has been enqueued
        self.prefetch_inflight = False
        self.prefetch_keepalive = None
        self.recon_cache = None         # descriptors of the last recon_step_ (what a hint for the next one repeats)

    def __del__(self):
        # a hint in flight (recon_prefetch_) writes this buffer from a stream torch's caching allocator knows nothing about
        try:
            if self.prefetch_inflight and self.buf is not None:
                torch.cuda.synchronize(self.buf.device)
        except Exception:
            pass

    def for_differentiable_forward(self, version=None) -> "Workspace":
        """the workspace a differentiable forward should run in.  A pending forward whose backward never came (the caller
        dropped the graph: the parameters have moved on since) no longer blocks this workspace."""
        if self.pending and version is not None and self.pending_version != version:
            self.pending = False
        if not self.pending:
            return self
        if self.sibling is None:
            self.sibling = Workspace()
        if self.sibling.pending and version is not None and self.sibling.pending_version != version:
            self.sibling.pending = False
        return self.sibling if not self.sibling.pending else self

    def ensure(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            old = self.buf
            if old is not None and self.prefetch_inflight:
                torch.cuda.synchronize(old.device)   # (a stream torch's allocator knows nothing about is still using the old buffer)
                self.prefetch_inflight = False
            self.recon_cache = None                  # (its descriptors point into the old buffer -- and would keep it alive)
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            keep = (old is not None and self.deferred is not None and self.deferred.dirty
                    and old.device == self.buf.device)
            if keep:
                # deferred-gradient mode: an accumulated gradient (and the packed grid in front of it) lives at fixed
                # offsets from the start of the buffer -- a render that needs a bigger workspace must not lose it
                self.buf[: old.numel()].copy_(old)
                self.deferred.clean_ptr = self.buf.data_ptr()
            else:
                self.key = None
            self.state_key = None
        return self.buf

    def invalidate(self):
        self.key = None
        self.state_key = None


def _pack_key(spec: GridSpec, densities: torch.Tensor, features: torch.Tensor):
    return (densities.data_ptr(), densities._version, features.data_ptr(), features._version,
            tuple(features.shape), spec.density_scale, spec.density_pre_act, spec.feature_kind)


def _state_key(pack_key, params: RenderParams, rays_o, rays_d, jitter, rng, route=None):
    """identity of a forward call: the backward may consume the ray states only of exactly this call -- and only when it
    resolves to the same kernels (`route` = voxe_render_route: ray-ordered and space-binned renders keep different
    tables, and the choice also depends on process-level tuning switches that may change between the two calls)"""
    fwd = tuple((k, v) for k, v in vars(params).items() if k not in ("linear_grad", "deterministic", "dispatch"))   # backward-only knobs
    fwd += (("dispatch", params.dispatch if params.dispatch is not None else _dispatch.current()),)
    return (pack_key, fwd, rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0],
            None if jitter is None else jitter.data_ptr(), tuple(rng), route)


def _route(g, c, R) -> int:
    return int(lib().voxe_render_route(C.byref(g), C.byref(c), int(R)))


def _descs(spec: GridSpec, params: RenderParams, densities, features, seed, rng_offset, reuse):
    X, Y, Z, F = features.shape
    g = make_grid_desc(densities.data_ptr(), features.data_ptr(), (X, Y, Z), F, spec.aabb,
                       spec.density_scale, spec.density_pre_act, spec.density_post_act, spec.feature_kind)
    c = make_render_cfg(params.num_samples, params.near, params.far, params.perturb,
                        params.linear_disparity, params.aabb_clip, params.white_bkgd, params.sh_degree,
                        params.render_diffuse, params.term_eps, seed, rng_offset, reuse, params.image_width,
                        image_height=params.image_height, deterministic=params.deterministic,
                        linear_grad=params.linear_grad,
                        dispatch=(params.dispatch if params.dispatch is not None else _dispatch.current()).struct())
    return g, c


def _validate_inputs(densities, features, rays_o, rays_d, jitter, params: RenderParams):
    for name, t in (("densities", densities), ("features", features), ("rays_o", rays_o), ("rays_d", rays_d)):
        require_device(t, f"voxe render ({name})")
    if densities.dim() != 4 or features.dim() != 4 or densities.shape[:3] != features.shape[:3] or densities.shape[3] != 1:
        raise VoxeError(f"grid tensors must be [X,Y,Z,1] and [X,Y,Z,F]; got {tuple(densities.shape)}, {tuple(features.shape)}")
    if rays_o.dim() != 2 or rays_o.shape[1] != 3 or rays_o.shape != rays_d.shape:
        raise VoxeError(f"rays must be flat [R,3]; got {tuple(rays_o.shape)}, {tuple(rays_d.shape)}")
    if jitter is not None and tuple(jitter.shape) != (rays_o.shape[0], params.num_samples):
        raise VoxeError(f"jitter must be [R,S]={rays_o.shape[0], params.num_samples}; got {tuple(jitter.shape)}")


def _next_rng():
    """(seed, offset) of the in-kernel counter-hash jitter stream, tied to torch's CPU generator so that
    torch.manual_seed() makes renders reproducible (no device sync involved)."""
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    offset = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    return seed, offset


def render_fwd_into(spec: GridSpec, params: RenderParams, densities, features, rays_o, rays_d, jitter,
                    colour, depth, acc, disparity, workspace: Workspace, rng=(0, 0), keep_for_backward: bool = True) -> None:
    """voxe_render_fwd on caller-provided output tensors (contiguous float32 on one device, no autograd).
    keep_for_backward=False (inference): the forward skips what only a backward of the same rays would read (the per-sample
    values of view-dependent grids); a later backward on this workspace re-marches."""
    device = densities.device
    ensure_gfx950(device)
    L = lib()
    R = rays_o.shape[0]
    key = _pack_key(spec, densities, features)
    g, c = _descs(spec, params, densities, features, rng[0], rng[1], workspace.key == key)
    with torch.cuda.device(device):
        c.ray_state_valid = 0 if keep_for_backward else -1     # (-1: the size query leaves out everything only a backward reads)
        ws = workspace.ensure(L.voxe_workspace_bytes(C.byref(g), C.byref(c), R), device)
        c.reuse_packed_grid = int(workspace.key == key)
        check(L.voxe_render_fwd(C.byref(g), C.byref(c), ptr(rays_o), ptr(rays_d), R, ptr(jitter), ptr(colour),
                                ptr(depth), ptr(acc), ptr(disparity), ptr(ws), ws.numel(),
                                stream_ptr(device)), "voxe_render_fwd")
    workspace.key = key
    workspace.state_key = _state_key(key, params, rays_o, rays_d, jitter, rng, _route(g, c, R)) if keep_for_backward else None


def render_bwd_into(spec: GridSpec, params: RenderParams, densities, features, rays_o, rays_d, jitter,
                    colour, depth, acc, g_colour, g_depth, g_acc, d_densities, d_features,
                    workspace: Workspace, rng=(0, 0), accumulate: bool = False) -> None:
    """voxe_render_bwd into caller-provided gradient tensors (either may be None to skip it)."""
    device = densities.device
    L = lib()
    R = rays_o.shape[0]
    key = _pack_key(spec, densities, features)
    g, c = _descs(spec, params, densities, features, rng[0], rng[1], workspace.key == key)
    with torch.cuda.device(device):
        ws = workspace.ensure(L.voxe_workspace_bytes(C.byref(g), C.byref(c), R), device)
        c.reuse_packed_grid = int(workspace.key == key)
        c.ray_state_valid = int(workspace.state_key == _state_key(key, params, rays_o, rays_d, jitter, rng, _route(g, c, R)))
        check(L.voxe_render_bwd(C.byref(g), C.byref(c), ptr(rays_o), ptr(rays_d), R, ptr(jitter), ptr(colour),
                                ptr(depth), ptr(acc), ptr(g_colour), ptr(g_depth), ptr(g_acc),
                                ptr(d_densities), ptr(d_features), int(accumulate), ptr(ws), ws.numel(),
                                stream_ptr(device)), "voxe_render_bwd")
    workspace.key = key


class _RenderFn(torch.autograd.Function):
    """colour, depth, acc, disparity = render(densities, features | rays)  with the fused HIP kernels."""

    @staticmethod
    def forward(ctx, densities, features, rays_o, rays_d, jitter, spec, params, workspace, rng):
        ctx.set_materialize_grads(False)
        device = densities.device
        ensure_gfx950(device)
        dens, feat = f32c(densities.detach()), f32c(features.detach())
        ro, rd = f32c(rays_o.detach()), f32c(rays_d.detach())
        jit = None if jitter is None else f32c(jitter.detach())
        R = ro.shape[0]
        cout = 1 if spec.feature_kind == abi.FEAT_ATTN else 3
        colour = torch.empty((R, cout), dtype=torch.float32, device=device)
        depth = torch.empty((R, 1), dtype=torch.float32, device=device)
        acc = torch.empty((R, 1), dtype=torch.float32, device=device)
        disp = torch.empty((R, 1), dtype=torch.float32, device=device)
        ctx.main_workspace = workspace
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:  # (grad mode itself is off inside Function.forward)
            version = (densities._version, features._version)
            workspace = workspace.for_differentiable_forward(version)
            workspace.pending, workspace.pending_version = True, version
        render_fwd_into(spec, params, dens, feat, ro, rd, jit, colour, depth, acc, disp, workspace, rng,
                        keep_for_backward=bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        ctx.spec, ctx.params, ctx.workspace, ctx.rng = spec, params, workspace, rng
        ctx.save_for_backward(densities, features, ro, rd, jit, colour, depth, acc)
        return colour, depth, acc, disp

    @staticmethod
    def backward(ctx, g_colour, g_depth, g_acc, g_disp):
        densities, features, ro, rd, jit, colour, depth, acc = ctx.saved_tensors
        need_d, need_f = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_d or need_f):
            return (None,) * 9
        device = densities.device
        spec, params, workspace = ctx.spec, ctx.params, ctx.workspace
        g_depth = None if g_depth is None else f32c(g_depth)
        g_acc = None if g_acc is None else f32c(g_acc)
        if g_disp is not None:
            # disparity = 1 / max(1e-10, depth / acc)  (accumulate.py:85-88): chained into d_depth and d_acc by one kernel
            g_depth, g_acc = disparity_bwd(depth, acc, f32c(g_disp), g_depth, g_acc)
        if g_colour is None:
            g_colour = torch.zeros_like(colour)
        g_colour = f32c(g_colour)
        dens, feat = f32c(densities.detach()), f32c(features.detach())
        main = ctx.main_workspace
        deferred = main.deferred
        if deferred is not None and main.buf is not None and dens.data_ptr() == densities.data_ptr() \
                and feat.data_ptr() == features.data_ptr():
            # deferred-gradient mode: the gradient stays in the MAIN workspace's gradient region (kernel layout) for the
            # fused grid step; autograd sees no gradient for the two grid tensors
            # (the region of a buffer this mode has not written yet holds whatever torch.empty returned)
            fresh = (not deferred.dirty) and deferred.clean_ptr != main.buf.data_ptr()
            if not params.linear_grad:   # every render of the step writes the SAME (linear) gradient layout
                params = dataclasses.replace(params, linear_grad=True)
            layout = render_bwd_acc(spec, params, dens, feat, ro, rd, jit, colour, depth, acc, g_colour, g_depth, g_acc,
                                    workspace, ctx.rng, zero_first=fresh,
                                    want_densities=bool(need_d and deferred.want_densities),
                                    want_features=bool(need_f and deferred.want_features),
                                    grad_workspace=(main if workspace is not main else None),
                                    expect_layout=deferred.layout if deferred.dirty else abi.GRAD_ANY)
            if layout is not None:
                deferred.clean_ptr = main.buf.data_ptr()
                if layout != abi.GRAD_ANY:
                    deferred.layout = layout
                    deferred.dirty = True
                workspace.pending = False
                return (None,) * 9
            # (the kernel that fits this render writes another gradient layout than what the region holds -- cannot happen
            #  with linear_grad -- : ordinary path; its un-pack leaves a gradient in ITS workspace's region)
            if workspace is main:
                if deferred.dirty:
                    raise VoxeError("deferred-gradient mode: a render with another gradient layout would overwrite the "
                                    "accumulated gradient")
                deferred.clean_ptr = 0
        d_dens = torch.empty_like(dens) if need_d else None
        d_feat = torch.empty_like(feat) if need_f else None
        render_bwd_into(spec, params, dens, feat, ro, rd, jit, colour, depth, acc, g_colour, g_depth, g_acc,
                        d_dens, d_feat, workspace, ctx.rng)
        workspace.pending = False
        return d_dens, d_feat, None, None, None, None, None, None, None


def render(spec: GridSpec, params: RenderParams, densities: torch.Tensor, features: torch.Tensor,
           rays_o: torch.Tensor, rays_d: torch.Tensor, jitter: Optional[torch.Tensor] = None,
           workspace: Optional[Workspace] = None, rng: Optional[Tuple[int, int]] = None):
    """Fused volumetric render.  Returns (colour [R,Cout], depth [R,1], acc [R,1], disparity [R,1]);
    differentiable w.r.t. `densities` and `features` when they require grad."""
    _validate_inputs(densities, features, rays_o, rays_d, jitter, params)
    if workspace is None:
        workspace = Workspace()
    if rng is None:
        rng = _next_rng() if (params.perturb and jitter is None) else (0, 0)
    if params.dispatch is None:
        # resolve the dispatch HERE, once: the backward runs on autograd's engine thread, where a `dispatch.override()` of the
        # calling context is not visible, and forward and backward of one render must use the same routes
        params = dataclasses.replace(params, dispatch=_dispatch.current())
    return _RenderFn.apply(densities, features, rays_o, rays_d, jitter, spec, params, workspace, rng)


def sample_probe(spec: GridSpec, params: RenderParams, densities, features, rays_o, rays_d, jitter=None,
                 rng=(0, 0), outputs=("idx", "inside", "z", "sigma", "rad")):
    """Per-sample probe (index math test hook): dict of the requested outputs
    idx [R,S,3] int32, inside [R,S] bool, z [R,S], sigma [R,S], rad [R,S,Cout]."""
    _validate_inputs(densities, features, rays_o, rays_d, jitter, params)
    device = densities.device
    ensure_gfx950(device)
    L = lib()
    dens, feat, ro, rd = f32c(densities.detach()), f32c(features.detach()), f32c(rays_o), f32c(rays_d)
    jit = None if jitter is None else f32c(jitter)
    R, S = ro.shape[0], params.num_samples
    cout = 1 if spec.feature_kind == abi.FEAT_ATTN else 3
    g, c = _descs(spec, params, dens, feat, rng[0], rng[1], False)
    shapes = {"idx": ((R, S, 3), torch.int32), "inside": ((R, S), torch.uint8), "z": ((R, S), torch.float32),
              "sigma": ((R, S), torch.float32), "rad": ((R, S, cout), torch.float32)}
    with torch.cuda.device(device):
        ws = torch.empty(L.voxe_workspace_bytes(C.byref(g), C.byref(c), R), dtype=torch.uint8, device=device)
        out = {k: torch.empty(shapes[k][0], dtype=shapes[k][1], device=device) for k in outputs}
        check(L.voxe_sample_probe(C.byref(g), C.byref(c), ptr(ro), ptr(rd), R, ptr(jit), ptr(out.get("idx")),
                                  ptr(out.get("inside")), ptr(out.get("z")), ptr(out.get("sigma")),
                                  ptr(out.get("rad")), ptr(ws), ws.numel(), stream_ptr(device)),
              "voxe_sample_probe")
    if "inside" in out:
        out["inside"] = out["inside"].bool()
    return out


class _QueryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, densities, features, points, spec, workspace):
        device = densities.device
        ensure_gfx950(device)
        L = lib()
        dens, feat, pts = f32c(densities.detach()), f32c(features.detach()), f32c(points.detach())
        N, F = pts.shape[0], feat.shape[-1]
        key = _pack_key(spec, dens, feat)
        g, _ = _descs(spec, RenderParams(1, 0.0, 1.0), dens, feat, 0, 0, False)
        with torch.cuda.device(device):
            ws = workspace.ensure(L.voxe_workspace_bytes(C.byref(g), None, 0), device)
            out = torch.empty((N, F + 1), dtype=torch.float32, device=device)
            check(L.voxe_query_fwd(C.byref(g), ptr(pts), N, ptr(out), int(workspace.key == key), ptr(ws), ws.numel(),
                                   stream_ptr(device)), "voxe_query_fwd")
        workspace.key = key
        ctx.spec, ctx.workspace = spec, workspace
        ctx.save_for_backward(densities, features, pts)
        return out

    @staticmethod
    def backward(ctx, g_out):
        densities, features, pts = ctx.saved_tensors
        need_d, need_f = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_d or need_f):
            return None, None, None, None, None
        device = densities.device
        L = lib()
        dens, feat = f32c(densities.detach()), f32c(features.detach())
        key = _pack_key(ctx.spec, dens, feat)
        g, _ = _descs(ctx.spec, RenderParams(1, 0.0, 1.0), dens, feat, 0, 0, False)
        with torch.cuda.device(device):
            ws = ctx.workspace.ensure(L.voxe_workspace_bytes(C.byref(g), None, 0), device)
            d_dens = torch.empty_like(dens) if need_d else None
            d_feat = torch.empty_like(feat) if need_f else None
            check(L.voxe_query_bwd(C.byref(g), ptr(pts), pts.shape[0], ptr(f32c(g_out)), ptr(d_dens), ptr(d_feat), 0,
                                   int(ctx.workspace.key == key), ptr(ws), ws.numel(), stream_ptr(device)),
                  "voxe_query_bwd")
        ctx.workspace.key = key
        ctx.workspace.state_key = None  # the gradient region of the workspace was reused
        return d_dens, d_feat, None, None, None


def query_points(spec: GridSpec, densities: torch.Tensor, features: torch.Tensor, points: torch.Tensor,
                 workspace: Optional[Workspace] = None) -> torch.Tensor:
    """Un-masked trilinear point query [N,3] -> [N,F+1] = (features, post(pre(density*scale)));
    VoxelGrid.forward semantics (thre3d_atom/thre3d_reprs/voxels.py:287-342), differentiable w.r.t. the grid."""
    for name, t in (("densities", densities), ("features", features), ("points", points)):
        require_device(t, f"query_points ({name})")
    if points.dim() != 2 or points.shape[1] != 3:
        raise VoxeError(f"points must be [N,3]; got {tuple(points.shape)}")
    return _QueryFn.apply(densities, features, points, spec, workspace or Workspace())


def cast_rays(height: int, width: int, focal: float, rotation, translation, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """rays_o, rays_d [H*W,3] on `device` (thre3d_atom/rendering/volumetric/utils/misc.py:12-50)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise VoxeError("cast_rays runs on the GPU only (no CPU fallback in the product path)")
    ensure_gfx950(device)
    rot = torch.as_tensor(rotation).detach().to("cpu", torch.float32).reshape(9).contiguous()
    tr = torch.as_tensor(translation).detach().to("cpu", torch.float32).reshape(3).contiguous()
    fp = C.POINTER(C.c_float)
    n = int(height) * int(width)
    with torch.cuda.device(device):
        ro = torch.empty((n, 3), dtype=torch.float32, device=device)
        rd = torch.empty((n, 3), dtype=torch.float32, device=device)
        check(lib().voxe_cast_rays(int(height), int(width), float(focal), C.cast(rot.data_ptr(), fp),
                                   C.cast(tr.data_ptr(), fp), ptr(ro), ptr(rd), stream_ptr(device)),
              "voxe_cast_rays")
    return ro, rd


def cast_rays_indexed(height: int, width: int, focal: float, poses: torch.Tensor,
                      flat_index: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rays of selected pixels of K cameras: poses [K,3,4] and flat_index int64 [B] = (camera*H + y)*W + x, both on
    the GPU; -> rays_o, rays_d [B,3].  No host synchronisation, no full-image ray buffers."""
    require_device(poses, "cast_rays_indexed")
    require_device(flat_index, "cast_rays_indexed")
    if poses.dim() != 3 or tuple(poses.shape[1:]) != (3, 4) or flat_index.dtype != torch.int64 or flat_index.dim() != 1:
        raise VoxeError("cast_rays_indexed: poses must be [K,3,4] float, flat_index int64 [B]")
    device = poses.device
    ensure_gfx950(device)
    p = f32c(poses)
    idx = flat_index.contiguous()
    n = int(idx.shape[0])
    with torch.cuda.device(device):
        ro = torch.empty((n, 3), dtype=torch.float32, device=device)
        rd = torch.empty((n, 3), dtype=torch.float32, device=device)
        check(lib().voxe_cast_rays_indexed(int(height), int(width), float(focal), ptr(p), int(p.shape[0]), ptr(idx), n,
                                           ptr(ro), ptr(rd), stream_ptr(device)), "voxe_cast_rays_indexed")
    return ro, rd


def random_subset(n: int, count: int, device, rng: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """`count` distinct pseudo-random indices of [0, n) (int64, random order) -- the role of torch.randperm(n)[:count]
    without permuting all n.  Reproducible: (seed, counter) come from torch's CPU generator state like the jitter."""
    device = torch.device(device)
    if device.type != "cuda":
        raise VoxeError("random_subset runs on the GPU only (no CPU fallback in the product path)")
    ensure_gfx950(device)
    seed, offset = rng if rng is not None else _next_rng()
    with torch.cuda.device(device):
        out = torch.empty((int(count),), dtype=torch.int64, device=device)
        check(lib().voxe_random_subset(int(n), int(count), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), ptr(out),
                                       stream_ptr(device)), "voxe_random_subset")
    return out


# ------------------------------------------------------------------------------------------------
# whole-grid passes
# ------------------------------------------------------------------------------------------------
_scratch = {}


def _scratch_for(device, nbytes: int) -> torch.Tensor:
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


class _DclFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sds_density, regular_density):
        require_device(sds_density, "density_correlation_loss")
        a, b = f32c(sds_density.detach()), f32c(regular_density.detach())
        if a.numel() != b.numel():
            raise VoxeError("density_correlation_loss: shape mismatch")
        device = a.device
        ensure_gfx950(device)
        L = lib()
        with torch.cuda.device(device):
            sc = _scratch_for(device, L.voxe_dcl_scratch_bytes(a.numel()))
            loss = torch.empty((), dtype=torch.float32, device=device)
            d_a = torch.empty_like(a) if ctx.needs_input_grad[0] else None
            # gradient for upstream 1.0 is produced in the same launch sequence and scaled in backward
            check(L.voxe_dcl_fwd_bwd(ptr(a), ptr(b), a.numel(), 1.0, ptr(loss), ptr(d_a), 0, ptr(sc),
                                     sc.numel(), stream_ptr(device)), "voxe_dcl_fwd_bwd")
        ctx.save_for_backward(d_a)
        ctx.shape = sds_density.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_a,) = ctx.saved_tensors
        if d_a is None:
            return None, None
        return (d_a * g).reshape(ctx.shape), None


def density_correlation_loss(sds_density: torch.Tensor, regular_density: torch.Tensor) -> torch.Tensor:
    """1 - corr(sds, regular)  (thre3d_atom/modules/sds_trainer.py:507-524); differentiable w.r.t. sds."""
    return _DclFn.apply(sds_density, regular_density)


class _DiffFn(torch.autograd.Function):
    """density_correlation_loss_fn's l2_mode / l1_mode (sds_trainer.py:494-503): value + gradient in one launch sequence"""

    @staticmethod
    def forward(ctx, sds_density, regular_density, kind):
        require_device(sds_density, "density_diff_loss")
        a, b = f32c(sds_density.detach()), f32c(regular_density.detach())
        if a.numel() != b.numel():
            raise VoxeError("density_diff_loss: shape mismatch")
        device = a.device
        ensure_gfx950(device)
        L = lib()
        with torch.cuda.device(device):
            sc = _scratch_for(device, L.voxe_dcl_scratch_bytes(a.numel()))
            loss = torch.empty((), dtype=torch.float32, device=device)
            d_a = torch.empty_like(a) if ctx.needs_input_grad[0] else None
            check(L.voxe_density_diff_fwd_bwd(ptr(a), ptr(b), a.numel(), int(kind), 1.0, ptr(loss), ptr(d_a), 0, ptr(sc),
                                              sc.numel(), stream_ptr(device)), "voxe_density_diff_fwd_bwd")
        ctx.save_for_backward(d_a)
        ctx.shape = sds_density.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_a,) = ctx.saved_tensors
        if d_a is None:
            return None, None, None
        return (d_a * g).reshape(ctx.shape), None, None


def density_diff_loss(sds_density: torch.Tensor, regular_density: torch.Tensor, l2_mode: bool) -> torch.Tensor:
    """mse_loss (l2_mode) / l1_loss of the two density grids (thre3d_atom/modules/sds_trainer.py:494-503); differentiable w.r.t. sds."""
    return _DiffFn.apply(sds_density, regular_density, abi.DREG_L2 if l2_mode else abi.DREG_L1)


class _FeatCorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sds_features, regular_features):
        require_device(sds_features, "feature_correlation_loss")
        f, r = f32c(sds_features.detach()), f32c(regular_features.detach())
        if f.shape != r.shape or f.dim() < 1:
            raise VoxeError("feature_correlation_loss: shape mismatch")
        device = f.device
        ensure_gfx950(device)
        L = lib()
        F = int(f.shape[-1])
        nvox = f.numel() // F
        with torch.cuda.device(device):
            sc = _scratch_for(device, L.voxe_dcl_scratch_bytes(nvox))
            loss = torch.empty((), dtype=torch.float32, device=device)
            d_f = torch.empty_like(f) if ctx.needs_input_grad[0] else None
            check(L.voxe_feature_correlation_fwd_bwd(ptr(f), ptr(r), nvox, F, 1.0, ptr(loss), ptr(d_f), 0, ptr(sc), sc.numel(),
                                                     stream_ptr(device)), "voxe_feature_correlation_fwd_bwd")
        ctx.save_for_backward(d_f)
        ctx.shape = sds_features.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_f,) = ctx.saved_tensors
        if d_f is None:
            return None, None
        return (d_f * g).reshape(ctx.shape), None


def feature_correlation_loss(sds_features: torch.Tensor, regular_features: torch.Tensor) -> torch.Tensor:
    """sum_v (sum_c sigmoid(f_vc) - sigmoid(r_vc))^2 (thre3d_atom/modules/sds_trainer.py:526-534); differentiable w.r.t. sds."""
    return _FeatCorrFn.apply(sds_features, regular_features)


class _TvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid):
        require_device(grid, "tv_loss_on_grid")
        gr = f32c(grid.detach())
        if gr.dim() != 4:
            raise VoxeError("tv_loss_on_grid expects [X,Y,Z,C]")
        device = gr.device
        ensure_gfx950(device)
        L = lib()
        X, Y, Z, Cn = gr.shape
        with torch.cuda.device(device):
            sc = _scratch_for(device, L.voxe_tv_scratch_bytes(X, Y, Z, Cn))
            loss = torch.empty((), dtype=torch.float32, device=device)
            d_g = torch.empty_like(gr) if ctx.needs_input_grad[0] else None
            check(L.voxe_tv_fwd_bwd(ptr(gr), X, Y, Z, Cn, 1.0, ptr(loss), ptr(d_g), 0, ptr(sc), sc.numel(),
                                    stream_ptr(device)), "voxe_tv_fwd_bwd")
        ctx.save_for_backward(d_g)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_g,) = ctx.saved_tensors
        return None if d_g is None else d_g * g


def tv_loss_on_grid(grid: torch.Tensor) -> torch.Tensor:
    """(mean|dx| + mean|dy| + mean|dz|)/3  (thre3d_atom/modules/sds_trainer.py:563-567)."""
    return _TvFn.apply(grid)


@torch.no_grad()
def adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
               step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place torch.optim.Adam update of one tensor (weight_decay=0, amsgrad=False)."""
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        require_device(t, f"adam_step_ ({name})")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != param.numel():
            raise VoxeError(f"adam_step_: {name} must be contiguous float32 of the parameter's size")
    device = param.device
    ensure_gfx950(device)
    with torch.cuda.device(device):
        check(lib().voxe_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(),
                                   float(lr), float(beta1), float(beta2), float(eps), int(step),
                                   stream_ptr(device)), "voxe_adam_step")
    # the library wrote through raw pointers: tell autograd (and the packed-grid cache keyed on
    # Tensor._version) that these tensors changed in place
    for t in (param, exp_avg, exp_avg_sq):
        torch.autograd.graph.increment_version(t)


def render_bwd_acc(spec: GridSpec, params: RenderParams, densities, features, rays_o, rays_d, jitter,
                   colour, depth, acc, g_colour, g_depth, g_acc, workspace: Workspace, rng=(0, 0),
                   zero_first: bool = True, want_densities: bool = True, want_features: bool = True,
                   grad_workspace: Optional[Workspace] = None, expect_layout: int = abi.GRAD_ANY) -> Optional[int]:
    """voxe_render_bwd_acc(_into): the backward of one render, its gradient LEFT in the workspace (kernel layout) for
    `grid_adam_step_` -- in `grad_workspace`'s gradient region when given (a second render of the same step that ran in
    its own workspace).  Returns the layout (abi.GRAD_*); renders accumulated into one step must agree on it:
    with `expect_layout` set, a render whose kernel writes the other layout is NOT run and None is returned."""
    device = densities.device
    L = lib()
    R = rays_o.shape[0]
    key = _pack_key(spec, densities, features)
    g, c = _descs(spec, params, densities, features, rng[0], rng[1], workspace.key == key)
    layout = C.c_int32(abi.GRAD_ANY)
    if expect_layout != abi.GRAD_ANY and R > 0 and expect_layout != predicted_grad_layout(spec, params, densities, features, R):
        return None
    with torch.cuda.device(device):
        ws = workspace.ensure(L.voxe_workspace_bytes(C.byref(g), C.byref(c), R), device)
        c.reuse_packed_grid = int(workspace.key == key)
        c.ray_state_valid = int(workspace.state_key == _state_key(key, params, rays_o, rays_d, jitter, rng, _route(g, c, R)))
        gws = None if grad_workspace is None else grad_workspace.buf
        check(L.voxe_render_bwd_acc_into(C.byref(g), C.byref(c), ptr(rays_o), ptr(rays_d), R, ptr(jitter), ptr(colour),
                                         ptr(depth), ptr(acc), ptr(g_colour), ptr(g_depth), ptr(g_acc),
                                         int(want_densities), int(want_features), int(zero_first), C.byref(layout),
                                         ptr(ws), ws.numel(), ptr(gws), 0 if gws is None else gws.numel(),
                                         stream_ptr(device)), "voxe_render_bwd_acc_into")
    workspace.key = key
    return int(layout.value)


def predicted_grad_layout(spec: GridSpec, params: RenderParams, densities, features, R: int) -> int:
    """which gradient layout the backward of this render writes (voxe_render_bwd_layout)"""
    g, c = _descs(spec, params, densities, features, 0, 0, False)
    return int(lib().voxe_render_bwd_layout(C.byref(g), C.byref(c), int(R)))


def workspace_grad_view(spec: GridSpec, densities, features, workspace: Workspace) -> torch.Tensor:
    """float32 view of the workspace's gradient region (what a data-parallel job all-reduces between
    `render_bwd_acc` and `grid_adam_step_`)."""
    g, _ = _descs(spec, RenderParams(num_samples=1, near=0.0, far=1.0), densities, features, 0, 0, False)
    L = lib()
    off, nbytes = L.voxe_workspace_grad_offset(C.byref(g)), L.voxe_workspace_grad_bytes(C.byref(g))
    if workspace.buf is None or workspace.buf.numel() < off + nbytes:
        raise VoxeError("workspace_grad_view: the workspace holds no gradient region yet")
    return workspace.buf[off:off + nbytes].view(torch.float32)


def workspace_packed_view(spec: GridSpec, densities, features, workspace: Workspace) -> torch.Tensor:
    """float32 view [X*Y*Z*(F+1)] of the workspace's packed grid (what the next render samples): a sharded optimiser
    all-gathers the x-slabs of THIS instead of the raw parameters."""
    n = densities.numel() + features.numel()
    if workspace.buf is None or workspace.buf.numel() < 4 * n:
        raise VoxeError("workspace_packed_view: the workspace holds no packed grid yet")
    return workspace.buf[: 4 * n].view(torch.float32)


@torch.no_grad()
def grid_adam_step_(spec: GridSpec, densities, features, grad_layout: int, workspace: Workspace, step: int, lr: float,
                    state_densities=None, state_features=None, extra_d_densities=None, extra_d_features=None,
                    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                    x_range: Optional[Tuple[int, int]] = None, step_features: Optional[int] = None,
                    dcl_reference: Optional[torch.Tensor] = None, dcl_weight: float = 0.0,
                    dcl_loss: Optional[torch.Tensor] = None, density_kind: int = 0,
                    feat_reference: Optional[torch.Tensor] = None, feat_weight: float = 0.0,
                    feat_loss: Optional[torch.Tensor] = None) -> None:
    """voxe_grid_adam_step: consume the workspace gradient (+ optional extra gradients in tensor layout), update
    densities / features in place with torch.optim.Adam arithmetic, leave the NEW grid packed and a zeroed gradient
    region in the workspace.  state_* = (exp_avg, exp_avg_sq) or None to freeze that tensor.  x_range = (x_begin, x_end)
    restricts the step to a slab of x-planes (sharded optimiser: the caller exchanges the other slabs of the packed grid
    before the next render).  `step` / `step_features`: the 1-based Adam step of the densities / of the features
    (torch.optim.Adam counts per parameter; `step_features` None = the same as `step`).
    `dcl_reference` (densities of the pretrained field, same shape as `densities`): the density-correlation regulariser of the
    SDS edit (modules/sds_trainer.py:507-524) with weight `dcl_weight` is evaluated INSIDE the step -- its moments by two small
    launches on the current parameters, its gradient per voxel in the Adam pass -- and `dcl_loss` (float32 scalar tensor on the
    device) receives the unweighted loss value.  `density_kind` (abi.DREG_*): the same three arguments as the l2_mode / l1_mode
    regulariser (sds_trainer.py:494-503: per-voxel terms, no reduction unless `dcl_loss` is given).  `feat_reference` /
    `feat_weight` / `feat_loss`: _feature_correlation_loss (sds_trainer.py:526-534) against the pretrained field's features,
    evaluated per voxel inside the step (SH-0 / attention grids)."""
    device = densities.device
    ensure_gfx950(device)
    tensors = [("densities", densities, densities), ("features", features, features)]
    for nm, st, ref in (("state_densities", state_densities, densities), ("state_features", state_features, features)):
        if st is not None:
            tensors += [(nm, st[0], ref), (nm, st[1], ref)]
    for nm, t, ref in (("extra_d_densities", extra_d_densities, densities), ("extra_d_features", extra_d_features, features),
                       ("dcl_reference", dcl_reference, densities), ("feat_reference", feat_reference, features)):
        if t is not None:
            tensors.append((nm, t, ref))
    for nm, t, ref in tensors:
        require_device(t, f"grid_adam_step_ ({nm})")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != ref.numel():
            raise VoxeError(f"grid_adam_step_: {nm} must be contiguous float32 of its parameter's size")
    if workspace.buf is None:
        raise VoxeError("grid_adam_step_: the workspace holds no gradient (call render_bwd_acc first)")
    g, _ = _descs(spec, RenderParams(num_samples=1, near=0.0, far=1.0), densities, features, 0, 0, False)
    m_d, v_d = state_densities if state_densities is not None else (None, None)
    m_f, v_f = state_features if state_features is not None else (None, None)
    with torch.cuda.device(device):
        ws = workspace.buf
        x0, x1 = (0, int(densities.shape[0])) if x_range is None else (int(x_range[0]), int(x_range[1]))
        reg = None
        if dcl_reference is not None or feat_reference is not None:
            for nm, t in (("dcl_loss", dcl_loss), ("feat_loss", feat_loss)):
                if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.numel() != 1):
                    raise VoxeError(f"grid_adam_step_: {nm} must be a float32 scalar tensor on the device")
            sc = _scratch_for(device, lib().voxe_dcl_scratch_bytes(densities.numel()))
            reg = abi.VoxeGridRegularisers()
            if dcl_reference is not None:
                reg.dcl_reference, reg.dcl_weight, reg.dcl_loss = ptr(dcl_reference), float(dcl_weight), ptr(dcl_loss)
                reg.density_kind = int(density_kind)
            if feat_reference is not None:
                reg.feat_reference, reg.feat_weight, reg.feat_loss = ptr(feat_reference), float(feat_weight), ptr(feat_loss)
            reg.scratch, reg.scratch_bytes = ptr(sc), sc.numel()
        check(lib().voxe_grid_adam_step(C.byref(g), int(grad_layout), x0, x1, ptr(extra_d_densities), ptr(extra_d_features),
                                        ptr(m_d), ptr(v_d), ptr(m_f), ptr(v_f), float(lr), float(beta1), float(beta2),
                                        float(eps), int(step), int(step if step_features is None else step_features),
                                        None if reg is None else C.byref(reg), ptr(ws), ws.numel(), stream_ptr(device)),
              "voxe_grid_adam_step")
    # (a frozen tensor -- no Adam state -- was not written: its version stays, so other workspaces that hold it packed, e.g. the
    #  two attention grids of the refinement stage over ONE density tensor, keep their packed copies)
    for t in (densities if m_d is not None else None, features if m_f is not None else None, m_d, v_d, m_f, v_f):
        if t is not None:
            torch.autograd.graph.increment_version(t)
    workspace.key = _pack_key(spec, densities, features)   # the workspace holds the updated grid packed
    workspace.state_key = None


@torch.no_grad()
def attn_masked_l1(render: torch.Tensor, attn_map: torch.Tensor):
    """voxe_attn_masked_l1: calc_loss_on_attn_grid (modules/refinement_functions.py:42-77) and its gradient w.r.t. the render.
    Returns (loss [scalar tensor], d_render like `render`)."""
    require_device(render, "attn_masked_l1 (render)")
    require_device(attn_map, "attn_masked_l1 (attn_map)")
    r, m = f32c(render.detach()).reshape(-1), f32c(attn_map.detach()).reshape(-1)
    if r.numel() != m.numel():
        raise VoxeError(f"attn_masked_l1: render ({r.numel()}) and map ({m.numel()}) differ in size")
    device = r.device
    ensure_gfx950(device)
    L = lib()
    d_r = torch.empty_like(r)
    loss = torch.zeros((), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        sc = _scratch_for(device, L.voxe_attn_masked_l1_scratch_bytes())
        check(L.voxe_attn_masked_l1(ptr(r), ptr(m), r.numel(), ptr(d_r), ptr(loss), ptr(sc), sc.numel(), stream_ptr(device)),
              "voxe_attn_masked_l1")
    return loss, d_r.reshape(render.shape)


@torch.no_grad()
def attn_refine_step_(spec: GridSpec, params: RenderParams, densities, attn, rays_o, rays_d, attn_map, workspace: Workspace,
                      step: int, lr: float, state, tv_weight: float, losses: Optional[torch.Tensor] = None, rng=(0, 0),
                      beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, attn_render: Optional[torch.Tensor] = None,
                      zero_gradient_first: bool = False, tv_loss_always: bool = True) -> None:
    """voxe_attn_refine_step: one attention grid's share of a refinement iteration (modules/attn_grid_trainer.py:335-378) in ONE
    library call -- attention render -> masked L1 against `attn_map` + `tv_weight` x TV -> backward -> Adam step of `attn` in place
    (`state` = (exp_avg, exp_avg_sq); the densities are frozen).  `losses` [2] (device) receives masked L1 and TV (unweighted),
    `attn_render` [R] the rendered attention image."""
    device = densities.device
    if spec.feature_kind != abi.FEAT_ATTN:
        raise VoxeError("attn_refine_step_ needs an attention grid spec (feature_kind = FEAT_ATTN)")
    R = int(rays_o.shape[0])
    for nm, t, n in (("densities", densities, densities.numel()), ("attn", attn, densities.numel()),
                     ("exp_avg", state[0], densities.numel()), ("exp_avg_sq", state[1], densities.numel()),
                     ("rays_o", rays_o, 3 * R), ("rays_d", rays_d, 3 * R), ("attn_map", attn_map, R)):
        require_device(t, f"attn_refine_step_ ({nm})")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise VoxeError(f"attn_refine_step_: {nm} must be contiguous float32 with {n} elements")
    for nm, t, n in (("losses", losses, 2), ("attn_render", attn_render, R)):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n):
            raise VoxeError(f"attn_refine_step_: {nm} must be contiguous float32 with {n} elements on the device")
    ensure_gfx950(device)      # (behind the tensor checks: CPU tensors are refused with a VoxeError, GPU or not)
    L = lib()
    key = _pack_key(spec, densities, attn)
    g, c = _descs(spec, params, densities, attn, rng[0], rng[1], workspace.key == key)
    rs = abi.VoxeAttnRefineStep()
    rs.attn_map, rs.tv_weight, rs.tv_loss_always = ptr(attn_map), float(tv_weight), int(bool(tv_loss_always))
    rs.lr, rs.beta1, rs.beta2, rs.eps, rs.step = float(lr), float(beta1), float(beta2), float(eps), int(step)
    rs.exp_avg, rs.exp_avg_sq = ptr(state[0]), ptr(state[1])
    rs.losses, rs.attn_render = ptr(losses), ptr(attn_render)
    with torch.cuda.device(device):
        had = workspace.buf
        ws = workspace.ensure(L.voxe_workspace_bytes(C.byref(g), C.byref(c), R), device)
        # (a buffer this call allocated holds whatever torch.empty returned in its gradient region)
        rs.zero_gradient_first = int(bool(zero_gradient_first) or ws is not had)
        c.reuse_packed_grid = int(workspace.key == key)
        need = L.voxe_attn_refine_scratch_bytes(C.byref(g), R)
        sc = workspace.recon_scratch.get("refine")
        if sc is None or sc.numel() < need or sc.device != ws.device:
            sc = workspace.recon_scratch["refine"] = torch.empty(need, dtype=torch.uint8, device=device)
        check(L.voxe_attn_refine_step(C.byref(g), C.byref(c), C.byref(rs), ptr(rays_o), ptr(rays_d), R, ptr(ws), ws.numel(),
                                      ptr(sc), sc.numel(), stream_ptr(device)), "voxe_attn_refine_step")
    for t in (attn, state[0], state[1]):
        torch.autograd.graph.increment_version(t)
    workspace.key = _pack_key(spec, densities, attn)       # the workspace holds the updated grid packed
    workspace.state_key = None


@torch.no_grad()
def _recon_call(entry: str, spec: GridSpec, params: RenderParams, densities, features, workspace: Workspace, workspace2: Workspace,
                height: int, width: int, focal: float, poses: torch.Tensor, image_rows: Optional[torch.Tensor],
                images: torch.Tensor, batch: int, diffuse_regularisation: bool, state_densities, state_features,
                step_densities: int, step_features: int, lr: float, losses: torch.Tensor, rng: Tuple[int, int],
                beta1: float, beta2: float, eps: float, zero_gradient_first: bool, scratch_holder: Optional[dict]):
    """argument marshalling shared by voxe_recon_step and voxe_recon_prefetch (the hint must announce EXACTLY what the step
    will pass: same descriptors, same buffers, same sizes)"""
    device = densities.device
    ensure_gfx950(device)
    for nm, t in (("densities", densities), ("features", features), ("poses", poses), ("images", images), ("losses", losses)):
        require_device(t, f"{entry} ({nm})")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise VoxeError(f"{entry}: {nm} must be contiguous float32")
    if images.dim() != 4 or images.shape[1] != 3 or tuple(images.shape[2:]) != (height, width):
        raise VoxeError(f"{entry}: images must be [N,3,{height},{width}], got {tuple(images.shape)}")
    if poses.dim() != 3 or tuple(poses.shape[1:]) != (3, 4) or losses.numel() < 4:
        raise VoxeError(f"{entry}: poses must be [K,3,4] and losses hold 4 floats")
    if image_rows is not None and (image_rows.dtype != torch.int64 or image_rows.numel() != poses.shape[0] or not image_rows.is_cuda):
        raise VoxeError(f"{entry}: image_rows must be int64 [K] on the device")
    L = lib()
    key = _pack_key(spec, densities, features)
    p = dataclasses.replace(params, linear_grad=True, image_width=0, image_height=0)
    g, c = _descs(spec, p, densities, features, rng[0], rng[1], workspace.key == key)
    rs = abi.VoxeReconStep()
    rs.H, rs.W, rs.focal = int(height), int(width), float(focal)
    rs.poses, rs.images, rs.image_rows = ptr(poses), ptr(images), ptr(image_rows)
    rs.num_images = int(images.shape[0])
    rs.K, rs.batch, rs.diffuse_regularisation = int(poses.shape[0]), int(batch), int(bool(diffuse_regularisation))
    rs.lr, rs.beta1, rs.beta2, rs.eps = float(lr), float(beta1), float(beta2), float(eps)
    rs.step_densities, rs.step_features = int(step_densities), int(step_features)
    m_d, v_d = state_densities if state_densities is not None else (None, None)
    m_f, v_f = state_features if state_features is not None else (None, None)
    rs.exp_avg_densities, rs.exp_avg_sq_densities = ptr(m_d), ptr(v_d)
    rs.exp_avg_features, rs.exp_avg_sq_features = ptr(m_f), ptr(v_f)
    rs.losses = ptr(losses)
    holder = scratch_holder if scratch_holder is not None else workspace.recon_scratch
    with torch.cuda.device(device):
        nbytes = L.voxe_workspace_bytes(C.byref(g), C.byref(c), int(batch))
        if diffuse_regularisation and p.sh_degree == 0:
            # SH-0: the library runs both renders as ONE launch of 2 * batch rays when the first workspace holds that launch
            nbytes = max(nbytes, L.voxe_workspace_bytes(C.byref(g), C.byref(c), 2 * int(batch)))
        had = workspace.buf
        ws = workspace.ensure(nbytes, device)
        # (a buffer this call allocated holds whatever torch.empty returned in its gradient region)
        rs.zero_gradient_first = int(bool(zero_gradient_first) or ws is not had)
        c.reuse_packed_grid = int(workspace.key == key)
        ws2 = None
        if diffuse_regularisation:
            # the second workspace runs the DIFFUSE render: for view-dependent grids that render may take another route (and
            # need other scratch) than the specular one -- size it with the diffuse cfg, and never below the first workspace
            # (SH-0: it holds the second set of segment tables, the one voxe_recon_prefetch fills ahead)
            _, c2 = _descs(spec, dataclasses.replace(p, render_diffuse=True), densities, features, rng[0], rng[1], False)
            ws2 = workspace2.ensure(max(nbytes, L.voxe_workspace_bytes(C.byref(g), C.byref(c2), int(batch))), device)
        need = L.voxe_recon_scratch_bytes(int(batch))
        sc = holder.get("buf")
        if sc is None or sc.numel() < need or sc.device != ws.device:
            if sc is not None and workspace.prefetch_inflight:
                torch.cuda.synchronize(device)     # (the library's side stream may still be writing the old buffer)
                workspace.prefetch_inflight = False
            sc = holder["buf"] = torch.empty(need, dtype=torch.uint8, device=device)
    return L, g, c, rs, ws, ws2, sc, (m_d, v_d, m_f, v_f)


def _recon_key(spec, params, densities, features, height, width, focal, images, batch, diffuse_regularisation, losses):
    """what has to be unchanged for the descriptors of the last recon_step_ to describe this call too (everything but the cameras,
    the streams, the optimiser's state / hyper-parameters and the flags)"""
    return (spec, params, params.dispatch if params.dispatch is not None else _dispatch.current(), densities.data_ptr(),
            tuple(densities.shape), densities.dtype, densities.is_contiguous(), features.data_ptr(), tuple(features.shape), features.dtype,
            features.is_contiguous(), int(height), int(width), float(focal), images.data_ptr(), tuple(images.shape), images.dtype,
            images.is_contiguous(), int(batch), bool(diffuse_regularisation), losses.data_ptr(), losses.dtype, losses.numel())


def _recon_cameras_ok(poses, image_rows, K) -> bool:
    return (poses.is_cuda and poses.dtype == torch.float32 and poses.is_contiguous() and tuple(poses.shape) == (K, 3, 4)
            and (image_rows is None or (image_rows.is_cuda and image_rows.dtype == torch.int64 and image_rows.numel() == K)))


def recon_step_(spec: GridSpec, params: RenderParams, densities, features, workspace: Workspace, workspace2: Workspace,
                height: int, width: int, focal: float, poses: torch.Tensor, image_rows: Optional[torch.Tensor],
                images: torch.Tensor, batch: int, diffuse_regularisation: bool, state_densities, state_features,
                step_densities: int, step_features: int, lr: float, losses: torch.Tensor, rng: Tuple[int, int],
                beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, zero_gradient_first: bool = True,
                scratch_holder: Optional[dict] = None) -> None:
    """voxe_recon_step: one reconstruction iteration (random pixel batch over the K cameras `poses` -> specular
    [+ diffuse] render -> L1 loss(es) against `images` -> backward -> Adam on both grid tensors) in ONE library call.
    `losses` [4] float32 on the device receives (L1 specular, MSE specular, L1 diffuse, MSE diffuse).  The jitter /
    subset streams derive from `rng` (subset: offset, specular: offset + 1, diffuse: offset + 2).  Afterwards
    `workspace` holds the updated grid packed and a cleared gradient region."""
    device = densities.device
    cache = workspace.recon_cache
    key = _recon_key(spec, params, densities, features, height, width, focal, images, batch, diffuse_regularisation, losses)
    if (cache is not None and scratch_holder is None and cache[0] == key and cache[4] is workspace.buf and workspace.buf is not None
            and cache[5] is workspace2.buf and _recon_cameras_ok(poses, image_rows, cache[3].K)):
        # the same loop as the last call (grid, cameras' shape, images, batch, buffers): its descriptors with this iteration's
        # cameras, streams, optimiser state and flags -- the trainer's loop is paced by this function's host time
        L = lib()
        _, g, c0, rs0, ws, ws2, sc = cache
        c = type(c0).from_buffer_copy(c0)
        rs = type(rs0).from_buffer_copy(rs0)
        c.seed, c.rng_offset = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, int(rng[1]) & 0xFFFFFFFFFFFFFFFF
        c.reuse_packed_grid = int(workspace.key == _pack_key(spec, densities, features))
        rs.poses, rs.image_rows = ptr(poses), ptr(image_rows)
        rs.lr, rs.beta1, rs.beta2, rs.eps = float(lr), float(beta1), float(beta2), float(eps)
        rs.step_densities, rs.step_features = int(step_densities), int(step_features)
        m_d, v_d = state_densities if state_densities is not None else (None, None)
        m_f, v_f = state_features if state_features is not None else (None, None)
        rs.exp_avg_densities, rs.exp_avg_sq_densities = ptr(m_d), ptr(v_d)
        rs.exp_avg_features, rs.exp_avg_sq_features = ptr(m_f), ptr(v_f)
        rs.zero_gradient_first = int(bool(zero_gradient_first))
    else:
        L, g, c, rs, ws, ws2, sc, (m_d, v_d, m_f, v_f) = _recon_call(
            "recon_step_", spec, params, densities, features, workspace, workspace2, height, width, focal, poses, image_rows, images,
            batch, diffuse_regularisation, state_densities, state_features, step_densities, step_features, lr, losses, rng, beta1,
            beta2, eps, zero_gradient_first, scratch_holder)
    with torch.cuda.device(device):
        check(L.voxe_recon_step(C.byref(g), C.byref(c), C.byref(rs), ptr(ws), ws.numel(), ptr(ws2),
                                0 if ws2 is None else ws2.numel(), ptr(sc), sc.numel(), stream_ptr(device)), "voxe_recon_step")
    workspace.prefetch_inflight = False                    # (the step waited for the side stream, hint taken or not)
    workspace2.prefetch_inflight = False
    # what a hint for the NEXT iteration has to repeat (recon_prefetch_ copies these descriptors instead of building them again:
    # the hint's host time is on the iteration's critical path once the device work is hidden)
    workspace.recon_cache = (key, g, c, rs, ws, ws2, sc)
    workspace.prefetch_keepalive = None
    for t in (densities, features, m_d, v_d, m_f, v_f):
        if t is not None:
            torch.autograd.graph.increment_version(t)
    workspace.key = _pack_key(spec, densities, features)   # the workspace holds the updated grid packed
    workspace.state_key = None
    workspace2.key = None                                  # (its packed grid is the previous iteration's)
    workspace2.state_key = None


def recon_prefetch_(spec: GridSpec, params: RenderParams, densities, features, workspace: Workspace, workspace2: Workspace,
                    height: int, width: int, focal: float, poses: torch.Tensor, image_rows: Optional[torch.Tensor],
                    images: torch.Tensor, batch: int, diffuse_regularisation: bool, losses: torch.Tensor, rng: Tuple[int, int],
                    scratch_holder: Optional[dict] = None) -> None:
    """voxe_recon_prefetch: announce the NEXT recon_step_ (same arguments; `poses`, `image_rows` and `rng` are the next
    iteration's; everything else must equal the last recon_step_ of this workspace, otherwise nothing is announced).  The library
    assembles that iteration's batch and segment tables on a stream of its own while the current iteration's backward and grid
    step run; the next recon_step_ takes them iff its arguments are the announced ones.  A hint: results never depend on it, and
    no buffer is ever allocated, grown or moved by it.  Call it right after recon_step_ (both workspaces and the scratch exist by then and keep
    their addresses); `poses` / `image_rows` are kept alive until the next recon_step_ of this workspace."""
    device = densities.device
    if workspace.buf is None:
        return                                             # (no step has sized the buffers yet: nothing to announce against)
    cache = workspace.recon_cache
    key = _recon_key(spec, params, densities, features, height, width, focal, images, batch, diffuse_regularisation, losses)
    if (cache is not None and scratch_holder is None and cache[0] == key and cache[4] is workspace.buf and cache[5] is workspace2.buf
            and _recon_cameras_ok(poses, image_rows, cache[3].K)):
        # the descriptors of the last step with the next iteration's cameras and streams
        L = lib()
        _, g, c0, rs0, ws, ws2, sc = cache
        c = type(c0).from_buffer_copy(c0)
        rs = type(rs0).from_buffer_copy(rs0)
        c.seed, c.rng_offset = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, int(rng[1]) & 0xFFFFFFFFFFFFFFFF
        rs.poses, rs.image_rows = ptr(poses), ptr(image_rows)
    else:
        # Anything else (another batch size, grid, image stack, ... than the last step's) is not announced: marshalling it afresh
        # could regrow a workspace, and a hint must never move a buffer -- the caller's knowledge of what its workspace holds
        # (packed grid, cleared gradient region: `zero_gradient_first`) would silently go stale.  The next step bins inline.
        return
    with torch.cuda.device(device):
        check(L.voxe_recon_prefetch(C.byref(g), C.byref(c), C.byref(rs), ptr(ws), ws.numel(), ptr(ws2),
                                    0 if ws2 is None else ws2.numel(), ptr(sc), sc.numel(), stream_ptr(device)), "voxe_recon_prefetch")
    workspace.prefetch_inflight = True
    if workspace2 is not None:
        workspace2.prefetch_inflight = True
    workspace.prefetch_keepalive = (poses, image_rows, images)


@torch.no_grad()
def upsample_trilinear(src: torch.Tensor, out_size: Sequence[int]) -> torch.Tensor:
    """[X,Y,Z,C] -> [X2,Y2,Z2,C], F.interpolate(trilinear, align_corners=False) semantics
    (thre3d_atom/thre3d_reprs/voxels.py:409-447)."""
    require_device(src, "upsample_trilinear")
    s = f32c(src)
    X, Y, Z, Cn = s.shape
    X2, Y2, Z2 = (int(v) for v in out_size)
    device = s.device
    ensure_gfx950(device)
    with torch.cuda.device(device):
        dst = torch.empty((X2, Y2, Z2, Cn), dtype=torch.float32, device=device)
        check(lib().voxe_upsample_trilinear(ptr(s), X, Y, Z, Cn, ptr(dst), X2, Y2, Z2, stream_ptr(device)),
              "voxe_upsample_trilinear")
    return dst


# ---- refinement stage: grid graph cut / connected components -----------------------------------------------
def graph_build(density_grid: torch.Tensor, feature_grid: torch.Tensor, sigma: float = 0.1,
                dilate_yz: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Nodes and quantised n-link capacities of the refinement graph
    (thre3d_atom/modules/refinement_functions.py:182-287).  density_grid [X,Y,Z(,1)], feature_grid [X,Y,Z,F]
    -> node_mask uint8 [X,Y,Z], cap int32 [6,X,Y,Z] (abi.DIR_* planes, abi.GRAPH_CAP_ONE units)."""
    require_device(density_grid, "graph_build")
    require_device(feature_grid, "graph_build")
    dens = f32c(density_grid)
    feat = f32c(feature_grid)
    X, Y, Z = (int(v) for v in dens.shape[:3])
    if dens.numel() != X * Y * Z or feat.dim() != 4 or tuple(feat.shape[:3]) != (X, Y, Z):
        raise VoxeError(f"graph_build: density grid {tuple(dens.shape)} / feature grid {tuple(feat.shape)} mismatch")
    device = dens.device
    ensure_gfx950(device)
    with torch.cuda.device(device):
        node = torch.empty((X, Y, Z), dtype=torch.uint8, device=device)
        cap = torch.empty((6, X, Y, Z), dtype=torch.int32, device=device)
        check(lib().voxe_graph_build(ptr(dens), ptr(feat), X, Y, Z, int(feat.shape[3]), float(sigma),
                                     int(bool(dilate_yz)), ptr(node), ptr(cap), stream_ptr(device)),
              "voxe_graph_build")
    return node, cap


def graphcut(node_mask: torch.Tensor, terminal: torch.Tensor, cap: torch.Tensor):
    """Exact minimum cut of the voxel graph (g.maxflow() + get_segment, refinement_functions.py:289-294).
    terminal int8 [X,Y,Z]: +1 edit (source) seed, -1 object (sink) seed.  `cap` is not modified.
    -> segment uint8 [X,Y,Z] (0 edit / 1 object / 255 no node), flow value (python int, capacity units)."""
    for t, name in ((node_mask, "node_mask"), (terminal, "terminal"), (cap, "cap")):
        require_device(t, f"graphcut({name})")
    X, Y, Z = (int(v) for v in node_mask.shape)
    if node_mask.dtype != torch.uint8 or terminal.dtype != torch.int8 or cap.dtype != torch.int32:
        raise VoxeError("graphcut: expected uint8 node_mask, int8 terminal, int32 cap")
    if tuple(terminal.shape) != (X, Y, Z) or tuple(cap.shape) != (6, X, Y, Z):
        raise VoxeError("graphcut: shape mismatch")
    device = node_mask.device
    ensure_gfx950(device)
    with torch.cuda.device(device):
        residual = cap.contiguous().clone()
        node, term = node_mask.contiguous(), terminal.contiguous()
        segment = torch.empty((X, Y, Z), dtype=torch.uint8, device=device)
        flow = torch.zeros((1,), dtype=torch.int64, device=device)
        nbytes = int(lib().voxe_graphcut_scratch_bytes(X, Y, Z))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        check(lib().voxe_graphcut(ptr(node), ptr(term), ptr(residual), X, Y, Z, ptr(segment), ptr(flow),
                                  ptr(scratch), nbytes, stream_ptr(device)), "voxe_graphcut")
    return segment, int(flow.item())


def cc_largest_k(mask: torch.Tensor, k: int) -> Tuple[torch.Tensor, int]:
    """cc3d.largest_k(mask, k, connectivity=26) (edit_pretrained_relu_field.py:384-389): int32 labels [X,Y,Z]
    (the M = min(k, N) largest components numbered 1..M by ascending size) and N."""
    require_device(mask, "cc_largest_k")
    m = (mask != 0).to(torch.uint8).contiguous()
    if m.dim() != 3:
        raise VoxeError(f"cc_largest_k: expected a [X,Y,Z] mask, got {tuple(m.shape)}")
    X, Y, Z = (int(v) for v in m.shape)
    device = m.device
    ensure_gfx950(device)
    with torch.cuda.device(device):
        labels = torch.empty((X, Y, Z), dtype=torch.int32, device=device)
        ncomp = torch.zeros((1,), dtype=torch.int32, device=device)
        nbytes = int(lib().voxe_cc_scratch_bytes(X, Y, Z, int(k)))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        check(lib().voxe_cc_largest_k(ptr(m), X, Y, Z, int(k), ptr(labels), ptr(ncomp), ptr(scratch), nbytes,
                                      stream_ptr(device)), "voxe_cc_largest_k")
    return labels, int(ncomp.item())


@torch.no_grad()
def disparity_bwd(depth: torch.Tensor, acc: torch.Tensor, g_disp: torch.Tensor, g_depth: Optional[torch.Tensor] = None,
                  g_acc: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """voxe_disparity_bwd: (g_depth + d disparity/d depth * g_disp, g_acc + d disparity/d acc * g_disp), all [R, 1]"""
    require_device(depth, "disparity_bwd")
    out_d, out_a = torch.empty_like(depth), torch.empty_like(acc)
    with torch.cuda.device(depth.device):
        check(lib().voxe_disparity_bwd(ptr(depth), ptr(acc), ptr(g_disp), ptr(g_depth), ptr(g_acc), ptr(out_d), ptr(out_a),
                                       depth.numel(), stream_ptr(depth.device)), "voxe_disparity_bwd")
    return out_d, out_a


def clock_probe(device, spin: int = 0) -> float:
    """sustained shader clock in Hz (voxe_clock_probe: enqueued on the current stream, then waited for)"""
    hz = C.c_double(0.0)
    with torch.cuda.device(device):
        check(lib().voxe_clock_probe(int(spin), C.byref(hz), stream_ptr(device)), "voxe_clock_probe")
    return float(hz.value)


def region_debug_tables(spec: GridSpec, params: RenderParams, densities, features, R: int, workspace: Workspace) -> dict:
    """views of the space-binned route's segment tables in `workspace` (as the last forward of these R rays left them):
    test aid, see voxe_region_debug_layout"""
    g, c = _descs(spec, params, densities, features, 0, 0, False)
    out = (C.c_int64 * 17)()
    check(lib().voxe_region_debug_layout(C.byref(g), C.byref(c), int(R), out), "voxe_region_debug_layout")
    base, (o_region, o_pos, o_seg, o_sorted, o_lane_n, o_count, o_start, nslots, nlanes, nreg, per_lane, bx, by, bz, ncls,
           chunk) = int(out[0]), [int(v) for v in out[1:]]
    buf = workspace.buf

    def view(off, n, dtype, width=1):
        nbytes = n * width * torch.empty((), dtype=dtype).element_size()
        return buf[base + off: base + off + nbytes].view(dtype).view(n, width) if width > 1 else \
            buf[base + off: base + off + nbytes].view(dtype)

    ncnt = (nreg + 1) * ncls + 1
    return {"slot_region": view(o_region, nslots, torch.int32), "slot_pos": view(o_pos, nslots, torch.int32),
            "slot_seg": view(o_seg, nslots, torch.int32, 2), "sorted": view(o_sorted, nslots, torch.int32, 4),
            "lane_n": view(o_lane_n, nlanes, torch.int32), "count": view(o_count, ncnt, torch.int32),
            "start": view(o_start, ncnt, torch.int32), "nreg": nreg, "slots_per_lane": per_lane,
            "region_cells": (bx, by, bz), "len_classes": ncls, "chunk": chunk}


def profile_enable(on: bool = True) -> None:
    check(lib().voxe_profile_enable(int(on)), "voxe_profile_enable")


def profile_read() -> dict:
    p = abi.VoxeProfile()
    check(lib().voxe_profile_read(C.byref(p)), "voxe_profile_read")
    return {name: getattr(p, name) for name, _ in abi.VoxeProfile._fields_}
