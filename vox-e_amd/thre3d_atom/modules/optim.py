"""Fused Adam for the voxel-grid tensors (new in this build).

`torch.optim.Adam(betas=(0.9, 0.999))` is what the reference's trainers construct
(modules/sds_trainer.py:200-203, modules/trainers.py:247-255).  `VoxeAdam` is an Optimizer with the same
state layout (`step`, `exp_avg`, `exp_avg_sq`) and update rule whose `step()` is ONE streaming HIP kernel
per tensor (voxe_adam_step: 7 * n * 4 bytes), so LR schedulers and checkpoint code that expect a
torch Optimizer keep working."""
import weakref
from typing import Iterable

import torch

from voxe_hip import ops as _ops


class VoxeAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state["step"] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _ops.adam_step_(p.data, grad, state["exp_avg"], state["exp_avg_sq"], state["step"],
                                lr=group["lr"], beta1=beta1, beta2=beta2, eps=group["eps"])
                torch.autograd.graph.increment_version(p)
        return loss


class FusedGridAdam(torch.optim.Optimizer):
    """`torch.optim.Adam` for the tensors of ONE VoxelGrid on top of the fused grid step (voxe_render_bwd_acc_into +
    voxe_grid_adam_step): while it is attached, the backward of every render through the grid LEAVES the grid gradient in
    the grid's workspace (kernel layout) and `step()` applies, in ONE streaming pass, the chain rule of the density
    pre-activation, Adam on both tensors, the re-pack of the grid for the next render and the clearing of the gradient
    (bit-identical parameters to un-pack + VoxeAdam per tensor: tests/test_hip_fused_step.py).  Gradients that reach the
    parameters through autograd instead (regularisers: DCL, TV; renders whose kernel writes another gradient layout) sit
    in `.grad` as usual and are added by the same pass.

    The same arithmetic, schedulers and `state_dict` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter) as the
    reference's `torch.optim.Adam(params=[{"params": grid.parameters(), "lr": lr}], betas=(0.9, 0.999))`
    (modules/trainers.py:247-255, modules/sds_trainer.py:200-203, modules/attn_grid_trainer.py:243-247).
    `kind`: "sh" optimises (densities, features), "attn" the attention grid alone (densities frozen).

    Scope of the deferred-gradient mode: it lasts from construction to `detach()`.  While it is on, renders through the
    grid return NO `.grad` for the grid tensors, so it must not outlive its optimiser: use the optimiser as a context
    manager (`with FusedGridAdam(grid, ...) as opt:` -- `detach()` runs on exit, also when the loop raises) or call
    `detach()` in a `finally`; a finalizer clears the mode when the optimiser is garbage collected without either, and
    a second optimiser on a grid whose mode is still on raises instead of silently taking over an accumulated gradient."""

    def __init__(self, voxel_grid, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, kind: str = "sh"):
        if kind not in ("sh", "attn"):
            raise ValueError("kind must be 'sh' or 'attn'")
        self.grid, self.kind = voxel_grid, kind
        if kind == "sh":
            self._dens, self._feat = voxel_grid.densities, voxel_grid.features
            params = [p for p in (self._dens, self._feat) if isinstance(p, torch.nn.Parameter) and p.requires_grad]
        else:
            self._dens, self._feat = voxel_grid.densities, voxel_grid.attn
            params = [self._feat]
        if not params:
            raise ValueError("FusedGridAdam: the grid has no trainable tensors")
        super().__init__([{"params": params}], dict(lr=lr, betas=betas, eps=eps))
        self.spec = voxel_grid.voxe_grid_spec(attn=(kind == "attn"))
        self.workspace = voxel_grid.voxe_workspace(kind)
        train_d = any(p is self._dens for p in params)
        train_f = any(p is self._feat for p in params)
        live = self.workspace.deferred
        if live is not None:
            raise RuntimeError("FusedGridAdam: this grid is already in deferred-gradient mode (another FusedGridAdam is "
                               "attached to it); detach() that optimiser first")
        mine = _ops.DeferredGrad(want_densities=train_d, want_features=train_f)
        self.workspace.deferred = mine
        self._train = (train_d, train_f)
        # dropped without detach() (an exception unwound the training loop, the caller forgot): leave the mode anyway
        self._finalizer = weakref.finalize(self, FusedGridAdam._release, self.workspace, mine)
        self._dcl = None          # (reference densities, weight, kind): density regulariser evaluated inside step()
        self.dcl_loss = None      # device scalar: its unweighted value at the last step()
        self._featcorr = None     # (reference features, weight): feature-correlation regulariser evaluated inside step()
        self.featcorr_loss = None

    @property
    def trains_densities(self) -> bool:
        """the density tensor of the attached grid receives gradients (set_density_correlation needs it)"""
        return bool(self._train[0])

    def set_density_correlation(self, regular_density, weight: float, l2_mode: bool = False, l1_mode: bool = False) -> None:
        """evaluate the SDS edit's density regulariser (modules/sds_trainer.py:494-524: 1 - corr(densities, `regular_density`)
        by default, their mse_loss / l1_loss with `l2_mode` / `l1_mode`; times `weight`) INSIDE step(): no autograd node, no
        [X,Y,Z,1] gradient tensor, no separate gradient kernel; `self.dcl_loss` holds its value (unweighted) after every step().
        `regular_density` None switches it off."""
        if regular_density is None:
            self._dcl, self.dcl_loss = None, None
            return
        if self.kind != "sh" or not self._train[0]:
            raise RuntimeError("set_density_correlation needs trainable densities of an SH grid")
        ref = regular_density.detach().to(self._dens.device, torch.float32).contiguous()
        if ref.numel() != self._dens.numel():
            raise ValueError("regular_density must have the shape of the grid's densities")
        kind = _ops.abi.DREG_L2 if l2_mode else (_ops.abi.DREG_L1 if l1_mode else _ops.abi.DREG_CORRELATION)   # (l2 wins: :498-503)
        self._dcl = (ref, float(weight), kind)
        self.dcl_loss = torch.zeros((), dtype=torch.float32, device=self._dens.device)

    @property
    def trains_features(self) -> bool:
        return bool(self._train[1])

    def set_feature_correlation(self, regular_features, weight: float) -> None:
        """evaluate _feature_correlation_loss (modules/sds_trainer.py:526-534) against `regular_features`, times `weight`, INSIDE
        step() (SH-0 grids: 4-channel texels); `self.featcorr_loss` holds its value (unweighted) after every step().  None
        switches it off."""
        if regular_features is None:
            self._featcorr, self.featcorr_loss = None, None
            return
        if self.kind != "sh" or not self._train[1] or self._feat.shape[-1] > 3:
            raise RuntimeError("set_feature_correlation needs trainable features of an SH degree-0 grid")
        ref = regular_features.detach().to(self._feat.device, torch.float32).contiguous()
        if ref.numel() != self._feat.numel():
            raise ValueError("regular_features must have the shape of the grid's features")
        self._featcorr = (ref, float(weight))
        self.featcorr_loss = torch.zeros((), dtype=torch.float32, device=self._feat.device)

    @staticmethod
    def _release(workspace, mine) -> None:
        if workspace.deferred is mine:
            if mine.dirty:
                workspace.invalidate()
                mine.clean_ptr = 0     # (the region holds an unconsumed gradient: whoever attaches next clears it first)
            workspace.deferred = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.detach()
        return False

    def detach(self) -> None:
        """leave the deferred-gradient mode (renders return ordinary .grad tensors again)"""
        self._finalizer.detach()
        if self.workspace.deferred is not None and self.workspace.deferred.dirty:
            self.workspace.invalidate()   # (an unconsumed gradient would otherwise leak into a later fused step)
            _ops.workspace_grad_view(self.spec, self._dens, self._feat, self.workspace).zero_()
        self.workspace.deferred = None

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none=True)
        d = self.workspace.deferred
        if d is not None and d.dirty:          # a gradient nobody stepped on: clear the region
            _ops.workspace_grad_view(self.spec, self._dens, self._feat, self.workspace).zero_()
            d.dirty, d.layout = False, _ops.abi.GRAD_ANY

    def _state_of(self, p):
        state = self.state[p]
        if not state:
            state["step"] = 0
            state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return state

    @torch.no_grad()
    def reconstruction_step(self, render_params, height: int, width: int, focal: float, poses, image_rows, images,
                            batch: int, diffuse_regularisation: bool, losses, rng) -> None:
        """One whole iteration of the reconstruction loop (modules/trainers.py:288-351) in ONE library call
        (voxe_recon_step): random pixel batch over the cameras `poses` -> specular [+ diffuse] render -> L1 loss(es) ->
        backward -> this optimiser's Adam step.  State, step counters and the current learning rate are this
        optimiser's (schedulers and checkpoints see nothing unusual); `losses` [4] (device) receives L1 / MSE of the
        specular render and L1 / MSE of the diffuse one."""
        if self.kind != "sh":
            raise RuntimeError("reconstruction_step optimises the (densities, features) of an SH grid")
        d = self.workspace.deferred
        if d is None:
            raise RuntimeError("reconstruction_step after detach()")
        if d.dirty:
            raise RuntimeError("reconstruction_step: an accumulated render gradient is waiting for step()")
        group = self.param_groups[0]
        beta1, beta2 = group["betas"]
        train_d, train_f = self._train
        st_d = self._state_of(self._dens) if train_d else None
        st_f = self._state_of(self._feat) if train_f else None
        # (the counters move only once the call went through: a raised VoxeError leaves the optimiser as it was)
        step_d = (st_d["step"] if st_d is not None else st_f["step"]) + 1
        step_f = (st_f["step"] if st_f is not None else step_d - 1) + 1
        ws = self.workspace
        if ws.sibling is None:
            ws.sibling = _ops.Workspace()
        fresh = ws.buf is None or d.clean_ptr != ws.buf.data_ptr()
        try:
            _ops.recon_step_(self.spec, render_params, self._dens, self._feat, ws, ws.sibling, height, width, focal, poses,
                             image_rows, images, batch, diffuse_regularisation,
                             None if st_d is None else (st_d["exp_avg"], st_d["exp_avg_sq"]),
                             None if st_f is None else (st_f["exp_avg"], st_f["exp_avg_sq"]), step_d, step_f, group["lr"],
                             losses, rng, beta1=beta1, beta2=beta2, eps=group["eps"], zero_gradient_first=fresh)
        except Exception:
            # the call may have failed BEHIND its first backward (e.g. the diffuse pass was rejected): the gradient region
            # then holds a partial gradient and the packed grid may be stale -- the next call must clear / re-pack first
            d.clean_ptr = 0
            ws.invalidate()
            raise
        # This path does not go through Optimizer.step(): registered step pre / post hooks are NOT invoked; the scheduler's
        # "optimizer.step() before lr_scheduler.step()" check is told that a step happened.
        self._opt_called = True
        for st in (st_d, st_f):
            if st is not None:
                st["step"] += 1
        d.clean_ptr = ws.buf.data_ptr()
        d.dirty, d.layout = False, _ops.abi.GRAD_ANY

    @torch.no_grad()
    def reconstruction_prefetch(self, render_params, height: int, width: int, focal: float, poses, image_rows, images,
                                batch: int, diffuse_regularisation: bool, losses, rng) -> None:
        """Announce the NEXT reconstruction_step (voxe_recon_prefetch): same arguments, `poses` / `image_rows` / `rng` of the
        iteration to come.  Its batch and segment tables are assembled on a stream of the library's own while the current
        iteration's backward and Adam step run.  A hint -- results never depend on it; call it right after
        reconstruction_step, with `poses` / `image_rows` that were COMPUTED before that step was enqueued (the side stream
        is ordered behind that step's forward, not behind this call)."""
        if self.kind != "sh" or self.workspace.deferred is None or self.workspace.sibling is None:
            return
        _ops.recon_prefetch_(self.spec, render_params, self._dens, self._feat, self.workspace, self.workspace.sibling, height,
                             width, focal, poses, image_rows, images, batch, diffuse_regularisation, losses, rng)

    @torch.no_grad()
    def attention_refinement_step(self, render_params, rays_o, rays_d, attn_map, tv_weight: float, losses=None, rng=(0, 0),
                                  attn_render=None) -> None:
        """One attention grid's share of a refinement iteration (modules/attn_grid_trainer.py:335-378) in ONE library call
        (voxe_attn_refine_step): attention render of the rays -> masked L1 against the UNet's cross-attention map `attn_map`
        [H, W] + `tv_weight` x TV of the attention grid -> backward -> this optimiser's Adam step.  State, step counter and
        the current learning rate are this optimiser's; `losses` [2] (device) receives masked L1 and TV (unweighted),
        `attn_render` [R] the rendered attention image."""
        if self.kind != "attn":
            raise RuntimeError("attention_refinement_step optimises the attention tensor of an attention grid")
        d = self.workspace.deferred
        if d is None:
            raise RuntimeError("attention_refinement_step after detach()")
        if d.dirty:
            raise RuntimeError("attention_refinement_step: an accumulated render gradient is waiting for step()")
        #  the library call does not go through Optimizer.step(): step hooks are not invoked, and an autograd gradient
        # a caller left on the attention tensor (an extra regulariser) would be silently ignored -- refuse it instead; the same
        # for a request the library would only turn down AFTER the error path below has invalidated the workspace
        if self._feat.grad is not None:
            raise RuntimeError("attention_refinement_step: the attention tensor carries an autograd .grad (an extra loss term?): the fused "
                               "call would ignore it -- use render + loss.backward() + step(), or clear the gradient first")
        if getattr(render_params, "deterministic", False):
            raise RuntimeError("attention_refinement_step: voxe_attn_refine_step has no deterministic mode (render_params.deterministic)")
        group = self.param_groups[0]
        beta1, beta2 = group["betas"]
        st = self._state_of(self._feat)
        ws = self.workspace
        fresh = ws.buf is None or d.clean_ptr != ws.buf.data_ptr()
        try:
            _ops.attn_refine_step_(self.spec, render_params, self._dens.detach(), self._feat.detach(), rays_o, rays_d,
                                   attn_map.detach().to(torch.float32).contiguous(), ws, st["step"] + 1, group["lr"],
                                   (st["exp_avg"], st["exp_avg_sq"]), tv_weight, losses, rng, beta1=beta1, beta2=beta2,
                                   eps=group["eps"], attn_render=attn_render, zero_gradient_first=fresh)
        except Exception:
            d.clean_ptr = 0
            ws.invalidate()
            raise
        self._opt_called = True      # (not through Optimizer.step(): see reconstruction_step)
        st["step"] += 1          # (the binding bumped the tensor versions: the detached views share their counters)
        d.clean_ptr = ws.buf.data_ptr()
        d.dirty, d.layout = False, _ops.abi.GRAD_ANY

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        beta1, beta2 = group["betas"]
        d = self.workspace.deferred
        train_d, train_f = self._train
        have_render_grad = d is not None and d.dirty and self.workspace.buf is not None
        if not have_render_grad:
            # no render gradient in the workspace this iteration (e.g. a regulariser-only step): per-tensor Adam
            if self._dcl is not None:     # (the in-step regulariser lives in the fused pass: here it goes through autograd)
                with torch.enable_grad():
                    if self._dcl[2] == _ops.abi.DREG_CORRELATION:
                        dcl = _ops.density_correlation_loss(self._dens, self._dcl[0])
                    else:
                        dcl = _ops.density_diff_loss(self._dens, self._dcl[0], self._dcl[2] == _ops.abi.DREG_L2)
                    (dcl * self._dcl[1]).backward()
                self.dcl_loss.copy_(dcl.detach())
            if self._featcorr is not None:
                with torch.enable_grad():
                    fcl = _ops.feature_correlation_loss(self._feat, self._featcorr[0])
                    (fcl * self._featcorr[1]).backward()
                self.featcorr_loss.copy_(fcl.detach())
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st["step"] += 1
                _ops.adam_step_(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], st["step"], lr=group["lr"],
                                beta1=beta1, beta2=beta2, eps=group["eps"])
                torch.autograd.graph.increment_version(p)
            return loss
        st_d = self._state_of(self._dens) if train_d else None
        st_f = self._state_of(self._feat) if train_f else None
        # torch.optim.Adam counts steps PER PARAMETER: after a step in which only one tensor had a gradient (the per-tensor
        # branch above) the two counters differ, and so do the bias corrections.  (The counters move only once the kernel call
        # went through: a raised VoxeError leaves the optimiser as it was.)
        step_d = (st_d["step"] if st_d is not None else st_f["step"]) + 1
        step_f = (st_f["step"] + 1) if st_f is not None else step_d
        extra_d = self._dens.grad.contiguous() if (train_d and self._dens.grad is not None) else None
        extra_f = self._feat.grad.contiguous() if (train_f and self._feat.grad is not None) else None
        _ops.grid_adam_step_(self.spec, self._dens, self._feat, d.layout, self.workspace, step_d, group["lr"], step_features=step_f,
                             state_densities=None if st_d is None else (st_d["exp_avg"], st_d["exp_avg_sq"]),
                             state_features=None if st_f is None else (st_f["exp_avg"], st_f["exp_avg_sq"]),
                             extra_d_densities=extra_d, extra_d_features=extra_f, beta1=beta1, beta2=beta2,
                             eps=group["eps"], dcl_reference=None if self._dcl is None else self._dcl[0],
                             dcl_weight=0.0 if self._dcl is None else self._dcl[1], dcl_loss=self.dcl_loss,
                             density_kind=0 if self._dcl is None else self._dcl[2],
                             feat_reference=None if self._featcorr is None else self._featcorr[0],
                             feat_weight=0.0 if self._featcorr is None else self._featcorr[1], feat_loss=self.featcorr_loss)
        for st in (st_d, st_f):
            if st is not None:
                st["step"] += 1
        d.dirty, d.layout = False, _ops.abi.GRAD_ANY
        return loss
