"""Ray-sharded data parallelism for the voxel-grid renderer (new in this build; the reference is
single-process / single-GPU: no torch.distributed call anywhere in it).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm).  The grid is
replicated, rays / cameras are split by rank with no collective inside the render, and the voxel-grid
gradient is summed with ONE all-reduce per step over a flat buffer that both parameter tensors view
(65.5 MB at 160^3), followed by the identical optimiser step on every rank (FlatGrid) -- or, with the fused grid
optimiser step, summed by a reduce-scatter over x-slabs, stepped on 1/world of the grid per rank and the PACKED grid
all-gathered for the next render (ShardedGridAdam: the wire bytes of the all-reduce, the optimiser pass / world).
Everything here is plumbing on torch tensors (works on CPU tensors with gloo, which is how it is tested).
"""
from typing import List, Tuple

import torch
import torch.distributed as dist

from thre3d_atom.thre3d_reprs.voxels import VoxelGrid


def world_info() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_items(num_items: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of cameras / images to ranks: rank, rank + world, ..."""
    return list(range(rank, num_items, world))


def shard_rows(height: int, rank: int, world: int, align: int = 8) -> Tuple[int, int]:
    """[row_begin, row_end) band of an image for this rank, aligned to the kernels' 8-row pixel tiles.
    Bands of all ranks partition [0, height)."""
    tiles = (height + align - 1) // align
    lo = (tiles * rank) // world
    hi = (tiles * (rank + 1)) // world
    return min(lo * align, height), min(hi * align, height)


class FlatGrid:
    """Re-homes VoxelGrid._features and ._densities (and their .grad) as views of two flat buffers
    [features | densities] so gradient exchange is a single collective and a fused optimiser can walk
    one contiguous range."""

    def __init__(self, voxel_grid: VoxelGrid):
        feats, dens = voxel_grid.features, voxel_grid.densities
        if not (isinstance(feats, torch.nn.Parameter) and isinstance(dens, torch.nn.Parameter)):
            raise ValueError("FlatGrid needs a tunable VoxelGrid (parameters)")
        nf, nd = feats.numel(), dens.numel()
        self.param = torch.empty(nf + nd, dtype=feats.dtype, device=feats.device)
        self.grad = torch.zeros_like(self.param)
        with torch.no_grad():
            self.param[:nf].copy_(feats.reshape(-1))
            self.param[nf:].copy_(dens.reshape(-1))
            feats.data = self.param[:nf].view(feats.shape)
            dens.data = self.param[nf:].view(dens.shape)
        feats.grad = self.grad[:nf].view(feats.shape)
        dens.grad = self.grad[nf:].view(dens.shape)
        self.voxel_grid = voxel_grid

    def zero_grad(self) -> None:
        self.grad.zero_()

    def all_reduce_grad(self, average: bool = False) -> None:
        """Sum (or average) the gradient over all ranks; no-op without an initialised process group."""
        rank, world = world_info()
        if world == 1:
            return
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        if average:
            self.grad.div_(world)

    def broadcast_param(self, src: int = 0) -> None:
        _, world = world_info()
        if world > 1:
            dist.broadcast(self.param, src=src)


class _GatherRows(torch.autograd.Function):
    """Differentiable all-gather of row bands: forward = all_gather_rows, backward = this rank's band of the
    incoming gradient (every rank evaluates the SAME loss on the SAME full image, so no reduction is needed)."""

    @staticmethod
    def forward(ctx, local, height, align):
        rank, world = world_info()
        ctx.band = shard_rows(height, rank, world, align)
        return all_gather_rows(local.detach(), height, align)

    @staticmethod
    def backward(ctx, grad_full):
        lo, hi = ctx.band
        return grad_full[lo:hi].contiguous(), None, None


def gather_image_rows(local: torch.Tensor, height: int, align: int = 8) -> torch.Tensor:
    """[rows_r, W, C] band of this rank -> full [H, W, C] image on every rank, differentiable (see _GatherRows)."""
    _, world = world_info()
    if world == 1:
        return local
    return _GatherRows.apply(local, height, align)


def all_gather_rows(local: torch.Tensor, height: int, align: int = 8) -> torch.Tensor:
    """Gather per-rank row bands [rows_r, W, C] (bands from shard_rows) into the full [H, W, C] image
    on every rank (used to hand the complete render to the SD UNet in ray-sharded SDS)."""
    rank, world = world_info()
    if world == 1:
        return local
    bands = [shard_rows(height, r, world, align) for r in range(world)]
    # all_gather wants equal shapes: pad every band to the tallest one, trim after the exchange
    tallest = max(hi - lo for lo, hi in bands)
    padded = torch.zeros((tallest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([part[: hi - lo] for part, (lo, hi) in zip(parts, bands)], dim=0)


def slab_of(extent: int, rank: int, world: int, align: int = 1):
    """[begin, end) of rank's EQUAL share of `extent` x-planes in multiples of `align`, or None when it does not divide
    (reduce_scatter_tensor / all_gather_into_tensor need equal sizes; see slabs_of for the general split)."""
    if world < 1 or extent % (world * align) != 0:
        return None
    per = extent // world
    return rank * per, (rank + 1) * per


def slabs_of(extent: int, world: int, align: int = 1) -> List[Tuple[int, int]]:
    """[begin, end) of EVERY rank's share of `extent` x-planes, in units of `align` planes, as even as possible (the
    first `units % world` ranks get one unit more; ranks beyond the number of units get an empty slab).  Equal to
    slab_of() whenever that is defined."""
    units = (extent + align - 1) // align
    base, rem = divmod(units, world)
    out, u = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((min(u * align, extent), min((u + n) * align, extent)))
        u += n
    return out


class ShardedGridAdam:
    """Optimiser step of ONE voxel grid in a data-parallel job, on top of the fused grid step
    (voxe_render_bwd_acc leaves every rank's gradient in its workspace; voxe_grid_adam_step consumes it).

    world == 1 : the plain fused step.  Otherwise one of three exchanges (`exchange=`):
      "reduce-scatter" (default) ZeRO-1 over x-slabs: reduce-scatter of the gradient region (each rank receives the sum
          of ITS slab), fused Adam on that slab only (raw parameters + moments of the other slabs are never touched on
          this rank), all-gather of the packed grid's slabs, in place, so every rank renders the same updated grid.
          Same wire bytes as one all-reduce; the optimiser's HBM pass shrinks by `world`.
      "all-to-all"  the same with both collectives spelled as direct transfers: ONE all-to-all of the gradient slabs
          (point-to-point over every xGMI link at once instead of the library's ring) + a local sum over the `world`
          received slabs, and the packed slabs sent to every peer as one batch of point-to-point operations.  Slabs may
          be UNEVEN: when X is not divisible by the world size (5 planes on 2 ranks, 160 on 7, the odd plane pair of a
          bricked gradient) "reduce-scatter" runs as this exchange too -- no fall-back to a replicated step.
      "all-reduce"  all-reduce of the whole gradient region + the replicated full step.
      "pipelined"   (r06) the direct exchange cut into `chunks` sub-slabs per rank and software-pipelined: the gradient sends
          of ALL chunks are posted at once (asynchronous point-to-point batches: they run back to back on the communication
          stream), and as chunk c's slabs arrive the compute stream sums them, steps Adam on that sub-slab and posts the
          packed sub-slab's sends -- so the local work of a step (sum over `world` slabs, the Adam pass, clearing the foreign
          slabs: ~40 us of kernels at 160^3 on 8 ranks) hides behind the wire time of the following chunks instead of sitting
          between the two collectives, and the packed grid's first chunks travel while the last gradient chunks are still
          being stepped.  Same bytes, same arithmetic per voxel (bit-identical parameters to "all-to-all").  The render itself
          cannot overlap the exchange: the backward produces the whole gradient, the next forward samples the whole grid.
    `autotune()` times the exchanges this job's backend supports (probed once, on all ranks together) on the job's own
    ranks and links (dry steps before training: zero gradient + zero moments leave every parameter bit-unchanged) and
    keeps the fastest -- the same choice on every rank.  `exchange_ms` accumulates the device time of the exchange
    part of every step (events on the launch stream), `exchange_steps` counts them.
    `gather_parameters()` makes the raw tensors whole again on every rank (checkpoints, upsampling between stages).

    `backend` is voxe_hip.ops; tests substitute a CPU stand-in with the same four functions."""

    EXCHANGES = ("reduce-scatter", "all-to-all", "all-reduce", "pipelined")
    _MODE_NAMES = {
        "reduce-scatter": "reduce-scatter + sharded step + all-gather of the packed grid",
        "all-to-all": "all-to-all + local sum + sharded step + point-to-point all-gather of the packed grid",
        "all-to-all-uneven": "all-to-all (uneven slabs) + local sum + sharded step + point-to-point all-gather of the packed grid",
        "all-reduce": "all-reduce + replicated step",
        "pipelined": "pipelined direct exchange (gradient sub-slabs -> local sum + sharded step -> packed sub-slabs, chunk by chunk)",
    }

    def __init__(self, spec, densities: torch.Tensor, features: torch.Tensor, lr: float, betas=(0.9, 0.999),
                 eps: float = 1e-8, train_densities: bool = True, train_features: bool = True, backend=None,
                 exercise_collectives: bool = False, exchange: str = "reduce-scatter", chunks: int = 4):
        if backend is None:
            from voxe_hip import ops as backend
        if exchange not in self.EXCHANGES:
            raise ValueError(f"exchange must be one of {self.EXCHANGES}, got {exchange!r}")
        self.ops, self.spec, self.densities, self.features = backend, spec, densities, features
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state_densities = (torch.zeros_like(densities), torch.zeros_like(densities)) if train_densities else None
        self.state_features = (torch.zeros_like(features), torch.zeros_like(features)) if train_features else None
        self.steps = 0
        self.exercise_collectives = exercise_collectives   # run the collectives even in a 1-rank group (bring-up)
        self.exchange = exchange
        self.chunks = max(1, int(chunks))   # sub-slabs per rank of the "pipelined" exchange
        self.tuned_ms = None        # autotune(): {exchange: milliseconds per dry step, max over ranks}
        self.supported = None       # probe_exchanges(): which exchanges this backend runs (same answer on all ranks)
        self._shard = None
        self._recv = None
        self.mode = "single"
        self._sharded_ran = False
        self._last_slabs = None
        self.exchange_ms, self.exchange_steps = 0.0, 0
        self._events = []           # (start, stop) pairs not read back yet

    # ---- geometry ------------------------------------------------------------------------------------------------
    def _dims(self):
        X, Y, Z = (int(v) for v in self.densities.shape[:3])
        return X, Y, Z, int(self.features.shape[-1]) + 1

    def _slab_table(self, grad_layout: int):
        """per rank: (x_begin, x_end, first float / float count in the gradient region, first / count in the packed grid)"""
        _, world = world_info()
        X, Y, Z, C = self._dims()
        bricked = grad_layout == 1   # VOXE_GRAD_BRICKED
        align = 2 if bricked else 1
        table = []
        for x0, x1 in slabs_of(X, world, align):
            if bricked:
                per_pair = ((Y + 1) // 2) * ((Z + 1) // 2) * 8 * C
                g0, g1 = ((x0 + 1) // 2) * per_pair, ((x1 + 1) // 2) * per_pair   # (x0 is even, or == X for an empty slab)
            else:
                g0, g1 = x0 * Y * Z * C, x1 * Y * Z * C
            table.append((x0, x1, g0, g1 - g0, x0 * Y * Z * C, (x1 - x0) * Y * Z * C))
        return table

    def _collective(self) -> bool:
        _, world = world_info()
        return world > 1 or (self.exercise_collectives and dist.is_initialized())

    def _host_staged(self) -> bool:
        """device tensors over a backend that stages through the host (gloo bring-up runs): such transfers are not
        ordered with the compute stream, so the exchange is fenced by device synchronisations (RCCL is stream-ordered)"""
        return self.densities.is_cuda and dist.is_initialized() and dist.get_backend() != "nccl"

    def _fence(self) -> None:
        if self._host_staged():
            torch.cuda.synchronize(self.densities.device)

    def _mark(self):
        if not self.densities.is_cuda:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.densities.device))
        return ev

    def read_exchange_ms(self) -> float:
        """mean device milliseconds per step spent between the end of the backward and the start of the next render that
        is NOT the local optimiser kernel: gradient exchange + packed-grid exchange (0 for a single process)"""
        for a, b, c, d in self._events:
            d.synchronize()
            self.exchange_ms += a.elapsed_time(b) + c.elapsed_time(d)
        self._events = []
        return self.exchange_ms / max(self.exchange_steps, 1)

    # ---- one exchange + step -------------------------------------------------------------------------------------
    def _run(self, workspace, grad_layout: int, exchange: str, step_no: int, timed: bool = False) -> str:
        """one exchange + optimiser step; returns the name of what ran"""
        rank, world = world_info()
        kw = dict(state_densities=self.state_densities, state_features=self.state_features, beta1=self.betas[0],
                  beta2=self.betas[1], eps=self.eps)
        args = (self.spec, self.densities, self.features, grad_layout, workspace, step_no, self.lr)
        if not self._collective():
            self.ops.grid_adam_step_(*args, **kw)
            return "single"
        region = self.ops.workspace_grad_view(self.spec, self.densities, self.features, workspace)
        table = self._slab_table(grad_layout)
        even = len({t[3] for t in table}) == 1 and len({t[5] for t in table}) == 1 and table[0][3] > 0
        supported = self.probe_exchanges()           # (lazy: the first step of a job that never called autotune())
        if exchange == "pipelined" and supported["pipelined"]:
            return self._run_pipelined(workspace, grad_layout, step_no, timed, region, table, args, kw)
        if not supported[exchange] or (not even and not supported["all-to-all"]):
            # uneven slabs only run as the direct exchange; a backend without all-to-all takes the replicated step
            exchange = "all-reduce"
        e0 = self._mark() if timed else None
        if exchange == "all-reduce":
            self._fence()
            dist.all_reduce(region)
            self._fence()
            e1 = self._mark() if timed else None
            self.ops.grid_adam_step_(*args, **kw)
            if timed and e0 is not None:
                self._events.append((e0, e1, e1, e1))
            return self._MODE_NAMES["all-reduce"]
        x0, x1, g0, gn, p0, pn = table[rank]
        mine = region[g0: g0 + gn]
        self._fence()
        direct = exchange == "all-to-all" or not even
        if direct:
            # slab j of every rank -> rank j (point-to-point over all links at once); sizes may differ between ranks
            if self._recv is None or self._recv.numel() != world * gn:
                self._recv = torch.empty(world * gn, dtype=region.dtype, device=region.device)
            span = table[-1][2] + table[-1][3]
            dist.all_to_all_single(self._recv, region[:span], output_split_sizes=[gn] * world,
                                   input_split_sizes=[t[3] for t in table])
            if gn > 0:
                torch.sum(self._recv.view(world, gn), dim=0, out=mine)
        else:
            if self._shard is None or self._shard.numel() != gn:
                self._shard = torch.empty(gn, dtype=region.dtype, device=region.device)
            dist.reduce_scatter_tensor(self._shard, region[: world * gn])
            mine.copy_(self._shard)
        self._fence()
        e1 = self._mark() if timed else None
        # the step reads (and clears) the gradient in place: the summed slab is where the kernel expects it; clear what
        # this rank's own backward left in the other slabs
        region[:g0].zero_()
        region[g0 + gn:].zero_()
        if x1 > x0:
            self.ops.grid_adam_step_(*args, x_range=(x0, x1), **kw)
        packed = self.ops.workspace_packed_view(self.spec, self.densities, self.features, workspace)
        self._fence()
        e2 = self._mark() if timed else None
        if direct:
            # the all-gather as direct sends too: this rank's packed slab to every peer, theirs into place
            if world > 1:
                ops_ = []
                for peer in range(world):
                    if peer == rank:
                        continue
                    if pn > 0:
                        ops_.append(dist.P2POp(dist.isend, packed[p0: p0 + pn], peer))
                    if table[peer][5] > 0:
                        ops_.append(dist.P2POp(dist.irecv, packed[table[peer][4]: table[peer][4] + table[peer][5]], peer))
                if ops_:
                    for req in dist.batch_isend_irecv(ops_):
                        req.wait()
        else:
            dist.all_gather_into_tensor(packed, packed[p0: p0 + pn])
        self._fence()
        if timed and e0 is not None:
            self._events.append((e0, e1, e2, self._mark()))
        self._sharded_ran = True
        self._last_slabs = [(t[0], t[1]) for t in table]
        if direct and not even:
            return self._MODE_NAMES["all-to-all-uneven"]
        return self._MODE_NAMES["all-to-all" if direct else "reduce-scatter"]

    def _run_pipelined(self, workspace, grad_layout, step_no, timed, region, table, args, kw) -> str:
        """the "pipelined" exchange (class docstring).  Units: a rank's slab [x0, x1) is cut into `chunks` sub-slabs on plane
        (bricked gradient: plane-pair) boundaries; float offsets of a plane boundary in the gradient region / packed grid
        come from the same formulas as _slab_table."""
        rank, world = world_info()
        X, Y, Z, C = self._dims()
        bricked = grad_layout == 1
        align = 2 if bricked else 1
        per_pair = ((Y + 1) // 2) * ((Z + 1) // 2) * 8 * C

        def g_off(x):
            return ((x + 1) // 2) * per_pair if bricked else x * Y * Z * C

        def p_off(x):
            return x * Y * Z * C

        k = self.chunks
        subs = []       # subs[j][c] = (xa, xb) of rank j's sub-slab c
        for (x0, x1, *_rest) in table:
            units = (x1 - x0 + align - 1) // align
            cuts = [x0 + min(((units * c) // k) * align, x1 - x0) for c in range(k)] + [x1]
            subs.append([(cuts[c], cuts[c + 1]) for c in range(k)])
        x0, x1, g0, gn, p0, pn = table[rank]
        mine = region[g0: g0 + gn]
        if self._recv is None or self._recv.numel() != world * gn:
            self._recv = torch.empty(world * gn, dtype=region.dtype, device=region.device)
        recv = self._recv.view(world, gn) if gn > 0 else self._recv.view(world, 0)
        packed = self.ops.workspace_packed_view(self.spec, self.densities, self.features, workspace)
        e0 = self._mark() if timed else None
        self._fence()
        # ---- all gradient chunks are posted at once: chunk c = sub-slab c of EVERY rank's slab, to its owner -----------
        rs_reqs = []
        for c in range(k):
            ops_ = []
            for peer in range(world):
                if peer == rank:
                    continue
                xa, xb = subs[peer][c]
                if xb > xa:
                    ops_.append(dist.P2POp(dist.isend, region[g_off(xa): g_off(xb)], peer))
                ma, mb = subs[rank][c]
                if mb > ma:
                    ops_.append(dist.P2POp(dist.irecv, recv[peer, g_off(ma) - g0: g_off(mb) - g0], peer))
            rs_reqs.append(dist.batch_isend_irecv(ops_) if ops_ else [])
        ag_reqs = []
        for c in range(k):
            for req in rs_reqs[c]:
                req.wait()          # (RCCL: the compute stream waits; host-staged backends: the host does)
            ma, mb = subs[rank][c]
            if mb > ma:
                a, b_ = g_off(ma) - g0, g_off(mb) - g0
                recv[rank, a:b_].copy_(mine[a:b_])
                torch.sum(recv[:, a:b_], dim=0, out=mine[a:b_])
                self.ops.grid_adam_step_(*args, x_range=(ma, mb), **kw)
            self._fence()
            ops_ = []
            for peer in range(world):
                if peer == rank:
                    continue
                if mb > ma:
                    ops_.append(dist.P2POp(dist.isend, packed[p_off(ma): p_off(mb)], peer))
                xa, xb = subs[peer][c]
                if xb > xa:
                    ops_.append(dist.P2POp(dist.irecv, packed[p_off(xa): p_off(xb)], peer))
            ag_reqs.append(dist.batch_isend_irecv(ops_) if ops_ else [])
        # every gradient send has completed (the last chunk's wait above): clear what this rank's backward left in the
        # foreign slabs -- behind the packed chunks already on the wire
        region[:g0].zero_()
        region[g0 + gn:].zero_()
        for reqs in ag_reqs:
            for req in reqs:
                req.wait()
        self._fence()
        if timed and e0 is not None:
            e1 = self._mark()
            self._events.append((e0, e1, e1, e1))     # (the sharded step runs INSIDE the exchange: the whole span is reported)
        self._sharded_ran = True
        self._last_slabs = [(t[0], t[1]) for t in table]
        return self._MODE_NAMES["pipelined"]

    @torch.no_grad()
    def step(self, workspace, grad_layout: int) -> None:
        self.steps += 1
        self.mode = self._run(workspace, grad_layout, self.exchange, self.steps, timed=True)
        self.exchange_steps += 1
        if len(self._events) > 256:
            self.read_exchange_ms()

    @torch.no_grad()
    def probe_exchanges(self):
        """which collectives this job's backend runs, decided ONCE and identically on every rank: each rank tries the tiny
        collective on its own (an unsupported one raises before anything is posted) and the verdicts are combined with a
        MIN all-reduce -- so no rank can later sit in a collective its peers skipped."""
        if self.supported is not None:
            return self.supported
        if not self._collective():
            self.supported = {e: True for e in self.EXCHANGES}
            return self.supported
        _, world = world_info()
        dev = self.densities.device
        probe = torch.zeros(world * 4, dtype=torch.float32, device=dev)
        verdict = []
        for name, call in (
            ("reduce-scatter", lambda: (dist.reduce_scatter_tensor(torch.empty(4, device=dev), probe),
                                        dist.all_gather_into_tensor(probe, probe[:4].clone()))),
            ("all-to-all", lambda: dist.all_to_all_single(torch.empty_like(probe), probe)),
            ("all-reduce", lambda: dist.all_reduce(probe)),
            ("pipelined", lambda: [r.wait() for r in dist.batch_isend_irecv(
                [dist.P2POp(dist.isend, probe[:4], (dist.get_rank() + 1) % world), dist.P2POp(dist.irecv, probe[4:8], (dist.get_rank() - 1) % world)]
                if world > 1 else [])] if world > 1 else None),
        ):
            try:
                call()
                verdict.append(1)
            except (RuntimeError, NotImplementedError):     # the backend lacks it: raised locally, nothing was posted
                verdict.append(0)
        self._fence()
        v = torch.tensor(verdict, dtype=torch.int32, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        self.supported = {e: bool(x) for e, x in zip(self.EXCHANGES, v.tolist())}
        if not self.supported["all-reduce"]:
            raise RuntimeError("the process group cannot even all-reduce")
        if not self.supported[self.exchange]:
            self.exchange = "all-reduce"
        return self.supported

    @torch.no_grad()
    def autotune(self, workspace, grad_layout: int, iters: int = 5, sync=None) -> str:
        """Time every SUPPORTED exchange with `iters` dry steps and keep the fastest (max over ranks, so all ranks agree).
        Call before the first real step, with the workspace holding a packed grid: the gradient region is cleared here,
        and with zero gradient and zero moments a step changes no parameter bit.  `sync`: device synchronisation around
        the timed loops (default: torch.cuda.synchronize when the grid lives on a GPU)."""
        import time

        if self.steps != 0:
            raise RuntimeError("autotune() must run before the first optimiser step")
        if not self._collective():
            return self.exchange
        if sync is None:
            sync = torch.cuda.synchronize if self.densities.is_cuda else (lambda: None)
        supported = self.probe_exchanges()
        self.ops.workspace_grad_view(self.spec, self.densities, self.features, workspace).zero_()
        table = self._slab_table(grad_layout)
        even = len({t[3] for t in table}) == 1 and len({t[5] for t in table}) == 1 and table[0][3] > 0
        times = []
        for exchange in self.EXCHANGES:
            # the same skips on every rank: what the backend lacks (probe_exchanges), and with uneven slabs
            # "reduce-scatter" -- it would run as the very same direct exchange as "all-to-all" (timing it twice says nothing)
            if not supported[exchange] or (not even and exchange == "reduce-scatter") or (exchange == "pipelined" and world_info()[1] < 2):
                times.append(float("inf"))
                continue
            self._run(workspace, grad_layout, exchange, 1)          # buffers, communicator warm-up
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                self._run(workspace, grad_layout, exchange, 1)
            sync()
            times.append((time.perf_counter() - t0) / iters * 1e3)
        t = torch.tensor(times, dtype=torch.float64, device=self.densities.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.tuned_ms = {e: (round(float(v), 4) if v != float("inf") else None) for e, v in zip(self.EXCHANGES, t.tolist())}
        self.exchange = min((e for e in self.EXCHANGES if self.tuned_ms[e] is not None), key=lambda e: self.tuned_ms[e])
        self._sharded_ran = False      # (dry steps changed nothing: the raw tensors are still whole)
        return self.exchange

    @torch.no_grad()
    def gather_parameters(self) -> None:
        """all ranks -> the full raw tensors (a no-op unless a sharded step ran)"""
        rank, world = world_info()
        if not self._sharded_ran:
            return
        slabs = self._last_slabs
        equal = len({b - a for a, b in slabs}) == 1
        for t in (self.densities, self.features):
            flat = t.view(-1)
            per_plane = flat.numel() // int(t.shape[0])
            if equal:
                per = flat.numel() // world
                dist.all_gather_into_tensor(flat, flat[rank * per: (rank + 1) * per].clone())
            else:       # uneven slabs: one broadcast per owner (rare: checkpoints, stage changes)
                for owner, (a, b) in enumerate(slabs):
                    if b > a:
                        dist.broadcast(flat[a * per_plane: b * per_plane], src=owner)
            torch.autograd.graph.increment_version(t)
        self._sharded_ran = False
