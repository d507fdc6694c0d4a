"""N > 1 path on CPU: 2 gloo processes exercise sharding, the single flat gradient all-reduce, replica
consistency after the optimiser step and the row-band all-gather (no GPU compute involved)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vox-e_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thre3d_atom.modules import parallel
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize

    g = torch.Generator().manual_seed(7)  # same initial grid on every rank
    vg = VoxelGrid(torch.rand(6, 5, 4, 1, generator=g), torch.rand(6, 5, 4, 3, generator=g), VoxelSize(0.5, 0.5, 0.5),
                   tunable=True)
    flat = parallel.FlatGrid(vg)
    assert vg.features.data_ptr() == flat.param.data_ptr() and vg.densities.grad.data_ptr() != 0
    flat.broadcast_param(0)
    # rank-dependent "render gradients" written through the parameter .grad views
    vg.features.grad.fill_(float(rank + 1))
    vg.densities.grad.copy_(torch.arange(vg.densities.numel(), dtype=torch.float32).view_as(vg.densities) * (rank + 1))
    flat.all_reduce_grad()
    tri = world * (world + 1) / 2
    assert torch.all(vg.features.grad == tri)
    assert torch.allclose(vg.densities.grad.reshape(-1), torch.arange(vg.densities.numel(), dtype=torch.float32) * tri)
    opt = torch.optim.Adam([vg.features, vg.densities], lr=0.03)
    opt.step()
    gathered = [torch.empty_like(flat.param) for _ in range(world)]
    dist.all_gather(gathered, flat.param)
    assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged"
    # sharding helpers
    cams = parallel.shard_items(11, rank, world)
    allc = [None] * world
    dist.all_gather_object(allc, cams)
    assert sorted(sum(allc, [])) == list(range(11))
    H, W = 44, 6
    lo, hi = parallel.shard_rows(H, rank, world)
    assert lo % 8 == 0 and (hi % 8 == 0 or hi == H)
    full = torch.arange(H * W * 3, dtype=torch.float32).view(H, W, 3)
    img = parallel.all_gather_rows(full[lo:hi].clone(), H)
    assert torch.equal(img, full)
    # differentiable gather: every rank evaluates the same loss on the full image, gets its band's gradient
    band = full[lo:hi].clone().requires_grad_(True)
    whole = parallel.gather_image_rows(band, H)
    weight = torch.linspace(0.0, 1.0, H * W * 3).view(H, W, 3)
    (whole * weight).sum().backward()
    assert torch.equal(band.grad, weight[lo:hi])
    results[rank] = True
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_and_sharding():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as manager:
        results = manager.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(100)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(results) == {0: True, 1: True}


def test_shard_rows_partition():
    import sys

    from thre3d_atom.modules.parallel import shard_rows

    for H in (1, 7, 8, 9, 400, 401, 800):
        for world in (1, 2, 3, 8):
            bands = [shard_rows(H, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == H
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))


# ------------------------------------------------------------------------------------------------------------------
# ShardedGridAdam (reduce-scatter of the gradient region, fused step on this rank's x-slab, all-gather of the packed
# grid) with a CPU stand-in for the four voxe_hip.ops functions it calls: same layouts (linear / 2x2x2 bricks), same
# contract (the step consumes + clears the gradient slab and writes the packed slab).
# ------------------------------------------------------------------------------------------------------------------
class _CpuWorkspace:
    def __init__(self, X, Y, Z, C):
        self.packed = torch.zeros(X * Y * Z * C)
        nb = ((X + 1) // 2) * ((Y + 1) // 2) * ((Z + 1) // 2) * 8
        self.grad = torch.zeros(nb * C + 16)          # + tail padding like the 256-byte aligned region


def _brick_slots(X, Y, Z):
    x, y, z = torch.meshgrid(torch.arange(X), torch.arange(Y), torch.arange(Z), indexing="ij")
    by, bz = (Y + 1) // 2, (Z + 1) // 2
    return (((x // 2) * by + y // 2) * bz + z // 2) * 8 + (x % 2) * 4 + (y % 2) * 2 + z % 2


class _CpuOps:
    """stand-in backend: packed texel = (features, density), Adam in float32 like torch.optim.Adam"""

    @staticmethod
    def workspace_grad_view(spec, dens, feat, ws):
        return ws.grad

    @staticmethod
    def workspace_packed_view(spec, dens, feat, ws):
        return ws.packed

    @staticmethod
    def store_gradient(ws, g_full, layout):       # g_full [X,Y,Z,C] -> the region in `layout`
        X, Y, Z, C = g_full.shape
        ws.grad.zero_()
        if layout == 1:
            ws.grad[: ws.grad.numel() - 16].view(-1, C)[_brick_slots(X, Y, Z).reshape(-1)] = g_full.reshape(-1, C)
        else:
            ws.grad[: g_full.numel()] = g_full.reshape(-1)

    @staticmethod
    def grid_adam_step_(spec, dens, feat, layout, ws, step, lr, state_densities=None, state_features=None, beta1=0.9,
                        beta2=0.999, eps=1e-8, x_range=None):
        X, Y, Z, F = feat.shape
        C = F + 1
        x0, x1 = x_range if x_range is not None else (0, X)
        if layout == 1:
            slots = _brick_slots(X, Y, Z)[x0:x1].reshape(-1)
            rows = ws.grad[: ws.grad.numel() - 16].view(-1, C)
            g = rows[slots].view(x1 - x0, Y, Z, C).clone()
            rows[slots] = 0.0
        else:
            flat = ws.grad[x0 * Y * Z * C: x1 * Y * Z * C]
            g = flat.view(x1 - x0, Y, Z, C).clone()
            flat.zero_()
        for p, st, gi in ((feat, state_features, g[..., :F]), (dens, state_densities, g[..., F:])):
            if st is None:
                continue
            m, v = st[0][x0:x1], st[1][x0:x1]
            m.lerp_(gi, 1 - beta1)
            v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
            denom = (v.sqrt() / (1 - beta2 ** step) ** 0.5).add_(eps)
            p[x0:x1].addcdiv_(m, denom, value=-lr / (1 - beta1 ** step))
        ws.packed.view(X, Y, Z, C)[x0:x1] = torch.cat((feat[x0:x1], dens[x0:x1]), dim=-1)


def _sharded_worker(rank, world, port, results):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vox-e_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thre3d_atom.modules import parallel

    assert parallel.slab_of(8, 1, 2) == (4, 8) and parallel.slab_of(8, 0, 2, 2) == (0, 4)
    assert parallel.slab_of(6, 0, 2, 2) is None and parallel.slab_of(5, 0, 2) is None
    assert parallel.slabs_of(5, 2) == [(0, 3), (3, 5)] and parallel.slabs_of(6, 2, 2) == [(0, 4), (4, 6)]
    modes = {}
    cases = {"linear": (6, 5, 3, 3, 0), "bricked": (8, 5, 3, 3, 1), "bricked_odd_x": (6, 4, 4, 1, 1), "indivisible": (5, 4, 3, 3, 0)}
    if world > 2:   # (the 8-rank run: planes that do not divide, fewer planes than ranks, an odd bricked extent)
        cases = {"linear": (16, 3, 2, 3, 0), "indivisible": (10, 3, 2, 3, 0), "fewer_planes_than_ranks": (5, 4, 3, 1, 0),
                 "bricked": (16, 4, 2, 3, 1), "bricked_odd_x": (13, 3, 3, 3, 1)}
    for name, (X, Y, Z, F, layout) in cases.items():
        C = F + 1
        gen = torch.Generator().manual_seed(3)
        dens0, feat0 = torch.randn(X, Y, Z, 1, generator=gen), torch.randn(X, Y, Z, F, generator=gen)
        grads = [[torch.randn(X, Y, Z, C, generator=gen) for _ in range(world)] for _ in range(3)]   # [step][rank]
        # reference: one process, summed gradient
        rd, rf, rws = dens0.clone(), feat0.clone(), _CpuWorkspace(X, Y, Z, C)
        rm = ((torch.zeros_like(rd), torch.zeros_like(rd)), (torch.zeros_like(rf), torch.zeros_like(rf)))
        for k, per_rank in enumerate(grads):
            _CpuOps.store_gradient(rws, sum(per_rank[1:], per_rank[0]), layout)
            _CpuOps.grid_adam_step_(None, rd, rf, layout, rws, k + 1, 0.05, rm[0], rm[1])
        # this rank of the sharded job: every exchange must give the single-process result (2 ranks: a + b is exact in
        # any order; more ranks: the gradients are small integers, exact in any order), and so must whatever autotune() picks
        if world > 2:
            grads = [[torch.randint(-8, 9, (X, Y, Z, C), generator=gen).float() for _ in range(world)] for _ in range(3)]
            rd, rf, rws = dens0.clone(), feat0.clone(), _CpuWorkspace(X, Y, Z, C)
            rm = ((torch.zeros_like(rd), torch.zeros_like(rd)), (torch.zeros_like(rf), torch.zeros_like(rf)))
            for k, per_rank in enumerate(grads):
                _CpuOps.store_gradient(rws, sum(per_rank[1:], per_rank[0]), layout)
                _CpuOps.grid_adam_step_(None, rd, rf, layout, rws, k + 1, 0.05, rm[0], rm[1])
        for exchange in ("reduce-scatter", "all-to-all", "all-reduce", "pipelined", "pipelined-3", "auto"):
            d, f, ws = dens0.clone(), feat0.clone(), _CpuWorkspace(X, Y, Z, C)
            # r06: the chunked, software-pipelined direct exchange (4 sub-slabs per rank by default; 3: sub-slabs that do not
            # divide a slab's planes, empty sub-slabs where a slab has fewer planes than chunks)
            opt = parallel.ShardedGridAdam(None, d, f, lr=0.05, backend=_CpuOps, chunks=3 if exchange == "pipelined-3" else 4,
                                           exchange="reduce-scatter" if exchange == "auto" else exchange.split("-3")[0])
            if exchange == "auto":
                ws.packed.copy_(torch.cat((f, d), dim=-1).reshape(-1))
                ws.grad.fill_(123.0)                       # autotune clears the region itself
                picked = opt.autotune(ws, layout, iters=2)
                assert picked in opt.EXCHANGES and set(opt.tuned_ms) == set(opt.EXCHANGES)
                assert torch.equal(d, dens0) and torch.equal(f, feat0) and opt.steps == 0     # dry steps change nothing
                assert float(opt.state_features[0].abs().max()) == 0.0 and float(ws.grad.abs().max()) == 0.0
                chosen = [None] * world
                dist.all_gather_object(chosen, picked)
                assert len(set(chosen)) == 1, "ranks disagree on the exchange"
            for per_rank in grads:
                _CpuOps.store_gradient(ws, per_rank[rank], layout)
                opt.step(ws, layout)
                assert float(ws.grad.abs().max()) == 0.0
            assert torch.equal(ws.packed, rws.packed), (name, exchange)
            opt.gather_parameters()
            assert torch.equal(d, rd) and torch.equal(f, rf), (name, exchange)
            if exchange == "all-to-all" and name in ("linear", "bricked"):
                assert opt.mode.startswith("all-to-all +")
            if exchange in ("all-to-all", "reduce-scatter") and name not in ("linear", "bricked"):
                assert opt.mode.startswith("all-to-all (uneven slabs)")      # no fall-back to a replicated step
            assert opt.exchange_steps == len(grads) and opt.read_exchange_ms() == 0.0      # (CPU tensors: no device timing)
            if exchange == "all-reduce":
                assert opt.mode.startswith("all-reduce")
            if exchange.startswith("pipelined"):
                assert opt.mode.startswith("pipelined direct exchange"), opt.mode
        d, f, ws = dens0.clone(), feat0.clone(), _CpuWorkspace(X, Y, Z, C)
        opt = parallel.ShardedGridAdam(None, d, f, lr=0.05, backend=_CpuOps)
        for per_rank in grads:
            _CpuOps.store_gradient(ws, per_rank[rank], layout)
            opt.step(ws, layout)
            # every rank holds the same, complete packed grid for the next render ...
            assert torch.equal(ws.packed, rws.packed) or per_rank is not grads[-1]
            # ... and a cleared gradient region for the next backward
            assert float(ws.grad.abs().max()) == 0.0
        modes[name] = opt.mode
        assert torch.equal(ws.packed, rws.packed), name
        x0, x1 = parallel.slabs_of(X, world, 2 if layout == 1 else 1)[rank]
        assert torch.equal(d[x0:x1], rd[x0:x1]) and torch.equal(f[x0:x1], rf[x0:x1]), name
        if world > 1:
            other = slice(0, x0) if x0 > 0 else slice(x1, X)
            assert torch.equal(d[other], dens0[other]), "a rank must not touch the raw parameters outside its slab"
        opt.gather_parameters()
        assert torch.equal(d, rd) and torch.equal(f, rf), name
        if name == "indivisible":
            # a backend WITHOUT all-to-all: uneven slabs only run as the direct exchange, so the step takes the
            # replicated all-reduce step instead of crashing in a collective the backend lacks; autotune() times
            # "reduce-scatter" and "all-to-all" only once when they are the same code path
            d, f, ws = dens0.clone(), feat0.clone(), _CpuWorkspace(X, Y, Z, C)
            opt = parallel.ShardedGridAdam(None, d, f, lr=0.05, backend=_CpuOps)
            opt.supported = {"reduce-scatter": True, "all-to-all": False, "all-reduce": True, "pipelined": False}
            for per_rank in grads:
                _CpuOps.store_gradient(ws, per_rank[rank], layout)
                opt.step(ws, layout)
            assert opt.mode.startswith("all-reduce"), opt.mode
            assert torch.equal(ws.packed, rws.packed) and torch.equal(d, rd) and torch.equal(f, rf)
            d, f, ws = dens0.clone(), feat0.clone(), _CpuWorkspace(X, Y, Z, C)
            ws.packed.copy_(torch.cat((f, d), dim=-1).reshape(-1))
            opt = parallel.ShardedGridAdam(None, d, f, lr=0.05, backend=_CpuOps)
            opt.autotune(ws, layout, iters=1)
            assert opt.tuned_ms["reduce-scatter"] is None and opt.tuned_ms["all-to-all"] is not None
    assert modes["linear"].startswith("reduce-scatter") and modes["bricked"].startswith("reduce-scatter")
    assert modes["bricked_odd_x"].startswith("all-to-all (uneven slabs)")    # e.g. 6 x-planes = 3 brick pairs on 2 ranks
    assert modes["indivisible"].startswith("all-to-all (uneven slabs)")
    assert opt.probe_exchanges() == {"reduce-scatter": True, "all-to-all": True, "all-reduce": True, "pipelined": True}
    results[rank] = True
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_grid_adam_ranks(world):
    """world 2 and world 8 (the driver's N = 8 job: x-planes that do not divide, fewer planes than ranks)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as manager:
        results = manager.dict()
        procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(200)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert len(results) == world
