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
