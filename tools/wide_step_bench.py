"""Fused grid step (voxe_grid_adam_step) on view-dependent grids: device time per call.   gpurun -- python tools/wide_step_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch  # noqa: E402
from voxe_hip import abi, ops  # noqa: E402

dev = torch.device("cuda:0")
G = 160
spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                    density_post_act=abi.ACT_SOFTPLUS)
for deg in (1, 2, 3):
    F = 3 * (deg + 1) ** 2
    dens = torch.rand((G, G, G, 1), device=dev)
    feat = torch.rand((G, G, G, F), device=dev)
    st_d = (torch.zeros_like(dens), torch.zeros_like(dens))
    st_f = (torch.zeros_like(feat), torch.zeros_like(feat))
    ws = ops.Workspace()
    ws.ensure(4 * 2 * (dens.numel() + feat.numel()) + (1 << 20), dev)
    ws.buf.zero_()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for n in range(1, 4):
        ops.grid_adam_step_(spec, dens, feat, abi.GRAD_LINEAR, ws, n, 1e-3, st_d, st_f)
    ev[0].record()
    K = 10
    for n in range(4, 4 + K):
        ops.grid_adam_step_(spec, dens, feat, abi.GRAD_LINEAR, ws, n, 1e-3, st_d, st_f)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / K
    mb = G ** 3 * (F + 1) * 4 * 9 / 1e6     # grad r+w, param r+w, two moments r+w, packed w
    print(f"SH-{deg} ({F + 1} channels): fused grid step {ms:.3f} ms  ({mb / ms:.0f} GB/s over {mb:.0f} MB)")
