"""Whole-grid streaming passes of the SDS loop at BASELINE size ("judged directly against 8 TB/s"):
density-correlation loss + gradient, TV loss + gradient, trilinear up-sampling, Adam, pack / fused grid step.
Bytes = what the pass must move once (reads + writes of its operands); time = HIP-event mean over `reps` calls through
the C ABI (the autograd wrappers of voxe_hip.ops included).   gpurun -- python tools/grid_pass_bench.py [side]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), ROOT]
import torch  # noqa: E402

from voxe_hip import ops  # noqa: E402
from voxe_hip.workload import random_grid  # noqa: E402

PEAK = 8000.0


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    dev = torch.device("cuda:0")
    dens, feat = (t.to(dev) for t in random_grid(side))
    n = side ** 3
    rows = []
    # DCL (modules/sds_trainer.py:507-524): moments pass reads a, b; gradient pass reads a, b, writes d_a
    a = dens.clone().requires_grad_(True)
    ref = dens.clone() * 0.9 + 0.05
    rows.append(("dcl_moments + dcl_grad (loss + d/d sds_density)", 5 * n * 4, timed(lambda: ops.density_correlation_loss(a, ref))))
    # TV (modules/sds_trainer.py:563-567): reads the grid once (neighbours from cache), writes the gradient
    for name, g in (("tv_kernel on densities [X,Y,Z,1]", dens), ("tv_kernel on features [X,Y,Z,3]", feat)):
        gg = g.clone().requires_grad_(True)
        rows.append((name + " (loss + gradient)", 2 * gg.numel() * 4, timed(lambda gg=gg: ops.tv_loss_on_grid(gg))))
    # coarse-to-fine up-sampling (thre3d_reprs/voxels.py:409-447): side/2 -> side
    half = side // 2
    src_d, src_f = dens[:half, :half, :half].contiguous(), feat[:half, :half, :half].contiguous()
    rows.append((f"upsample_kernel densities {half}^3 -> {side}^3", (half ** 3 + n) * 4, timed(lambda: ops.upsample_trilinear(src_d, (side,) * 3))))
    rows.append((f"upsample_kernel features {half}^3 -> {side}^3", (half ** 3 + n) * 3 * 4, timed(lambda: ops.upsample_trilinear(src_f, (side,) * 3))))
    # Adam on the features tensor (modules/sds_trainer.py:200-203): RMW of param, m, v + read of the gradient
    p, g, m, v = feat.clone(), torch.randn_like(feat) * 1e-3, torch.zeros_like(feat), torch.zeros_like(feat)
    step = [0]

    def adam():
        step[0] += 1
        ops.adam_step_(p, g, m, v, step[0], 1e-4)

    rows.append(("adam_kernel on features", 7 * p.numel() * 4, timed(adam)))
    print(f"# whole-grid passes at {side}^3 on {torch.cuda.get_device_name(0)}; peak {PEAK:.0f} GB/s (HBM3E spec)")
    print(f"{'pass':62s} {'MB':>8s} {'ms':>8s} {'GB/s':>8s} {'of peak':>8s}")
    for name, nbytes, ms in rows:
        gbs = nbytes / (ms * 1e-3) / 1e9
        print(f"{name:62s} {nbytes / 1e6:8.1f} {ms:8.4f} {gbs:8.0f} {gbs / PEAK:8.3f}")
    print("# (the fused grid step grid_adam_kernel<4> -- 590 MB -- is timed by bench.py: roofline.phases_ms / the step breakdown)")


if __name__ == "__main__":
    main()
