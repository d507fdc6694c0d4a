"""Whole-grid streaming passes of the SDS loop at BASELINE size ("judged directly against 8 TB/s"):
density-correlation loss + gradient, TV loss + gradient, trilinear up-sampling, Adam.
Bytes = what the pass must move once (reads + writes of its operands); time = HIP-event mean over `reps` back-to-back calls
of the C ABI entry point with pre-allocated buffers (so the queue never runs dry: device time, not launch overhead; the
rocprofv3 kernel stats of this script are in profiles/rNN_grid_kernel_stats.csv).
    gpurun -- python tools/grid_pass_bench.py [side]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), ROOT]
import torch  # noqa: E402

from voxe_hip import ops  # noqa: E402
from voxe_hip.runtime import check, lib, ptr, stream_ptr  # noqa: E402
from voxe_hip.workload import random_grid  # noqa: E402

PEAK = 8000.0


def timed(fn, reps=200, warm=10):
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
    L = lib()
    st = stream_ptr(dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    rows = []
    # DCL (modules/sds_trainer.py:507-524): moments pass reads a, b; gradient pass reads a, b, writes d_a
    ref = dens.clone() * 0.9 + 0.05
    d_a = torch.empty_like(dens)
    sc = torch.empty(L.voxe_dcl_scratch_bytes(n), dtype=torch.uint8, device=dev)
    rows.append(("dcl_moments + dcl_grad (loss + d/d sds_density)", 5 * n * 4,
                 timed(lambda: check(L.voxe_dcl_fwd_bwd(ptr(dens), ptr(ref), n, 1.0, ptr(loss), ptr(d_a), 0, ptr(sc), sc.numel(), st), "dcl"))))
    # TV (modules/sds_trainer.py:563-567): reads the grid once (neighbours from cache), writes the gradient
    for name, g in (("tv_kernel on densities / an attention grid [X,Y,Z,1]", dens), ("tv_kernel on features [X,Y,Z,3]", feat)):
        d_g = torch.empty_like(g)
        Cn = int(g.shape[-1])
        sc2 = torch.empty(L.voxe_tv_scratch_bytes(side, side, side, Cn), dtype=torch.uint8, device=dev)
        rows.append((name + " (loss + gradient)", 2 * g.numel() * 4,
                     timed(lambda g=g, d_g=d_g, Cn=Cn, sc2=sc2: check(L.voxe_tv_fwd_bwd(ptr(g), side, side, side, Cn, 1.0, ptr(loss), ptr(d_g), 0,
                                                                                        ptr(sc2), sc2.numel(), st), "tv"))))
    # coarse-to-fine up-sampling (thre3d_reprs/voxels.py:409-447): side/2 -> side
    half = side // 2
    for name, src in (("densities", dens[:half, :half, :half].contiguous()), ("features", feat[:half, :half, :half].contiguous())):
        Cn = int(src.shape[-1])
        dst = torch.empty((side, side, side, Cn), dtype=torch.float32, device=dev)
        rows.append((f"upsample_kernel {name} {half}^3 -> {side}^3", (half ** 3 + n) * Cn * 4,
                     timed(lambda src=src, dst=dst, Cn=Cn: check(L.voxe_upsample_trilinear(ptr(src), half, half, half, Cn, ptr(dst), side, side, side, st), "up"))))
    # Adam on the features tensor (modules/sds_trainer.py:200-203): RMW of param, m, v + read of the gradient
    p, g, m, v = feat.clone(), torch.randn_like(feat) * 1e-3, torch.zeros_like(feat), torch.zeros_like(feat)
    step = [0]

    def adam():
        step[0] += 1
        check(L.voxe_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), 1e-4, 0.9, 0.999, 1e-8, step[0], st), "adam")

    rows.append(("adam_kernel on features", 7 * p.numel() * 4, timed(adam)))
    # the whole-grid work of ONE default SDS iteration (density-correlation regulariser + fused Adam of both tensors):
    # r03 = voxe_dcl_fwd_bwd into a gradient tensor, then voxe_grid_adam_step with it as extra_d_densities (5 launches);
    # r04 = voxe_grid_adam_step with VoxeGridRegularisers (moments + finalize + the step: 3 launches, no gradient tensor)
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0)
    gdesc, _ = ops._descs(spec, ops.RenderParams(num_samples=1, near=0.0, far=1.0), dens, feat, 0, 0, False)
    dd, ff = dens.clone(), feat.clone()
    st_d, st_f = (torch.zeros_like(dd), torch.zeros_like(dd)), (torch.zeros_like(ff), torch.zeros_like(ff))
    ws = ops.Workspace()
    ws.buf = torch.zeros(L.voxe_workspace_bytes(C.byref(gdesc), None, 0), dtype=torch.uint8, device=dev)
    cnt = [0]

    def sds_r03():
        cnt[0] += 1
        check(L.voxe_dcl_fwd_bwd(ptr(dd), ptr(ref), n, 200.0, ptr(loss), ptr(d_a), 0, ptr(sc), sc.numel(), st), "dcl")
        ops.grid_adam_step_(spec, dd, ff, 0, ws, cnt[0], 1e-4, state_densities=st_d, state_features=st_f, extra_d_densities=d_a)

    def sds_r04():
        cnt[0] += 1
        ops.grid_adam_step_(spec, dd, ff, 0, ws, cnt[0], 1e-4, state_densities=st_d, state_features=st_f, dcl_reference=ref,
                            dcl_weight=200.0, dcl_loss=loss)

    def adam_only():
        cnt[0] += 1
        ops.grid_adam_step_(spec, dd, ff, 0, ws, cnt[0], 1e-4, state_densities=st_d, state_features=st_f)

    rows.append(("fused grid step alone (grid_adam_v5_kernel)", 9 * 4 * n * 4, timed(adam_only)))
    rows.append(("SDS iteration, r03: dcl_fwd_bwd -> extra_d -> fused grid step", (9 * 4 + 5 + 1) * n * 4, timed(sds_r03)))
    rows.append(("SDS iteration, r04: DCL inside the fused grid step", (9 * 4 + 2 + 1) * n * 4, timed(sds_r04)))
    print(f"# whole-grid passes at {side}^3 on {torch.cuda.get_device_name(0)}; peak {PEAK:.0f} GB/s (HBM3E spec); device time per call")
    print(f"{'pass':62s} {'MB':>8s} {'ms':>8s} {'GB/s':>8s} {'of peak':>8s}")
    for name, nbytes, ms in rows:
        gbs = nbytes / (ms * 1e-3) / 1e9
        print(f"{name:62s} {nbytes / 1e6:8.1f} {ms:8.4f} {gbs:8.0f} {gbs / PEAK:8.3f}")
    print("# (the fused grid step grid_adam_v5_kernel -- 590 MB -- is timed in the rocprofv3 kernel statistics of bench.py: profiles/rNN_bench_kernel_stats.csv)")


if __name__ == "__main__":
    main()
