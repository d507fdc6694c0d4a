"""One iteration of the attention-grid refinement loop (BASELINE.json configs[3]; reference modules/attn_grid_trainer.py:335-378)
without the UNet, three ways, same arithmetic:
  A  autograd render + calc_loss_on_attn_grid + tv_loss_on_grid + VoxeAdam      (what bench.py's secondary.refine_iteration times)
  C  the binding's entry points called directly, no autograd: render_fwd_into -> masked-L1 gradient (torch, 6 small ops) ->
     render_bwd_acc -> voxe_tv_fwd_bwd -> grid_adam_step_(extra_d_features)
  L  ONE library call per attention grid: voxe_attn_refine_step (when the library exports it)
    gpurun -- python tools/refine_iter_bench.py [grid side] [image side] [iterations]"""
import ctypes as C
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vox-e_amd"), ROOT]
import torch  # noqa: E402

from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402
from voxe_hip import abi, ops  # noqa: E402
from voxe_hip.runtime import check, lib, ptr, stream_ptr  # noqa: E402
from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, random_grid, synth_pose_angles  # noqa: E402


def timed(fn, iters, warm=5):
    gc.collect()
    gc.disable()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    e = (time.perf_counter() - t) / iters
    gc.enable()
    return e


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    hw = int(sys.argv[2]) if len(sys.argv) > 2 else 266
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    which = sys.argv[4] if len(sys.argv) > 4 else "ACL"
    dev = torch.device("cuda:0")
    dens, _ = (t.to(dev) for t in random_grid(side))
    S = 256
    spec_a = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                          density_post_act=abi.ACT_SOFTPLUS, feature_kind=abi.FEAT_ATTN)
    p_i = pose_spherical(*synth_pose_angles(3, 100), RADIUS)
    ro, rd = ops.cast_rays(hw, hw, focal_for(hw), p_i.rotation, p_i.translation, dev)
    R = ro.shape[0]
    p3 = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw)
    maps = [torch.rand((hw, hw), generator=torch.Generator().manual_seed(47 + i)).to(dev) for i in range(2)]
    tv_w, lr = 0.01, 0.035

    def fresh():
        return [torch.full((side, side, side, 1), -2.0, device=dev) for _ in range(2)]

    res = {}
    if "A" in which:
        from thre3d_atom.modules.optim import VoxeAdam
        from thre3d_atom.modules.refinement_functions import calc_loss_on_attn_grid
        grids = [g_.requires_grad_(True) for g_ in fresh()]
        opts = [VoxeAdam([{"params": [a_], "lr": lr}], betas=(0.9, 0.999)) for a_ in grids]
        wss = [ops.Workspace(), ops.Workspace()]

        def it_a():
            for a_, o_, m_, w_ in zip(grids, opts, maps, wss):
                att, _, _, _ = ops.render(spec_a, p3, dens, a_, ro, rd, workspace=w_)
                loss = calc_loss_on_attn_grid(att, m_) + ops.tv_loss_on_grid(a_) * tv_w
                loss.backward()
                o_.step()
                o_.zero_grad()

        res["A autograd + VoxeAdam"] = timed(it_a, iters)
        res["_A_final"] = [g_.detach().clone() for g_ in grids]
    if "C" in which:
        L = lib()
        grids = fresh()
        states = [(torch.zeros_like(g_), torch.zeros_like(g_)) for g_ in grids]
        wss = [ops.Workspace(), ops.Workspace()]
        outs = [[torch.empty((R, n), dtype=torch.float32, device=dev) for n in (1, 1, 1, 1)] for _ in range(2)]
        tvg = [torch.empty_like(g_) for g_ in grids]
        tvl = torch.zeros((), dtype=torch.float32, device=dev)
        sc = torch.empty(L.voxe_tv_scratch_bytes(side, side, side, 1), dtype=torch.uint8, device=dev)
        st = stream_ptr(dev)
        cnt = [0]

        def it_c():
            cnt[0] += 1
            for a_, s_, m_, w_, o_, t_ in zip(grids, states, maps, wss, outs, tvg):
                rng = (43, cnt[0])
                ops.render_fwd_into(spec_a, p3, dens, a_, ro, rd, None, *o_, w_, rng)
                att = o_[0].view(hw, hw)
                mask = (att > 0.0).float()
                g_att = (torch.sign(att - m_) * mask / mask.sum()).view(R, 1)
                layout = ops.render_bwd_acc(spec_a, p3, dens, a_, ro, rd, None, o_[0], o_[1], o_[2], g_att, None, None, w_, rng,
                                            zero_first=(cnt[0] == 1), want_densities=False)
                check(L.voxe_tv_fwd_bwd(ptr(a_), side, side, side, 1, tv_w, ptr(tvl), ptr(t_), 0, ptr(sc), sc.numel(), st), "tv")
                ops.grid_adam_step_(spec_a, dens, a_, layout, w_, cnt[0], lr, state_features=s_, extra_d_features=t_)

        res["C direct entry points"] = timed(it_c, iters)
    if "L" in which and hasattr(lib(), "voxe_attn_refine_step"):
        grids = fresh()
        states = [(torch.zeros_like(g_), torch.zeros_like(g_)) for g_ in grids]
        wss = [ops.Workspace(), ops.Workspace()]
        losses = torch.zeros((2, 2), dtype=torch.float32, device=dev)
        cnt = [0]

        def it_l():
            cnt[0] += 1
            for i, (a_, s_, m_, w_) in enumerate(zip(grids, states, maps, wss)):
                ops.attn_refine_step_(spec_a, p3, dens, a_, ro, rd, m_, w_, cnt[0], lr, s_, tv_w, losses[i], rng=(43, cnt[0]))

        res["L one library call per grid"] = timed(it_l, iters)
    print(f"# attention-refinement iteration, {side}^3, two {hw}x{hw} attention renders, S = {S}, TV weight {tv_w}, Adam lr {lr}")
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:32s} {1e3 * v:8.4f} ms per iteration   {2 * R / v / 1e6:8.2f} M rendered rays/s")


if __name__ == "__main__":
    main()
