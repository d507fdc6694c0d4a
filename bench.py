#!/usr/bin/env python3
"""bench.py -- rendered rays/s (fwd+bwd) of the voxel-grid volumetric render hot path on MI355X.

Workload (BASELINE.json configs[1] / SURVEY.md 8d): 160^3 SH-0 softplus grid (U(-1,1) init, seed 42,
expected_density_scale 100/3), one 400x400 synthetic camera per GPU, S = 256 samples per ray with the
reference's always-on stratified jitter (in-kernel counter hash), white background, upstream gradient
d_colour ~ N(0,1) (seed 43).  One STEP = render forward + render backward through the C ABI (voxe_render_fwd,
voxe_render_bwd_acc: the gradient stays in the workspace), the RCCL exchange when N > 1 (reduce-scatter of the
gradient over x-slabs of the grid, all-gather of the updated packed grid: thre3d_atom/modules/parallel.py) and the
fused Adam update of both grid tensors (voxe_grid_adam_step) -- so the grid really changes every step and nothing is
cached across steps.  Inputs are resident in HBM before timing.  `--optimizer split` = the same arithmetic as
voxe_render_bwd + all-reduce + voxe_adam_step per tensor.

    python bench.py [--gpus N --steps K --warmup W]           (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel, HIP-event timed on the launch
stream through voxe_profile_*) and `cpu_baseline` (the CPU oracle timed on the host cores on a bounded
sample) objects described in DESIGN.md.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "vox-e_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SHADER_CLOCK_HZ = 2.4e9  # MI355X peak engine clock (MI355X_MICROARCH.md); the run MEASURES the sustained clock (voxe_clock_probe)
                         # and quotes the issue-rate ceilings against that; this constant is reported next to it
# Same-host factor between the CPU oracle (kind "port") and the reference's PyTorch CPU path, measured in the 8-vCPU build
# container: oracle 50.5 k rays/s vs reference 5.7 k rays/s at 160^3, 400x400, fwd+bwd, 8 threads
ORACLE_VS_REFERENCE_400 = 8.9
ORACLE_VS_REFERENCE_100 = 2.6
# Untimed clock-settling run before the caller's W warm-up steps: the chip's power state settles over hundreds of
# milliseconds, not over a fixed number of 1 ms steps -- step() runs until this much time has passed (r03's fixed 20 steps
# = 23 ms left the driver's `--steps 20 --warmup 5` line 3 - 5 % below the builder's `--steps 100 --warmup 20`)
PRE_WARM_MS = float(os.environ.get("VOXE_BENCH_PRE_WARM_MS", "400"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)   # (the first ~10 steps run below the sustained clocks)
    ap.add_argument("--grid", type=int, default=160)
    ap.add_argument("--image", type=int, default=400)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--scene", choices=["random", "sphere"], default="random")
    ap.add_argument("--term-eps", type=float, default=0.0, help="gradient truncation (NOT in the reference): the backward stops a ray once T < eps; the forward is always exact; 0 = off")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra 100x100 measurement of the default run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="side of the image the CPU oracle is timed on (0: the full image if a probe says it fits ~30 s)")
    ap.add_argument("--no-adam", action="store_true")
    ap.add_argument("--optimizer", choices=["fused", "split"], default="fused",
                    help="fused: voxe_render_bwd_acc + voxe_grid_adam_step (gradient un-pack, Adam, re-pack and gradient "
                         "clear in one pass over the grid); split: voxe_render_bwd + voxe_adam_step (same arithmetic, bit for bit)")
    ap.add_argument("--no-jitter", action="store_true")
    ap.add_argument("--ray-order", choices=["image", "linear", "random"], default="image",
                    help="image: row-major image rays with the width hint (LDS-window backward); linear: same rays without the "
                         "hint; random: the rays in a random permutation (what a random-ray training batch looks like)")
    ap.add_argument("--camera", type=int, default=None, help="index of the first synthetic camera (of 100) rendered by rank 0 (default 3); "
                                                               "given explicitly it also means --views 1 unless --views says otherwise")
    ap.add_argument("--views", type=int, default=None,
                    help="cameras of the 100-view set the steps cycle through (BASELINE configs[1] is '100 views @ 400x400'): step i renders "
                         "camera (camera + 5 (i mod V)) mod 100 (+ 37 per rank).  Default 20: the driver's --steps 20 renders each once")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: one camera per rank (total work grows with N); strong: ONE camera, every rank renders a band of "
                         "its rows (what a multi-GPU SDS iteration does: one image per step, modules/sds_trainer.py) -- same "
                         "gradient exchange")
    ap.add_argument("--no-gpu-baseline", action="store_true", help="skip timing the PyTorch restatement of the path on this GPU")
    return ap.parse_args()


def main():
    args = parse()
    n_views = args.views if args.views is not None else (1 if args.camera is not None else 20)
    if args.camera is None:
        args.camera = 3
    if n_views < 1:
        raise SystemExit("--views must be >= 1")
    from voxe_hip.workload import FAR, NEAR, RADIUS, focal_for, random_grid, sphere_grid, synth_pose_angles
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import abi, ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # VOXE_BENCH_BACKEND=gloo: bring-up aid for boxes with fewer GPUs than ranks -- the ranks share the visible GPUs and
    # exchange through gloo (host staging); it exercises the N > 1 code with the real kernels, it is NOT a measurement
    backend = os.environ.get("VOXE_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    if world > 1 or os.environ.get("VOXE_BENCH_FORCE_DIST") == "1":  # FORCE: exercise the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    G, HW, S = args.grid, args.image, args.samples
    dens_cpu, feat_cpu = random_grid(G) if args.scene == "random" else sphere_grid(G)
    aabb = ((-1.5, 1.5),) * 3
    spec = ops.GridSpec(aabb=aabb, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                        density_post_act=abi.ACT_SOFTPLUS)
    nvox = G ** 3
    # one flat gradient / parameter buffer: features first, densities last -> ONE all-reduce message
    flat_p = torch.empty(nvox * 4, dtype=torch.float32, device=dev)
    flat_g = torch.zeros_like(flat_p)
    feat = flat_p[: nvox * 3].view(G, G, G, 3)
    dens = flat_p[nvox * 3:].view(G, G, G, 1)
    feat.copy_(feat_cpu)
    dens.copy_(dens_cpu)
    d_feat = flat_g[: nvox * 3].view(G, G, G, 3)
    d_dens = flat_g[nvox * 3:].view(G, G, G, 1)
    exp_avg, exp_avg_sq = torch.zeros_like(flat_p), torch.zeros_like(flat_p)

    # weak scaling: every rank renders its own camera (cameras rank, rank + world, ... of the 100-view set);
    # strong scaling: ONE camera, rank r renders the rows [row_lo, row_hi) (8-row aligned bands: whole pixel tiles)
    strong = args.scaling == "strong"
    if strong and args.ray_order != "image":
        raise SystemExit("--scaling strong shards an IMAGE by rows: use --ray-order image")
    def camera_of(view, rk):
        return (args.camera + 5 * view + (0 if strong else 37 * rk)) % 100

    rows = (0, HW)
    if strong:
        from thre3d_atom.modules.parallel import shard_rows

        rows = shard_rows(HW, rank, world)
    perm = torch.randperm((rows[1] - rows[0]) * HW, generator=torch.Generator().manual_seed(7)).to(dev) if args.ray_order == "random" else None
    view_rays, view_cams = [], []
    for j in range(n_views):
        yaw, pitch = synth_pose_angles(camera_of(j, rank), 100)
        pose_j = pose_spherical(yaw, pitch, RADIUS)
        ro_j, rd_j = ops.cast_rays(HW, HW, focal_for(HW), pose_j.rotation, pose_j.translation, dev)
        if strong:
            ro_j = ro_j[rows[0] * HW: rows[1] * HW].contiguous()
            rd_j = rd_j[rows[0] * HW: rows[1] * HW].contiguous()
        if perm is not None:
            ro_j, rd_j = ro_j[perm].contiguous(), rd_j[perm].contiguous()
        view_rays.append((ro_j, rd_j))
        view_cams.append(camera_of(j, rank))
    pose = pose_spherical(*synth_pose_angles(args.camera, 100), RADIUS)     # (the first view: CPU / GPU baselines, secondary lines)
    rays_o, rays_d = view_rays[0]
    R = rays_o.shape[0]
    R_job = HW * HW if strong else world * R          # rays of the whole job per step
    hint = HW if args.ray_order == "image" else 0
    params = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=not args.no_jitter, white_bkgd=True,
                              term_eps=args.term_eps, image_width=hint)
    gen = torch.Generator().manual_seed(43 + (0 if strong else rank))
    g_colour = torch.randn((HW * HW, 3), generator=gen)[rows[0] * HW: rows[1] * HW].contiguous().to(dev)
    colour = torch.empty((R, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
    acc = torch.empty((R, 1), dtype=torch.float32, device=dev)
    disp = torch.empty((R, 1), dtype=torch.float32, device=dev)
    ws = ops.Workspace()

    # in-AABB sample count of this camera (un-jittered depths), outside the timed region
    probe_params = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, white_bkgd=True, image_width=hint)
    s_in_views = []
    for ro_j, rd_j in view_rays:
        inside = ops.sample_probe(spec, probe_params, dens, feat, ro_j, rd_j, outputs=("inside",))["inside"]
        s_in_views.append(int(inside.sum().item()))
        del inside

    step_no = [0]
    # `--optimizer split`, N > 1: ONE all-reduce of the flat gradient, then the replicated Adam on every rank
    fused = args.optimizer == "fused" and not args.no_adam
    first = [True]
    if fused:
        from thre3d_atom.modules.parallel import ShardedGridAdam

        # N = 1: the fused step.  N > 1: reduce-scatter of the gradient region over x-slabs, the fused step on this
        # rank's slab, all-gather of the packed grid (thre3d_atom/modules/parallel.py)
        # VOXE_GRAD_EXCHANGE = auto (default: time the three exchanges on this job's ranks before the warm-up and keep
        # the fastest) | reduce-scatter | all-to-all | all-reduce
        want_exchange = os.environ.get("VOXE_GRAD_EXCHANGE", "auto")
        opt = ShardedGridAdam(spec, dens, feat, lr=1e-4, exercise_collectives=os.environ.get("VOXE_BENCH_FORCE_DIST") == "1",
                              exchange="reduce-scatter" if want_exchange == "auto" else want_exchange)
        exp_avg = exp_avg_sq = None   # (the split path's moments; the fused optimiser owns its own)

    def fused_step(prm, ro, rd, outs, gcol, wsx, ev=None):
        # the gradient stays in the workspace in the backward kernel's layout; one pass then applies the chain rule of
        # the density pre-activation and Adam to both tensors, writes the packed grid of the next forward and clears
        # the gradient for the next backward.  `ev` (timed steps): events behind the forward and behind the backward
        step_no[0] += 1
        rng = (42, step_no[0])
        ops.render_fwd_into(spec, prm, dens, feat, ro, rd, None, *outs, wsx, rng)
        if ev is not None:
            ev[0].record()
        layout = ops.render_bwd_acc(spec, prm, dens, feat, ro, rd, None, outs[0], outs[1], outs[2], gcol, None, None,
                                    wsx, rng, zero_first=first[0])
        if ev is not None:
            ev[1].record()
        first[0] = False
        opt.step(wsx, layout)

    view_of_step = []     # (which view every step() call rendered, in call order)

    def step(ev=None):
        ro_v, rd_v = view_rays[len(view_of_step) % n_views]
        view_of_step.append(len(view_of_step) % n_views)
        return step_on(ro_v, rd_v, ev)

    def step_on(rays_o, rays_d, ev=None):
        if fused:
            return fused_step(params, rays_o, rays_d, (colour, depth, acc, disp), g_colour, ws, ev)
        step_no[0] += 1
        rng = (42, step_no[0])
        ops.render_fwd_into(spec, params, dens, feat, rays_o, rays_d, None, colour, depth, acc, disp, ws, rng)
        ops.render_bwd_into(spec, params, dens, feat, rays_o, rays_d, None, colour, depth, acc, g_colour, None,
                            None, d_dens, d_feat, ws, rng)
        if dist is not None:
            dist.all_reduce(flat_g)  # sum over ranks (RCCL, xGMI)
        if not args.no_adam:
            ops.adam_step_(flat_p, flat_g, exp_avg, exp_avg_sq, step_no[0], lr=1e-4)
        else:
            ws.invalidate()  # without the optimiser the grid would look unchanged: force the re-pack

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if fused and dist is not None and want_exchange == "auto":
        # outside the timed region: one render so that the workspace exists and holds the packed grid, then dry optimiser
        # steps (zero gradient, zero moments: no parameter bit changes) through each exchange
        ops.render_fwd_into(spec, params, dens, feat, rays_o, rays_d, None, colour, depth, acc, disp, ws, (42, 0))
        layout0 = ops.render_bwd_acc(spec, params, dens, feat, rays_o, rays_d, None, colour, depth, acc, g_colour, None,
                                     None, ws, (42, 0), zero_first=True)
        opt.autotune(ws, layout0)
        first[0] = False                      # autotune left the gradient region cleared
    # untimed, TIME based: step() until PRE_WARM_MS of wall time (device kept busy: the queue never drains between the
    # synchronisations of a 10-step batch) has passed; every rank runs the same number of steps (collectives inside)
    def timed_batch(nsteps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    # Everything that could leave the GPU idle between the clock-settling run and the timed steps happens BEFORE that run
    # (garbage collection, event creation, profiler arming): r04 found the `--steps 20 --warmup 5` line 5 % below the
    # `--steps 100 --warmup 20` line of the same lease because the chip dropped its clocks during the ~50 ms of host work
    # (gc.collect, 1024 hipEventCreate) between the two and was still ramping back up inside the short timed region.
    gc.collect()          # like timeit: no interpreter garbage collection inside the timed steps
    gc.disable()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # Kernel times of the timed steps.  A timing event on the launch stream costs ~5.5 us of device time (kernel timeline of this
    # script, profiles/r06_bench_trace.txt: gaps of 5.9 / 10.6 / 11 us wherever 1 / 2 events sit between two kernels, 0.0 us between
    # kernels with none).  The fused step therefore carries THREE events: the step mark (= start of the forward), one behind the
    # forward (= start of the backward), one behind the backward (= start of the grid step) -- the library's per-phase timers
    # (start + stop per phase: five per step with the mark, 27 us of an 0.82 ms step) stay for the split-optimiser path, whose
    # pack / memset / unpack phases they separate.
    phase_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if fused else None
    ops.profile_enable(True)    # (creates the library's timing events)
    untimed_steps, untimed_s = 0, 0.0
    if PRE_WARM_MS > 0:
        step()                                # (first call: allocations, lazy initialisation)
        untimed_s += timed_batch(10)
        untimed_steps = 11
        per_step = untimed_s / 10
        if dist is not None:
            tt = torch.tensor([per_step], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            per_step = float(tt.item())
        more = int(min(max(PRE_WARM_MS * 1e-3 - untimed_s, 0.0) / max(per_step, 1e-6), 20000))
        while more > 0:
            n = min(more, 200)
            untimed_s += timed_batch(n)
            untimed_steps += n
            more -= n
    for _ in range(args.warmup):
        step()
    if fused:   # exchange timing of the timed region only (events on the launch stream; 0 for one process)
        opt.read_exchange_ms()
        opt.exchange_ms, opt.exchange_steps = 0.0, 0
    ops.profile_enable(not fused)
    barrier()
    del view_of_step[:]         # (the timed steps start at view 0 whatever the warm-up rendered)
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        step(phase_ev[i] if fused else None)
    marks[args.steps].record()
    barrier()
    elapsed = time.perf_counter() - t0
    timed_views = list(view_of_step)
    # in-AABB samples of the launches that were timed (every view as often as it was rendered)
    s_in_total = sum(s_in_views[v] for v in timed_views) / max(len(timed_views), 1)
    gc.enable()
    step_ms_in_order = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    step_ms = sorted(step_ms_in_order)
    exchange_ms = opt.read_exchange_ms() if fused else None
    prof = ops.profile_read()
    ops.profile_enable(False)
    if fused:   # (events of this script, see above: forward = mark -> first event, backward = first -> second)
        prof = dict(prof)
        prof.update(ms_fwd=sum(marks[i].elapsed_time(phase_ev[i][0]) for i in range(args.steps)), n_fwd=args.steps,
                    ms_bwd=sum(phase_ev[i][0].elapsed_time(phase_ev[i][1]) for i in range(args.steps)), n_bwd=args.steps)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: every rank must hold the same packed grid (what the next render samples); the sharded
    # optimiser then makes the raw parameter tensors whole again
    replicas_consistent = None
    if dist is not None and fused:
        packed = ops.workspace_packed_view(spec, dens, feat, ws).double()
        chk = torch.stack([packed.sum(), packed.abs().sum(), (packed * packed).sum()])
        lo_chk, hi_chk = chk.clone(), chk.clone()
        dist.all_reduce(lo_chk, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_chk, op=dist.ReduceOp.MAX)
        replicas_consistent = bool(torch.equal(lo_chk, hi_chk))
        opt.gather_parameters()
        del packed

    # sustained shader clock, measured right behind the timed region (the chip is at its working temperature / power state)
    clock_hz = ops.clock_probe(dev)
    rays_per_s = R_job * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (HIP events on the launch stream, voxe_profile_*) -------------
    ms_fwd = prof["ms_fwd"] / max(prof["n_fwd"], 1)
    ms_bwd = prof["ms_bwd"] / max(prof["n_bwd"], 1)
    # N > 1: every rank's camera, in-AABB samples per ray and render kernel times -- the ranks of a weak-scaling job render
    # DIFFERENT views, and the step costs 157 - 182 M rays/s depending on the view: with these a reader can
    # tell view imbalance from the cost of the exchange
    per_rank = None
    if dist is not None:
        mine = torch.tensor([float(view_cams[0]), s_in_total / max(R, 1), ms_fwd, ms_bwd],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"first_camera": [int(t[0].item()) for t in allr], "in_aabb_samples_per_ray": [round(t[1].item(), 2) for t in allr],
                    "fwd_ms": [round(t[2].item(), 4) for t in allr], "bwd_ms": [round(t[3].item(), 4) for t in allr]}
    # the forward kernel of this launch: image-ordered SH-0 renders march through the LDS texel window (r03) unless switched off
    from voxe_hip import dispatch as _dispatch

    # (r05: the lean tile-ordered kernels of voxe_render_tile4.hip wherever they apply -- the headline configuration included)
    lean = args.ray_order == "image" and _dispatch.current().tile_lean >= 0 and not args.no_jitter
    fwd_kernel = (("voxe::render_fwd_tile4w_kernel<false>" if _dispatch.current().fwd_window >= 0 else "voxe::render_fwd_tile4_kernel<3, false>") if lean else
                  "voxe::render_fwd_tile_kernel" if args.ray_order == "image" and _dispatch.current().fwd_window >= 0
                  else "voxe::render_fwd_seg_kernel<3, 1, 1>")
    # algorithmic bytes: per in-AABB sample 8 corners x 4 ch x 4 B = 128 B read (fwd),
    # 128 B re-read + 128 B gradient scatter (bwd); per ray 24 B rays + outputs/upstream I/O
    bytes_fwd = s_in_total * 128 + R * (24 + 12 + 12)
    bytes_bwd = s_in_total * 256 + R * (24 + 12 + 20)
    if ms_bwd >= ms_fwd:
        bwd_name = (("render_bwd_tile4_kernel<8,false,0>" if lean else "render_bwd_tile_kernel<3,1,1,true,true,0,8,false>")
                    if args.ray_order == "image" else "region_bwd_kernel<3,1>")
        kname, kbytes, kms = bwd_name, bytes_bwd, ms_bwd
    else:
        kname, kbytes, kms = fwd_kernel.replace("voxe::", "").replace(", ", ","), bytes_fwd, ms_fwd
    alg_gbs = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    # ---- what binds the dominant kernel (r06) ------------------------------------------------------------------------------
    # The render kernels work out of the Infinity Cache / L2 / LDS: SURVEY 8(d)'s requested-bytes figure (`alg`) exceeds the HBM
    # peak because corner fetches shared by neighbouring rays never reach HBM.  The top-level `frac` is therefore the fraction of
    # the ceiling that PHYSICALLY binds the kernel (<= 1 by construction), out of three measured candidates:
    #   valu_issue  sum over instruction classes of (dynamic count, PMC) x (clk per wave64 instruction per SIMD, measured by
    #               tools/microbench/valu_rate.hip: profiles/r06_valu_rate.txt) / (1024 SIMDs x measured clock x launch time).
    #               Counts: SQ_INSTS_VALU and its class counters (TRANS_F32, {ADD,MUL,FMA}_F64, CVT, {ADD,MUL,FMA}_F32) per launch
    #               from the PMC summary; price of a class: the static mix of the kernel's sample loop
    #               (tools/isa_issue_model.py -> profiles/r06_issue_model.json: which of its f32 / integer instructions are
    #               double-rate forms without an SGPR operand -- 2.15 clk -- and which are not -- 4.2 clk)
    #   lds_issue   SQ_LDS_IDX_ACTIVE / (256 CUs x clock x time): share of the launch the LDS pipelines are busy
    #   hbm         (2 FETCH_SIZE + WRITE_SIZE) KiB / time against 8 TB/s (MI355X_MICROARCH.md, HBM section)
    # Counters are per launch from the committed PMC summary whose source_hash equals this tree's (tools/gpu_pmc.sh: rocprofv3
    # --pmc in separate passes over THIS script); launch time (HIP events on the launch stream) and shader clock
    # (voxe_clock_probe right behind the timed steps) are this run's.  A summary of other sources is reported as stale, never used.
    from voxe_hip.build import source_hash

    src_hash = source_hash()
    default_cfg = (G, HW, S, args.scene, args.term_eps, args.no_jitter, args.camera, n_views, args.ray_order, strong) == (160, 400, 256, "random", 0.0, False, 3, 20, "image", False)
    pmc, pmc_rel, pmc_stale = None, None, None
    pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))
    if default_cfg and pmc_files:
        pmc_rel = os.path.join("profiles", pmc_files[-1])
        summary = json.load(open(os.path.join(ROOT, pmc_rel)))
        pmc_stale = summary.get("source_hash") != src_hash
        if not pmc_stale:
            pmc = summary["kernels"]
    issue_models = {}
    im_path = os.path.join(ROOT, "profiles", "r06_issue_model.json")
    if os.path.exists(im_path):
        im = json.load(open(im_path))
        if im.get("source_hash") == src_hash:
            issue_models = im["kernels"]

    PMC_CLASSES = ("TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "CVT", "ADD_F32", "MUL_F32", "FMA_F32")

    def issue_model(kernel_key, cnt):
        """VALU issue cycles per launch (all SIMDs together) = sum_class count x clk; None without the inputs"""
        model = issue_models.get(kernel_key)
        if model is None or "SQ_INSTS_VALU" not in cnt:
            return None
        price = model["clk_per_valu_by_pmc_class"]
        total, known, detail = 0.0, 0.0, {}
        for cls in PMC_CLASSES:
            n = cnt.get("SQ_INSTS_VALU_" + cls)
            if n is None:
                continue
            c = price.get(cls, 8.1 if cls == "TRANS_F32" else 4.2)
            total += n * c
            known += n
            detail[cls] = {"count": n, "clk": c}
        rest = max(cnt["SQ_INSTS_VALU"] - known, 0.0)
        c_rest = price.get("OTHER", model["clk_per_valu"]) if known > 0 else model["clk_per_valu"]
        total += rest * c_rest
        detail["OTHER" if known > 0 else "ALL (no class counters in the summary: static mean of the sample loop)"] = {"count": rest, "clk": c_rest}
        return {"cycles": total, "clk_per_valu": total / max(cnt["SQ_INSTS_VALU"], 1.0), "classes": detail}

    def physical_of(prefix, accept, launch_ms):
        """ceilings of one kernel: counters per launch from the hash-matched PMC summary, launch time and clock from THIS run"""
        if pmc_stale:
            return {"stale": True, "reason": f"{pmc_rel} was collected on other kernel sources (source_hash differs from "
                                             f"{src_hash}): re-run tools/gpu_pmc.sh + tools/pmc_to_json.py"}
        if pmc is None or launch_ms <= 0:
            return None
        keys = [k for k in pmc if k.startswith(prefix) and accept(k)]
        cnt = pmc[keys[0]] if keys else {}
        if not all(k in cnt for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE")):
            return None
        t = launch_ms * 1e-3
        hbm_bytes = int((2.0 * cnt["FETCH_SIZE"] + cnt["WRITE_SIZE"]) * 1024)
        im_ = issue_model(keys[0], cnt)
        valu_cycles = im_["cycles"] if im_ else cnt["SQ_INSTS_VALU"] * 4.2     # (no static model: every instruction at the single rate -- an upper bound)
        ceil = {"valu_issue": valu_cycles / (1024 * clock_hz * t), "lds_issue": cnt["SQ_LDS_IDX_ACTIVE"] / (256 * clock_hz * t),
                "hbm": hbm_bytes / t / 1e9 / HBM_PEAK_GBS}
        binding = max(ceil, key=ceil.get)
        return {
            "kernel": keys[0], "launch_ms": round(launch_ms, 4),
            "valu_issue_frac": round(ceil["valu_issue"], 4), "lds_issue_frac": round(ceil["lds_issue"], 4),
            "hbm_frac_measured": round(ceil["hbm"], 4), "traffic_bytes": hbm_bytes,
            "binding": binding, "binding_frac": round(ceil[binding], 4),
            "valu_issue_cycles_per_launch": round(valu_cycles, 0), "valu_clk_per_instruction": round(im_["clk_per_valu"], 3) if im_ else 4.2,
            "valu_issue_model": ({k: {"count": round(v["count"], 0), "clk": v["clk"]} for k, v in im_["classes"].items()} if im_ else
                                 "no static model for these sources (python tools/isa_issue_model.py --write): 4.2 clk per instruction"),
            # the SQ's own busy counter next to the model (quad-cycles per MI355X_MICROARCH.md: x 4)
            "sq_active_inst_valu_frac": (round(cnt["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * clock_hz * t), 4) if "SQ_ACTIVE_INST_VALU" in cnt else None),
            "resident_waves_per_simd": model_occupancy.get(keys[0]),
            "counters": {k: cnt[k] for k in ("SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU",
                                             "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "FETCH_SIZE", "WRITE_SIZE") if k in cnt},
        }

    model_occupancy = {k: v.get("occupancy_waves_per_simd") for k, v in issue_models.items()}
    phys_bwd = physical_of("voxe::render_bwd_tile4_kernel<8, false, 0>", lambda k: True, ms_bwd) if lean else None
    phys_fwd = physical_of(fwd_kernel, lambda k: True, ms_fwd)
    physical = phys_bwd if ms_bwd >= ms_fwd else phys_fwd
    usable = bool(physical) and not physical.get("stale")
    traffic = physical.get("traffic_bytes") if usable else None
    traffic_src = (f"{pmc_rel} (rocprofv3 --pmc, per launch, mean over the {n_views} views; (2*FETCH_SIZE + WRITE_SIZE) KiB; source_hash {src_hash}), "
                   f"kernel {physical['kernel']}") if traffic else None
    if usable:
        physical = dict(physical)
        physical["compulsory_bytes"] = int(2 * nvox * 4 * 4)       # read the grid once + write the gradient once
        physical["traffic_over_compulsory"] = round(traffic / (2 * nvox * 4 * 4), 2)
    binding = physical.get("binding") if usable else None
    # top level: the binding ceiling in ITS units.  valu_issue / lds_issue: issue cycles used per second against the cycles the
    # chip has (1024 SIMDs resp. 256 LDS pipelines x measured clock); hbm: measured traffic against 8 TB/s.  Without a usable PMC
    # summary (other configurations than the headline one, stale sources) only the requested-bytes figure can be stated: `frac`
    # is then null rather than a number above 1.
    if binding == "valu_issue":
        units, r_peak = "G issue-clk/s (1024 SIMDs)", 1024 * clock_hz / 1e9
        r_ach = physical["valu_issue_frac"] * r_peak
    elif binding == "lds_issue":
        units, r_peak = "G issue-clk/s (256 LDS pipelines)", 256 * clock_hz / 1e9
        r_ach = physical["lds_issue_frac"] * r_peak
    elif binding == "hbm":
        units, r_peak = "GB/s", HBM_PEAK_GBS
        r_ach = physical["hbm_frac_measured"] * r_peak
    else:
        units, r_peak, r_ach = "GB/s", HBM_PEAK_GBS, None
    roofline = {
        "bound": binding or "unknown (no PMC summary for this configuration / these sources: see `alg`)",
        "achieved": round(r_ach, 2) if r_ach is not None else None, "peak": round(r_peak, 2), "unit": units,
        "frac": round(r_ach / r_peak, 4) if r_ach is not None else None,
        "frac_is": "utilisation of the ceiling that binds the dominant kernel (<= 1): see `physical` for all three candidates and "
                   "`alg` for SURVEY 8(d)'s requested-bytes throughput in HBM units",
        "traffic": traffic, "traffic_source": traffic_src,
        "kernel": kname, "launch_ms": round(kms, 4),
        # SURVEY.md 8(d)'s REQUESTED-bytes model (every trilinear corner fetch / scatter counted): a throughput in HBM units that
        # exceeds the HBM peak for these cache-resident kernels -- kept because the survey's contract quotes it
        "alg": {"bytes_per_launch": int(kbytes), "gbs": round(alg_gbs, 2), "frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
                "note": "256 B per in-AABB sample + 56 B per ray (backward) / 128 B + 48 B (forward); corner fetches shared by neighbouring "
                        "rays are served from L1 / L2 / the LDS gradient window, so this is not a physical utilisation"},
        "hbm_measured_gbs": (round(traffic / (kms * 1e-3) / 1e9, 1) if traffic else None),
        "physical": physical,
        # the other render kernel of the step (the forward when the backward dominates), same derivation
        "physical_other": (phys_fwd if ms_bwd >= ms_fwd else phys_bwd),
        "clock_hz_measured": round(clock_hz, 0), "clock_hz_peak": SHADER_CLOCK_HZ, "source_hash": src_hash,
        "phases_ms": {"pack": round(prof["ms_pack"] / max(prof["n_pack"], 1), 4), "fwd": round(ms_fwd, 4),
                      "memset": round(prof["ms_memset"] / max(prof["n_memset"], 1), 4), "bwd": round(ms_bwd, 4),
                      "unpack": round(prof["ms_unpack"] / max(prof["n_unpack"], 1), 4)},
        "in_aabb_samples_per_ray": round(s_in_total / max(R, 1), 2),
        # whole-step figure in BASELINE.md's convention: rays/s x (S_in*384 + 72) B, per GPU
        "step_alg_gbs_per_gpu": round(rays_per_s / world * (s_in_total / max(R, 1) * 384 + 72) / 1e9, 2),
    }

    # ---- secondary lines: the same step at 100x100 (BASELINE.json asks for both image sizes), N = 1 default run only:
    # one camera per step (what one SDS iteration at that size is) and a multi-view step of 8 cameras in ONE launch
    # (VoxeRenderCfg::image_height: K images back to back, pixel tiles per camera) ----
    secondary = None
    if world == 1 and default_cfg and not args.no_secondary:
        hw2 = 100

        def small_step_bench(K, hw2=hw2, cam0=None, steps=None):
            cam0 = args.camera if cam0 is None else cam0
            steps = args.steps if steps is None else steps
            ros, rds = [], []
            for i in range(K):
                p_i = pose_spherical(*synth_pose_angles(cam0 + 11 * i, 100), RADIUS)
                a, b = ops.cast_rays(hw2, hw2, focal_for(hw2), p_i.rotation, p_i.translation, dev)
                ros.append(a)
                rds.append(b)
            ro2, rd2 = torch.cat(ros).contiguous(), torch.cat(rds).contiguous()
            R2 = ro2.shape[0]
            # r06: the one-camera line cycles through the same cameras as the headline (its `value` is a mean over views too)
            sets = [(ro2, rd2)]
            if K == 1 and n_views > 1:
                sets = []
                for j in range(n_views):
                    p_j = pose_spherical(*synth_pose_angles(camera_of(j, 0), 100), RADIUS)
                    sets.append(ops.cast_rays(hw2, hw2, focal_for(hw2), p_j.rotation, p_j.translation, dev))
            turn = [0]
            p2 = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw2,
                                  image_height=hw2 if K > 1 else 0)
            g2 = torch.randn((R2, 3), generator=torch.Generator().manual_seed(44)).to(dev)
            out2 = [torch.empty((R2, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
            ws2 = ops.Workspace()
            probe2 = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, white_bkgd=True, image_width=hw2,
                                      image_height=hw2 if K > 1 else 0)
            s_in2 = sum(int(ops.sample_probe(spec, probe2, dens, feat, a_, b_, outputs=("inside",))["inside"].sum().item())
                        for a_, b_ in sets) / len(sets)

            def step2():
                ro2, rd2 = sets[turn[0] % len(sets)]
                turn[0] += 1
                if fused:
                    return fused_step(p2, ro2, rd2, out2, g2, ws2)
                step_no[0] += 1
                rng = (42, step_no[0])
                ops.render_fwd_into(spec, p2, dens, feat, ro2, rd2, None, *out2, ws2, rng)
                ops.render_bwd_into(spec, p2, dens, feat, ro2, rd2, None, out2[0], out2[1], out2[2], g2, None, None,
                                    d_dens, d_feat, ws2, rng)
                ops.adam_step_(flat_p, flat_g, exp_avg, exp_avg_sq, step_no[0], lr=1e-4)

            first[0] = True   # (another workspace: its gradient region starts uncleared)
            gc.collect()      # (a generation-2 collection of the interpreter inside the timed steps showed up as a 35 ms stall;
            gc.disable()      #  behind the warm-up it would let the chip drop its clocks: it runs BEFORE the warm-up)
            for _ in range(max(args.warmup, 10)):
                step2()
            ops.profile_enable(True)
            torch.cuda.synchronize()
            turn[0] = 0
            steps = max(steps, len(sets))       # (every view at least once)
            t2 = time.perf_counter()
            for _ in range(steps):
                step2()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t2
            gc.enable()
            pr2 = ops.profile_read()
            ops.profile_enable(False)
            b2 = pr2["ms_bwd"] / max(pr2["n_bwd"], 1)
            phys2 = None
            if K > 1:   # the 8-camera launch takes the space-binned route: ceilings of its backward kernel (one launch per PH_BWD)
                phys2 = physical_of("voxe::region_bwd_kernel<3, 1", lambda k: True, b2)
            return {
                "workload": f"same grid and step, {K} x {hw2}x{hw2} camera(s) in one launch"
                            + (f", the steps cycle through the headline's {len(sets)} cameras" if len(sets) > 1 else ""), "cameras_per_step": K,
                "views": len(sets),
                "value": round(R2 * steps / e2, 1), "unit": "rays/s",
                "ms_per_step": round(1e3 * e2 / steps, 4), "in_aabb_samples_per_ray": round(s_in2 / R2, 2),
                "bwd_ms": round(b2, 4), "fwd_ms": round(pr2["ms_fwd"] / max(pr2["n_fwd"], 1), 4),
                "roofline_frac_bwd": round((s_in2 * 256 + R2 * 56) / (b2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if b2 > 0 else None,
                "physical_bwd": phys2,
            }

        secondary = small_step_bench(1)
        secondary["multi_view"] = small_step_bench(8)
        # the reconstruction trainer's iteration (BASELINE.json configs[1] scale: 32 768 random rays over 8 cameras, specular +
        # diffuse L1, fused Adam) as ONE library call, voxe_recon_step -- tools/recon_bench.py has the trainer-level version
        def recon_iteration_bench(iters):
            K, B = 8, 32768
            d2, f2 = dens.clone(), feat.clone()
            st_d, st_f = (torch.zeros_like(d2), torch.zeros_like(d2)), (torch.zeros_like(f2), torch.zeros_like(f2))
            cams = [pose_spherical(*synth_pose_angles(3 + 11 * i, 100), RADIUS) for i in range(K)]
            poses = torch.stack([torch.cat([p_i.rotation, p_i.translation], dim=-1) for p_i in cams]).to(dev).contiguous()
            images = torch.rand((K, 3, HW, HW), generator=torch.Generator().manual_seed(45)).to(dev)
            losses = torch.zeros(4, device=dev)
            ws_a, ws_b = ops.Workspace(), ops.Workspace()
            pr = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True)

            def it(n, hint):
                ops.recon_step_(spec, pr, d2, f2, ws_a, ws_b, HW, HW, focal_for(HW), poses, None, images, B, True, st_d, st_f,
                                n, n, 1e-4, losses, (77, 10 * n), zero_gradient_first=(n == 1))
                if hint:   # voxe_recon_prefetch: iteration n + 1's batch + segment tables behind this iteration's backward / Adam
                    ops.recon_prefetch_(spec, pr, d2, f2, ws_a, ws_b, HW, HW, focal_for(HW), poses, None, images, B, True, losses,
                                        (77, 10 * (n + 1)))

            def timed(first, hint):
                for n in range(first, first + 10):
                    it(n, hint)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for n in range(first + 10, first + 10 + iters):
                    it(n, hint)
                torch.cuda.synchronize()
                return (time.perf_counter() - t3) / iters

            gc.collect()
            gc.disable()
            e_plain = timed(1, False)
            e3 = timed(11 + iters, True)
            gc.enable()
            return {"workload": f"voxe_recon_step: {B} random rays over {K} cameras ({HW}x{HW}), specular + diffuse render, L1 losses, "
                                "backward, fused Adam -- one library call per iteration, the next iteration's batch + segment tables "
                                "assembled behind this one's backward / Adam (voxe_recon_prefetch, as the trainer does)",
                    "ms_per_iteration": round(1e3 * e3, 4), "iterations_per_s": round(1.0 / e3, 1),
                    "value": round(2 * B / e3, 1), "unit": "rendered rays/s (2 renders, fwd + bwd)",
                    "ms_per_iteration_without_prefetch": round(1e3 * e_plain, 4)}

        secondary["recon_iteration"] = recon_iteration_bench(max(args.steps, 20))

        # BASELINE.json configs[2]: one SDS-edit iteration without the UNet -- the 266x266 render the edit loop draws per step
        # (modules/sds_trainer.py:283-340 of the reference: render -> [score distillation: dL/dcolour from the UNet, PyTorch-ROCm]
        # -> backward -> density-correlation regulariser (weight 200) -> Adam), with a fixed dL/dcolour in place of the network
        def sds_iteration_bench(iters, hw3=266):
            d3, f3 = dens.clone(), feat.clone()
            ref3 = dens.clone()
            st_d, st_f = (torch.zeros_like(d3), torch.zeros_like(d3)), (torch.zeros_like(f3), torch.zeros_like(f3))
            p_i = pose_spherical(*synth_pose_angles(args.camera, 100), RADIUS)
            ro3, rd3 = ops.cast_rays(hw3, hw3, focal_for(hw3), p_i.rotation, p_i.translation, dev)
            R3 = ro3.shape[0]
            p3 = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw3)
            g3 = torch.randn((R3, 3), generator=torch.Generator().manual_seed(46)).to(dev)
            out3 = [torch.empty((R3, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
            ws3 = ops.Workspace()
            loss3 = torch.zeros((), dtype=torch.float32, device=dev)
            cnt = [0]

            def it():
                cnt[0] += 1
                rng = (43, cnt[0])
                ops.render_fwd_into(spec, p3, d3, f3, ro3, rd3, None, *out3, ws3, rng)
                layout = ops.render_bwd_acc(spec, p3, d3, f3, ro3, rd3, None, out3[0], out3[1], out3[2], g3, None, None, ws3, rng,
                                            zero_first=(cnt[0] == 1))
                ops.grid_adam_step_(spec, d3, f3, layout, ws3, cnt[0], 1e-4, state_densities=st_d, state_features=st_f,
                                    dcl_reference=ref3, dcl_weight=200.0, dcl_loss=loss3)

            gc.collect()
            gc.disable()
            for _ in range(10):
                it()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(iters):
                it()
            torch.cuda.synchronize()
            e3 = (time.perf_counter() - t3) / iters
            # the per-phase device times come from a SEPARATE short run: the phase timer's event pairs cost host and queue time
            # (0.9 -> 1.26 ms on the refinement iteration), which does not belong in ms_per_iteration
            ops.profile_enable(True)
            for _ in range(5):
                it()
            torch.cuda.synchronize()
            gc.enable()
            pr3 = ops.profile_read()
            ops.profile_enable(False)
            return {"workload": f"SDS-edit iteration without the UNet: {hw3}x{hw3} render forward + backward (fixed dL/dcolour), "
                                "density-correlation regulariser inside the fused grid step (BASELINE.json configs[2])",
                    "ms_per_iteration": round(1e3 * e3, 4), "value": round(R3 / e3, 1), "unit": "rays/s",
                    "fwd_ms": round(pr3["ms_fwd"] / max(pr3["n_fwd"], 1), 4), "bwd_ms": round(pr3["ms_bwd"] / max(pr3["n_bwd"], 1), 4)}

        # BASELINE.json configs[3]: one iteration of the attention-grid refinement without the UNet
        # (modules/attn_grid_trainer.py:335-378 of the reference: the two attention renders, masked L1 against the cross-attention
        # maps, TV on both attention grids, Adam), with fixed maps in place of the network's
        def refine_iteration_bench(iters, hw3=266):
            spec_a = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=100.0 / 3.0, density_pre_act=abi.ACT_IDENTITY,
                                  density_post_act=abi.ACT_SOFTPLUS, feature_kind=abi.FEAT_ATTN)
            p_i = pose_spherical(*synth_pose_angles(args.camera, 100), RADIUS)
            ro3, rd3 = ops.cast_rays(hw3, hw3, focal_for(hw3), p_i.rotation, p_i.translation, dev)
            p3 = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=hw3)
            maps = [torch.rand((hw3, hw3), generator=torch.Generator().manual_seed(47 + i)).to(dev) for i in range(2)]
            tv_w, lr3 = 0.01, 0.035

            def run(it, warm):
                gc.collect()
                gc.disable()
                for _ in range(warm):
                    it()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(iters):
                    it()
                torch.cuda.synchronize()
                e = (time.perf_counter() - t3) / iters
                ops.profile_enable(True)      # (per-phase times from a separate short run, see sds_iteration_bench)
                for _ in range(5):
                    it()
                torch.cuda.synchronize()
                gc.enable()
                pr = ops.profile_read()
                ops.profile_enable(False)
                return e, pr

            # the product path (modules/attn_grid_trainer.py of this package): ONE library call per attention grid and iteration
            # (voxe_attn_refine_step: attention render -> masked L1 + TV -> backward -> Adam)
            grids = [torch.full((dens.shape[0], dens.shape[1], dens.shape[2], 1), -2.0, device=dev) for _ in range(2)]
            states = [(torch.zeros_like(a_), torch.zeros_like(a_)) for a_ in grids]
            wss = [ops.Workspace(), ops.Workspace()]
            losses3 = torch.zeros((2, 2), dtype=torch.float32, device=dev)
            cnt = [0]

            def it_lib():
                cnt[0] += 1
                for i, (a_, s_, m_, w_) in enumerate(zip(grids, states, maps, wss)):
                    ops.attn_refine_step_(spec_a, p3, dens, a_, ro3, rd3, m_.reshape(-1), w_, cnt[0], lr3, s_, tv_w, losses3[i],
                                          rng=(43, cnt[0]))

            e3, pr3 = run(it_lib, 5)

            # the same iteration written like the reference (render -> calc_loss_on_attn_grid + TV -> loss.backward() ->
            # optimiser.step()) through the binding's autograd entry points
            from thre3d_atom.modules.optim import VoxeAdam
            from thre3d_atom.modules.refinement_functions import calc_loss_on_attn_grid
            grids_a = [torch.full((dens.shape[0], dens.shape[1], dens.shape[2], 1), -2.0, device=dev).requires_grad_(True) for _ in range(2)]
            opts = [VoxeAdam([{"params": [a_], "lr": lr3}], betas=(0.9, 0.999)) for a_ in grids_a]
            wss_a = [ops.Workspace(), ops.Workspace()]

            def it_autograd():
                for a_, o_, m_, w_ in zip(grids_a, opts, maps, wss_a):
                    att, _, _, _ = ops.render(spec_a, p3, dens, a_, ro3, rd3, workspace=w_)
                    loss = calc_loss_on_attn_grid(att, m_) + ops.tv_loss_on_grid(a_) * tv_w
                    loss.backward()
                    o_.step()
                    o_.zero_grad()

            e3a, _ = run(it_autograd, 5)
            return {"workload": f"attention-refinement iteration without the UNet: two {hw3}x{hw3} attention renders (forward + backward), "
                                "masked L1, TV on both attention grids, Adam (BASELINE.json configs[3]); one library call per grid "
                                "(voxe_attn_refine_step)",
                    "ms_per_iteration": round(1e3 * e3, 4), "value": round(2 * hw3 * hw3 / e3, 1), "unit": "rendered rays/s (2 renders, fwd + bwd)",
                    "fwd_ms_per_render": round(pr3["ms_fwd"] / max(pr3["n_fwd"], 1), 4),
                    "bwd_ms_per_render": round(pr3["ms_bwd"] / max(pr3["n_bwd"], 1), 4),
                    "autograd_loop_ms_per_iteration": round(1e3 * e3a, 4)}

        # ---- what the 1 -> 8 GPU curve should look like: a MODEL, from this GPU's measured render times and
        # the link rates of SURVEY.md section 5 -- there is no multi-GPU node in the builder's pool, the driver's SCALE_rNN.json is
        # to be read against these predictions.  Per step and rank: render (fwd + bwd of its rays) + Adam on 1/N of the grid +
        # exchange of the grid gradient (reduce-scatter) and of the packed grid (all-gather), G = 16 B x voxels each way.
        #   direct  every rank sends slab j straight to rank j over its own xGMI link (all-to-all / "pipelined"): 2 (G / N) / link
        #   ring    per-link bound ring collectives:                                                       2 (N - 1) / N G / link
        def scaling_model():
            from thre3d_atom.modules.parallel import shard_rows

            link = 153.0e9                                   # B/s per xGMI link and direction
            g_bytes = nvox * 16.0
            t_adam = max(ms_per_step - ms_fwd - ms_bwd, 0.0)   # everything of the 1-GPU step that is not the two render phases
            ro0, rd0 = view_rays[0]

            def band_ms(lo, hi):
                ro_b, rd_b = ro0[lo * HW: hi * HW].contiguous(), rd0[lo * HW: hi * HW].contiguous()
                Rb = ro_b.shape[0]
                pb = ops.RenderParams(num_samples=S, near=NEAR, far=FAR, perturb=True, white_bkgd=True, image_width=HW)
                outs = [torch.empty((Rb, n), dtype=torch.float32, device=dev) for n in (3, 1, 1, 1)]
                gb = g_colour[lo * HW: hi * HW].contiguous()
                wsb = ops.Workspace()
                for i in range(9):
                    if i == 3:
                        torch.cuda.synchronize()
                        ops.profile_enable(True)
                    ops.render_fwd_into(spec, pb, dens, feat, ro_b, rd_b, None, *outs, wsb, (42, 900 + i))
                    ops.render_bwd_acc(spec, pb, dens, feat, ro_b, rd_b, None, outs[0], outs[1], outs[2], gb, None, None, wsb,
                                       (42, 900 + i), zero_first=True)
                torch.cuda.synchronize()
                prb = ops.profile_read()
                ops.profile_enable(False)
                return (prb["ms_fwd"] + prb["ms_memset"] + prb["ms_bwd"]) / max(prb["n_bwd"], 1)

            rows_out = {}
            for n in (2, 4, 8):
                bands = [shard_rows(HW, r, n) for r in range(n)]
                times = [band_ms(lo, hi) for lo, hi in bands]
                t_band = max(times)
                x_direct = 2.0 * (g_bytes / n) / link * 1e3
                x_ring = 2.0 * (n - 1) / n * g_bytes / link * 1e3
                t_full = ms_fwd + ms_bwd
                rows_out[str(n)] = {
                    "band_rows": [hi - lo for lo, hi in bands], "band_render_ms_max": round(t_band, 4), "band_render_ms_mean": round(sum(times) / n, 4),
                    "adam_ms": round(t_adam / n, 4), "exchange_ms_direct": round(x_direct, 4), "exchange_ms_ring": round(x_ring, 4),
                    "weak": {"rays_per_s_direct": round(n * HW * HW / ((t_full + t_adam / n + x_direct) * 1e-3), 1),
                             "rays_per_s_ring": round(n * HW * HW / ((t_full + t_adam / n + x_ring) * 1e-3), 1)},
                    "strong": {"rays_per_s_direct": round(HW * HW / ((t_band + t_adam / n + x_direct) * 1e-3), 1),
                               "rays_per_s_ring": round(HW * HW / ((t_band + t_adam / n + x_ring) * 1e-3), 1)},
                }
            return {"what": "PREDICTION, not a measurement: 1-GPU render times of this run (row bands of the first view for --scaling strong: "
                            "the slowest band of the N) + Adam / N + gradient and packed-grid exchange at 153 GB/s per xGMI link and direction "
                            "(direct: all N - 1 links at once; ring: per-link bound).  Nothing overlaps the exchange (DESIGN.md section 6)",
                    "one_gpu": {"render_ms": round(ms_fwd + ms_bwd, 4), "adam_and_rest_ms": round(t_adam, 4), "rays_per_s": round(rays_per_s, 1)},
                    "grid_bytes_each_way": int(g_bytes), "by_gpus": rows_out}

        secondary["scaling_model"] = scaling_model()
        secondary["sds_iteration"] = sds_iteration_bench(max(args.steps, 20))
        secondary["refine_iteration"] = refine_iteration_bench(max(args.steps, 20))

    # ---- same-GPU baseline: a plain PyTorch restatement of the path (the reference's execution model: ~40 ATen ops with
    # [rays x samples] temporaries + autograd + torch.optim.Adam; tools/torch_baseline.py, pinned to the reference's
    # outputs and gradients by tests/test_torch_baseline.py) under PyTorch-ROCm on this very GPU, outside the timed region ----
    gpu_baseline = None
    if rank == 0 and world == 1 and not args.no_gpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import torch_baseline as tb

        torch.cuda.empty_cache()

        def gpu_pass(hw, steps):
            ro_b, rd_b = ops.cast_rays(hw, hw, focal_for(hw), pose.rotation, pose.translation, dev)
            gb = torch.randn((hw * hw, 3), generator=torch.Generator().manual_seed(43)).to(dev)
            dt = tb.time_step(dens_cpu.to(dev), feat_cpu.to(dev), aabb, 100.0 / 3.0, ro_b, rd_b, gb, S, NEAR, FAR,
                              chunk=32768, steps=steps, warmup=1, median=True)
            return hw * hw / dt, dt

        try:
            rps_full, dt_full = gpu_pass(HW, 3)
            rps_small, dt_small = gpu_pass(100, 5)
            gpu_baseline = {
                "value": round(rps_full, 1), "unit": "rays/s", "kind": "port",
                "what": "PyTorch-ROCm restatement of the reference path (F.grid_sample x2, softplus, exp, cumprod, autograd, "
                        "torch.optim.Adam; 32768-ray chunks like parallel_rays_chunk_size) on this GPU, same grid / camera / S",
                "ms_per_step": round(1e3 * dt_full, 2), "speedup": round(rays_per_s / rps_full, 1),
                "reps": 3, "protocol": "1 warm-up step + median of 3 individually timed steps (5 at 100x100)",
                "at_100x100": {"value": round(rps_small, 1), "ms_per_step": round(1e3 * dt_small, 2)},
                "torch": torch.__version__,
            }
        except RuntimeError as e:          # (e.g. out of memory on a shared box): report, do not fail the bench line
            gpu_baseline = {"error": str(e).splitlines()[0][:200]}
        torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle (a C port of the reference path) on the host cores, bounded sample ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import voxe_oracle as vo
        from voxe_hip.desc import make_render_cfg

        grid = vo.Grid(dens_cpu.numpy(), feat_cpu.numpy(), aabb, 100.0 / 3.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
        cfg = make_render_cfg(S, NEAR, FAR, perturb=not args.no_jitter, white_bkgd=True, seed=42, rng_offset=1)

        def cpu_pass(hw):
            o, d = vo.cast_rays(hw, hw, focal_for(hw), pose.rotation.numpy(), pose.translation.numpy())
            gc = np.random.default_rng(43).standard_normal((hw * hw, 3)).astype(np.float32)
            t1 = time.perf_counter()
            vo.render_fwd(grid, cfg, o, d)
            vo.render_bwd(grid, cfg, o, d, gc)
            return time.perf_counter() - t1

        # bounded sample: the GPU's full image when a 128x128 probe says four passes fit ~30 s of wall time, a smaller image
        # otherwise.  Protocol: one warm-up pass, then the MEDIAN of `reps` timed passes.
        hw = args.cpu_sample
        cpu_pass(64)                          # thread start-up / first touch
        if hw <= 0:
            probe = cpu_pass(128)
            full = probe * (HW / 128.0) ** 2  # upper bound: the per-call fixed cost (gradient grids) does not scale
            hw = HW if 4.0 * full <= 30.0 else int(max(128, 128 * (6.0 / max(probe, 1e-3)) ** 0.5))
        cpu_pass(hw)                          # warm-up at the sample's size
        reps = 3
        passes = sorted(cpu_pass(hw) for _ in range(reps))
        dt = passes[reps // 2]
        threads = vo.num_threads()
        cpu_baseline = {
            "value": round(hw * hw / dt, 1), "unit": "rays/s", "cores": threads, "kind": "port", "reps": reps,
            "passes_s": [round(x, 3) for x in passes],
            # the port is FASTER than the reference's own CPU path: on the same 8 vCPUs (build container) it renders
            # 8.9x the reference PyTorch's rays/s at 400x400 and 2.6x at 100x100, so the reference
            # on these host cores would be about value / vs_reference_factor
            "vs_reference_factor": ORACLE_VS_REFERENCE_400 if hw >= 256 else ORACLE_VS_REFERENCE_100,
            "reference_equivalent_rays_per_s": round(hw * hw / dt / (ORACLE_VS_REFERENCE_400 if hw >= 256 else ORACLE_VS_REFERENCE_100), 1),
            "sample": f"{hw}x{hw} rays of camera {args.camera} (same {G}^3 grid, S={S}, jitter on), 1 forward + 1 backward "
                      f"of oracle/voxe_cpu.c with OpenMP per pass; 1 warm-up + median of {reps} passes: {dt:.2f} s wall x {threads} "
                      f"threads = {dt * threads:.0f} core-seconds",
        }

    views_report = None
    if rank == 0:
        by_view = {}
        for v, ms in zip(timed_views, step_ms_in_order):
            by_view.setdefault(v, []).append(ms)
        mean_ms = {v: sum(x) / len(x) for v, x in by_view.items()}
        slow, fast = max(mean_ms, key=mean_ms.get), min(mean_ms, key=mean_ms.get)
        views_report = {
            "count": n_views, "cameras_of_rank0": view_cams,
            "ms_per_step_by_view": [round(mean_ms[v], 4) if v in mean_ms else None for v in range(n_views)],
            "in_aabb_samples_per_ray_by_view": [round(x / max(R, 1), 1) for x in s_in_views],
            "min_rays_per_s": round(R / (mean_ms[slow] * 1e-3), 1), "min_camera": view_cams[slow],
            "max_rays_per_s": round(R / (mean_ms[fast] * 1e-3), 1), "max_camera": view_cams[fast],
            "note": "per-view figures are rank 0's rays over the device time of its step (events on the launch stream)",
        }
        out = {
            "metric": "rendered rays/sec (fwd+bwd)", "value": round(rays_per_s, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # per-step device time between consecutive events on the launch stream (rank 0): median / min / max of the K
            # timed steps; `ms_per_step` above is the contract's wall clock over all K steps / K, max over ranks
            "ms_per_step_median": round(step_ms[len(step_ms) // 2], 4), "ms_per_step_min": round(step_ms[0], 4),
            "ms_per_step_max": round(step_ms[-1], 4), "ms_first_steps": [round(x, 4) for x in step_ms_in_order[:6]],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{G}^3 SH-0 softplus ReLU-field grid ({args.scene}), "
                            + (f"ONE {HW}x{HW} camera per step split into row bands over the GPUs, " if strong else f"one {HW}x{HW} camera per GPU and step, ") +
                            (f"the steps cycle through {n_views} cameras of the 100-view set, " if n_views > 1 else f"camera {args.camera}, ") +
                            f"S={S}, jitter {'off' if args.no_jitter else 'on'}, white bkgd, ray order {args.ray_order}; step = render fwd + bwd"
                            f"{' + RCCL all-reduce of the grid gradient' if world > 1 else ''}"
                            f"{'' if args.no_adam else (' + Adam (fused grid step)' if fused else ' + Adam')}",
                "grid": G, "image": [HW, HW], "samples_per_ray": S, "rays_per_gpu_per_step": R,
                # r06 (the single headline camera was the fastest of seven): the timed steps cycle through
                # `count` cameras of the 100-view set, so `value` IS the mean over views; per view (rank 0): device time between
                # the step's events on the launch stream
                "views": views_report,
                "grad_exchange": (opt.mode if fused else ("all-reduce" if dist is not None else "none")), "parallelism": (f"rows of one image sharded over {world} GPU(s), grid replicated" if strong else f"rays sharded by camera over {world} GPU(s), grid replicated"),
                "rows_of_rank0": list(rows),
                "replicas_consistent": replicas_consistent, "backend": (backend if dist is not None else None),
                "exchange_autotune_ms": (opt.tuned_ms if fused else None),
                # device time per step between the backward and the next render that is NOT the local optimiser kernel:
                # gradient reduce-scatter / all-to-all + all-gather of the packed grid (nothing overlaps it: the backward
                # produces the whole gradient, the next forward consumes the whole grid -- DESIGN.md section 6)
                "exchange_ms_per_step": (round(exchange_ms, 4) if exchange_ms is not None else None),
                "per_rank": per_rank,
                "term_eps": args.term_eps, "optimizer": ("none" if args.no_adam else ("fused" if fused else "split")),
                "untimed_warmup_ms": round(1e3 * untimed_s, 1), "untimed_steps_before_timing": untimed_steps + args.warmup,
            },
            "roofline": roofline, "secondary": secondary,
            "gpu_baseline": gpu_baseline, "cpu_baseline": cpu_baseline,
        }
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so the JSON is the LAST line
        import ctypes

        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
