#!/bin/bash
# Full evidence run for a round: GPU tests, smoke, bench lines, rocprofv3 kernel stats.
#   1. gpurun -- bash tools/gpu_pmc.sh            2. python tools/pmc_to_json.py r04   (stamps the summary with the source hash)
#   3. gpurun --timeout 3000 -- bash tools/gpu_round.sh r04        4. copy what is to be judged from gpurun_out/r04 to profiles/
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
nproc > $O/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $O/host.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.log
# the driver's command line first (its contract: --steps 20 --warmup 5), then the long one on the same lease
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_400.json
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_400_long.json
timeout 300 python bench.py --scene sphere --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_400_sphere.json
timeout 300 python bench.py --image 100 --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_100.json
timeout 300 python bench.py --grid 256 --image 800 --no-cpu-baseline --no-gpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_256_800.json
timeout 300 env VOXE_BWD_MODE=scatter python bench.py --no-cpu-baseline --no-gpu-baseline --steps 5 2>/dev/null | tail -1 > $O/bench_400_scatter_bwd.json
timeout 300 python bench.py --term-eps 1e-4 --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_400_term1e-4.json
timeout 300 python bench.py --image 200 --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_200.json
timeout 300 python bench.py --image 266 --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_266.json
timeout 300 python bench.py --optimizer split --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_400_split_optimizer.json
timeout 300 env VOXE_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_400_rccl_1rank.json
# two ranks sharing the one GPU, exchanging through gloo: the N > 1 code path with the real kernels (correctness, not a measurement)
for ex in reduce-scatter all-to-all all-reduce pipelined auto; do timeout 600 env VOXE_GRAD_EXCHANGE=$ex VOXE_BENCH_BACKEND=gloo VOXE_BENCH_PRE_WARM_MS=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | tail -1; done > $O/two_ranks_one_gpu_gloo.jsonl
# ... and the driver's N = 8 command line, eight ranks sharing the GPU
timeout 600 env VOXE_GRAD_EXCHANGE=reduce-scatter VOXE_BENCH_BACKEND=gloo VOXE_BENCH_PRE_WARM_MS=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 8 --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/eight_ranks_one_gpu_gloo.json
timeout 300 python tools/refine_bench.py 160 2>/dev/null | tail -4 > $O/refine_bench.txt; cat $O/refine_bench.txt
timeout 300 python tools/recon_bench.py 2>/dev/null | tail -6 > $O/recon_bench.txt; cat $O/recon_bench.txt
(timeout 300 python tools/sh_bench.py 160 400 123; timeout 300 python tools/sh_bench.py 160 180 123 random) 2>/dev/null > $O/sh_bench.txt; cat $O/sh_bench.txt
# r03: strong scaling (ONE camera split into row bands) with two ranks on the one GPU (gloo), whole-grid passes with kernel-level
# times, gradient error per magnitude band, LDS / VALU / clock microbenchmarks, gradient truncation on a surface-like scene
timeout 600 env VOXE_GRAD_EXCHANGE=reduce-scatter VOXE_BENCH_BACKEND=gloo VOXE_BENCH_PRE_WARM_MS=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong 2>/dev/null | tail -1 > $O/two_ranks_one_gpu_gloo_strong.json
timeout 300 python tools/grid_pass_bench.py 2>/dev/null > $O/grid_passes.txt; cat $O/grid_passes.txt
timeout 300 python tools/band_probe.py 2>/dev/null > $O/band_probe.txt; tail -12 $O/band_probe.txt
timeout 300 python bench.py --scene sphere --term-eps 1e-4 --no-cpu-baseline --no-gpu-baseline 2>/dev/null | tail -1 > $O/bench_400_sphere_term1e-4.json
timeout 300 env RECON_PYTHON_ITER=1 python tools/recon_bench.py 2>/dev/null | tail -6 > $O/recon_bench_python_iteration.txt
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2),'Mrays/s', d['ms_per_step'],'ms', d['roofline']['phases_ms'])" 2>&1)"; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline --no-gpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof_bench.log 2>&1
# r06: the headline step alone (no secondary lines: every launch of the render kernels is a 400x400 view of the timed set), so that the
# kernels' average durations can be held against the bench line's HIP-event times
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_headline -o ${TAG}_headline -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-gpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof_headline.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_refine -o ${TAG}_refine -- python $GRAFT_REPO_ROOT/tools/refine_bench.py 160 > /dev/null 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_recon -o ${TAG}_recon -- python $GRAFT_REPO_ROOT/tools/recon_bench.py 20 > /dev/null 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_grid -o ${TAG}_grid -- python $GRAFT_REPO_ROOT/tools/grid_pass_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
head -6 $O/prof_recon/${TAG}_recon_kernel_stats.csv | cut -c1-160
head -12 $O/prof_refine/${TAG}_refine_kernel_stats.csv | cut -c1-160
head -8 $O/prof/${TAG}_kernel_stats.csv | cut -c1-200
# r04: bit-identity of the LDS-window forward over random cases, and the wide fuzz soak (shipped dispatch + tile kernel on small images)
(timeout 900 python tools/fwd_identity_sweep.py 60 2>/dev/null | tail -1; timeout 900 python tools/fwd_identity_sweep.py 60 123 2>/dev/null | tail -1) > $O/fwd_identity_sweep.txt; cat $O/fwd_identity_sweep.txt
VOXE_FUZZ_SEEDS=4000 timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -m gpu 2>&1 | tail -3 > $O/fuzz_soak.txt; cat $O/fuzz_soak.txt
# (PMC counters: run tools/gpu_pmc.sh BEFORE this script and tools/pmc_to_json.py <tag> locally -- bench.py only uses a PMC summary
#  whose source_hash equals the kernels it runs, so the summary has to exist, with the final sources, when the lines above are taken)
