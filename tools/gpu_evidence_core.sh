#!/bin/bash
# The part of tools/gpu_round.sh that the judged numbers come from, in ~15 minutes of box time: GPU suite, smoke, the driver's
# bench line, SH / reconstruction / grid-pass benches, rocprofv3 kernel statistics of bench and recon.  (The full script adds the
# alternative bench lines, the multi-rank-on-one-GPU runs, the fuzz soak and the forward identity sweep.)
#   gpurun --timeout 1500 -- bash tools/gpu_evidence_core.sh r05
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_400.json
timeout 300 python bench.py --image 100 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_100.json
timeout 300 python bench.py --image 266 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_266.json
timeout 300 python tools/recon_bench.py 2>/dev/null | tail -6 > $O/recon_bench.txt; cat $O/recon_bench.txt
(timeout 300 python tools/sh_bench.py 160 400 123; timeout 300 python tools/sh_bench.py 160 180 123 random) 2>/dev/null > $O/sh_bench.txt; cat $O/sh_bench.txt
timeout 300 python tools/grid_pass_bench.py 2>/dev/null > $O/grid_passes.txt; cat $O/grid_passes.txt
timeout 300 python tools/band_probe.py 2>/dev/null > $O/band_probe.txt; tail -12 $O/band_probe.txt
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2),'Mrays/s', d['ms_per_step'],'ms', d['roofline']['phases_ms'])" 2>&1)"; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline --no-gpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof_bench.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_recon -o ${TAG}_recon -- python $GRAFT_REPO_ROOT/tools/recon_bench.py 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
head -8 $O/prof_recon/${TAG}_recon_kernel_stats.csv | cut -c1-160
head -8 $O/prof/${TAG}_kernel_stats.csv | cut -c1-200
