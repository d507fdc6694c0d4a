#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Run through gpurun:
#   gpurun --timeout 1500 -- bash tools/gpu_check.sh
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== host =="; nproc; lscpu | grep -E "Model name|Socket|Core" | head -4
echo "== rocm-smi =="; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench =="
timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
timeout 300 python bench.py --scene sphere --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_sphere.log
timeout 300 python bench.py --image 100 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_100.log
echo "== rocprofv3 =="
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
