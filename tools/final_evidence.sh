#!/bin/bash
# refresh the evidence that depends on the last kernel changes: GPU suite, smoke, default bench line, recon kernel stats
set -u
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.log
python bench.py 2>/dev/null | tail -1 > $O/bench_400.json
python tools/recon_bench.py 40 2>/dev/null | tail -6 | tee $O/recon_bench.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_recon -o recon -- python $GRAFT_REPO_ROOT/tools/recon_bench.py 20 > /dev/null 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline > /dev/null 2>&1
