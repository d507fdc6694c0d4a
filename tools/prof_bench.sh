#!/bin/bash
# rocprofv3 kernel statistics of bench.py with the given flags:  gpurun -- bash tools/prof_bench.sh <tag> [bench flags]
export TMPDIR=/tmp
tag=$1; shift
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gpu-baseline --no-secondary --steps 30 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1)
tail -1 gpurun_out/prof_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], d['roofline']['phases_ms'])"
python - "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/prof_{sys.argv[1]}/b_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.3: print("   ", r["Name"].replace("void ", "").replace("voxe::", "")[:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
