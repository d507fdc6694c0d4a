#!/bin/bash
# A/B runs of bench.py under environment variants; prints value + per-phase ms.
#   gpurun --timeout 900 -- bash tools/ab_bench.sh "VOXE_TILE_MAP=band" "VOXE_TILE_MAP=rows" ...
set -u
mkdir -p gpurun_out
EXTRA=${AB_ARGS:-"--no-cpu-baseline --steps 20"}
for v in "$@"; do
  out=$(env $v python bench.py $EXTRA 2>/dev/null | tail -1)
  echo "$v :: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'Mrays/s', d['ms_per_step'],'ms/step', r['phases_ms'])")"
done | tee -a gpurun_out/ab.log
