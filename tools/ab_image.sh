#!/bin/bash
# A/B of dispatch switches / library variants on one image size of bench.py (headline line only, no secondary):
#   bash tools/ab_image.sh <image> "<env assignments>" [more env sets ...]      e.g.  bash tools/ab_image.sh 100 "" "VOXE_TILE_KL=16"
img=$1; shift
for envs in "$@"; do
  r=$(env $envs python bench.py --image $img --steps 50 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(round(d['value']/1e6,2),'M rays/s', d['ms_per_step'],'ms  fwd',p['fwd'],'bwd',p['bwd'])")
  echo "image $img [$envs]: $r"
done
