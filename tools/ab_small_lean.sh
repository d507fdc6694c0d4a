#!/bin/bash
# small images: shipped window choice vs the 8-wide window (lean kernel) vs lean off; fwd / bwd / step ms, M rays/s
for img in 266 200 100; do for env in "" "VOXE_TILE_KL=8" "VOXE_TILE_LEAN=0"; do
  r=$(env $env python bench.py --image $img --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step'], round(d['value']/1e6,1))")
  echo "image $img [$env]: $r"; done; done
