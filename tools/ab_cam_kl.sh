#!/bin/bash
# backward ms (and M rays/s) per camera for the lateral-window choices:  bash tools/ab_cam_kl.sh "<images>" "<cameras>" "<KLs>"
for img in $1; do for cam in $2; do line="image $img cam $cam:"; for kl in $3; do
  r=$(VOXE_TILE_KL=$kl python bench.py --image $img --camera $cam --steps 40 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['phases_ms']['bwd'], round(d['value']/1e6,1))")
  line="$line  KL$kl bwd/Mrays $r"; done; echo "$line"; done; done
