#!/bin/bash
# per camera: in-AABB samples per ray, forward / backward ms, ns per 1000 in-AABB samples   bash tools/cam_table.sh <image> "<cameras>"
img=$1
for cam in $2; do
  python bench.py --image $img --camera $cam --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['in_aabb_samples_per_ray']; p=r['phases_ms']; n=$img*$img*s
print('image $img cam $cam: samples/ray', s, 'fwd', p['fwd'], 'bwd', p['bwd'], ' ps per in-AABB sample: fwd', round(p['fwd']*1e9/n,2), 'bwd', round(p['bwd']*1e9/n,2), ' M rays/s', round(d['value']/1e6,1))"
done
