#!/bin/bash
# bench over several synthetic cameras (view directions): gpurun -- bash tools/ab_cameras.sh
for cam in 0 3 17 50 77 91; do
  out=$(python bench.py --no-cpu-baseline --steps 10 --camera $cam 2>/dev/null | tail -1)
  echo "cam $cam :: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'Mrays/s', r['phases_ms']['fwd'], r['phases_ms']['bwd'], r['in_aabb_samples_per_ray'])")"
done
