#!/bin/bash
# A/B of environment knobs over the cameras of the driver's `secondary.views`: tools/ab_env_cams.sh "VAR=val" ...  ("X=0" = shipped)
for kv in "$@"; do
  line="$kv:"
  for cam in 3 0 12 26 40 77 90; do
    r=$(env $kv python bench.py --no-cpu-baseline --no-gpu-baseline --no-secondary --steps 30 --camera $cam 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step'])")
    line="$line  cam$cam $r"
  done
  echo "$line"
done
