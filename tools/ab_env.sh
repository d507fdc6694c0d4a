#!/bin/bash
# A/B of environment knobs: tools/ab_env.sh "VAR=val" "VAR=val2" ... -> fwd / bwd / step for three cameras
for kv in "$@"; do
  line="$kv:"
  for cam in 3 40 77; do
    r=$(env $kv python bench.py --no-cpu-baseline --steps 30 --camera $cam 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step'])")
    line="$line  cam$cam fwd/bwd/step $r"
  done
  echo "$line"
done
