#!/bin/bash
# A/B of bench.py command-line flag sets (headline line only):  bash tools/ab_flags.sh "<flags>" "<flags>" ...
for flags in "$@"; do
  r=$(python bench.py $flags --steps 50 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(round(d['value']/1e6,2),'M rays/s', d['ms_per_step'],'ms  fwd',p['fwd'],'bwd',p['bwd'])")
  echo "[$flags]: $r"
done
