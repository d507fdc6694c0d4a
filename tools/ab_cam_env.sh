#!/bin/bash
# backward ms per camera for environment sets:  bash tools/ab_cam_env.sh "<images>" "<cameras>" "<env set>" ...   ("-" = empty)
imgs=$1; cams=$2; shift 2
for img in $imgs; do for cam in $cams; do line="image $img cam $cam:"; for envs in "$@"; do
  [ "$envs" = "-" ] && e="" || e="$envs"
  r=$(env $e python bench.py $BENCH_FLAGS --image $img --camera $cam --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['phases_ms']['bwd'])")
  line="$line  [$envs] $r"; done; echo "$line"; done; done
