#!/bin/bash
# A/B of an environment switch on the headline bench:  bash tools/ab_env_bench.sh VAR v1 v2 ... [-- extra bench args]
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "${1:-}" == "--" ] && shift
for v in "${vals[@]}"; do
  env $var=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
  python - "$var=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print(sys.argv[1], round(d["value"] / 1e6, 2), "M rays/s", d["ms_per_step"], "ms", d["roofline"]["phases_ms"])
PY
done
