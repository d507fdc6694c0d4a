#!/bin/bash
# A/B of the fused optimiser pass: kernel time of grid_adam_* per library (rocprofv3 kernel trace of bench.py) + the headline line
#   bash tools/ab_adam.sh "<lib or empty>" ...
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "${lib:-in-tree}")
  out=/tmp/ab_adam_$tag; rm -rf $out
  (cd /tmp && env ${lib:+VOXE_HIP_LIB=$GRAFT_REPO_ROOT/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- \
     python $GRAFT_REPO_ROOT/bench.py ${BENCH_FLAGS:-} --steps 50 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-secondary > $out.log 2>&1)
  python - $out $tag <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1:3]
line = [l for l in open(out + ".log") if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else None
print(f"== {tag}:", (round(d["value"] / 1e6, 2), "M rays/s", d["ms_per_step"], "ms") if d else "no bench line")
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "grid_adam" in r["Name"]:
            print("    ", r["Name"].split("(")[0][-40:], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
