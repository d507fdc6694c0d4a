#!/bin/bash
# kernel times of the space-binned pipeline on bench.py's 8-camera line, per library:  bash tools/ab_region.sh "<lib or empty>" ...
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "${lib:-in-tree}")
  out=/tmp/ab_region_$tag; rm -rf $out
  (cd /tmp && env ${lib:+VOXE_HIP_LIB=$GRAFT_REPO_ROOT/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $out.log 2>&1)
  python - $out $tag <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1:3]
line = [l for l in open(out + ".log") if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else None
mv = d["secondary"]["multi_view"] if d else None
print(f"== {tag}: headline", round(d["value"] / 1e6, 2) if d else None, "8 cameras", round(mv["value"] / 1e6, 2) if mv else None, "M rays/s", mv["ms_per_step"] if mv else None, "ms")
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "region_" in r["Name"]:
            print("    ", r["Name"].split("(")[0][-34:], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
