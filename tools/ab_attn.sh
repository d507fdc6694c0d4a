export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in "VOXE_TILE_KL=0" "VOXE_TILE_KL=8" "VOXE_TILE_KL=10" "VOXE_TILE_LEAN=0" "VOXE_TILE_LEAN=0 VOXE_TILE_KL=8"; do
  echo "== $cfg"
  cd /tmp && env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o x -- python $R/tools/refine_iter_bench.py 160 266 40 L 2>&1 | grep "library call"
  python - <<PY
import csv
for r in list(csv.DictReader(open("/tmp/prof_x/x_kernel_stats.csv")))[:3]:
    print("   %-70s %5s %9.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf /tmp/prof_x
done
