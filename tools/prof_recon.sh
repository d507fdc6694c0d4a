#!/bin/bash
# rocprofv3 kernel statistics of the reconstruction iteration for library variants:  gpurun -- bash tools/prof_recon.sh tag1 tag2 ...  ("base" = in-tree)
export TMPDIR=/tmp
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=$GRAFT_REPO_ROOT/variants/libvoxe_hip_$tag.so
  (cd /tmp && VOXE_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_recon_$tag -o recon -- python $GRAFT_REPO_ROOT/tools/recon_bench.py 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_recon_$tag.log 2>&1)
  echo "== $tag: $(grep 'reconstruction iteration' gpurun_out/prof_recon_$tag.log | sed -e 's/.*spec+diffuse): //')"
  python - "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/prof_recon_{sys.argv[1]}/recon_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4: print("   ", r["Name"].replace("void ", "").replace("voxe::", "")[:50].ljust(50), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done
