#!/bin/bash
# A/B of whole-library variants on the headline step:  gpurun -- bash tools/ab_libs.sh <outdir> "<bench flags>" base tag1 tag2 ...   (base = in-tree)
OUT=gpurun_out/$1; FLAGS=$2; shift; shift; mkdir -p $OUT
F=$OUT/ab_libs.txt; : > $F
for rep in 1 2; do
  for tag in "$@"; do
    lib=""; [ "$tag" != "base" ] && lib=$PWD/variants/libvoxe_hip_$tag.so
    VOXE_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-gpu-baseline --no-secondary --steps 40 $FLAGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,2), 'M rays/s', d['ms_per_step'], 'ms', d['roofline']['phases_ms']['fwd'], d['roofline']['phases_ms']['bwd'])" >> $F
  done
done
cat $F
