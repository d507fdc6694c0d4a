#!/bin/bash
# A/B of library variants (tools/variants.py) on the whole step: ms/step of the headline bench + the 100x100 secondary.
#   gpurun -- bash tools/ab_step.sh tag1 tag2 ...   ("base" = the in-tree library); two interleaved rounds
for round in 1 2; do
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  r=$(VOXE_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['secondary']['ms_per_step'])")
  echo "$tag: $r"
done
done
