#!/bin/bash
# A/B of kernel variants built by tools/variants.py: backward time of the headline bench for two cameras.
#   gpurun -- bash tools/ab_variants.sh tag1 tag2 ...   ("base" = the in-tree library)
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  line="$tag:"
  for cam in 3 40 77; do
    r=$(VOXE_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 30 --camera $cam 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['phases_ms']['bwd'], d['ms_per_step'])")
    line="$line  cam$cam bwd/step $r"
  done
  echo "$line"
done
