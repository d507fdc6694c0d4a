#!/bin/bash
# backward time of the headline bench over many cameras for library variants: tools/ab_cams.sh tag...
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  line="$tag:"
  for cam in 0 7 14 21 28 35 42 49 56 63 70 77 84 91 98; do
    r=$(VOXE_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 15 --camera $cam 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['phases_ms']['bwd'],3))")
    line="$line $r"
  done
  echo "$line"
done
