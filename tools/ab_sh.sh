#!/bin/bash
# A/B of library variants on the SH degree 1-3 backward (tools/sh_bench.py):  gpurun -- bash tools/ab_sh.sh tag1 tag2 ...
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  echo "$tag: $(VOXE_HIP_LIB=$lib python tools/sh_bench.py 160 400 ${DEGS:-13} 2>/dev/null | sed -E 's/.*S=256: ([0-9.]+) ms.*bwd ([0-9.]+) memset.*/step \1 bwd \2 |/' | tr '\n' ' ')"
done
