#!/bin/bash
# A/B of library variants (tools/variants.py) on the reconstruction iteration and the multi-view 100x100 step.
#   gpurun -- bash tools/ab_recon.sh tag1 tag2 ...   ("base" = the in-tree library)
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  r=$(VOXE_HIP_LIB=$lib python tools/recon_bench.py 30 2>/dev/null | grep -E "reconstruction iteration|kernel phases" | sed -e 's/reconstruction iteration.*spec+diffuse): //' | tr '\n' ' ')
  echo "$tag: $r"
done
