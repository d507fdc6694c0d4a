#!/bin/bash
# forward / backward ms per camera for library variants:  bash tools/ab_cam_fwdlib.sh "<images>" "<cameras>" "<tags>"   (base = in-tree)
for img in $1; do for cam in $2; do line="image $img cam $cam:"; for tag in $3; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  r=$(VOXE_HIP_LIB=$lib python bench.py --image $img --camera $cam --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'])")
  line="$line  $tag $r"; done; echo "$line"; done; done
