#!/bin/bash
# forward / backward / step ms per camera for library variants, optional extra bench flags:  bash tools/ab_cam_lib2.sh "<cameras>" "<tags>" [bench flags]
cams=$1; tags=$2; shift 2
for cam in $cams; do line="cam $cam:"; for tag in $tags; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  r=$(VOXE_HIP_LIB=$lib python bench.py --camera $cam --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step'])")
  line="$line  $tag $r"; done; echo "$line"; done
