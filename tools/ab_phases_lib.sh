#!/bin/bash
# backward ms per camera / image size for library variants x VOXE_TILE_PHASES:  bash tools/ab_phases_lib.sh "<tags>" ["<phases values>"]
for tag in $1; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  for ph in ${2:-1 0}; do
    line="$tag phases=$ph:"
    for cam in 3 12 58 88; do
      r=$(VOXE_HIP_LIB=$lib VOXE_TILE_PHASES=$ph python bench.py --camera $cam --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['bwd'])")
      line="$line cam$cam $r"
    done
    for img in 100 200; do
      r=$(VOXE_HIP_LIB=$lib VOXE_TILE_PHASES=$ph python bench.py --image $img --camera 3 --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['bwd'])")
      line="$line img$img $r"
    done
    r=$(VOXE_HIP_LIB=$lib VOXE_TILE_PHASES=$ph python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['ms_per_step'])")
    echo "$line  views-mean $r"
  done
done
