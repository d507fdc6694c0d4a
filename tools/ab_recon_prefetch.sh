#!/bin/bash
# A/B of voxe_recon_prefetch's schedule knobs (GPU box): fork point (behind the forward / at the step's start) x side-stream priority
# -> gpurun_out/$1/ab_recon_prefetch.txt     build the variants first: VARIANT_SRC=voxe_api.hip python tools/variants.py f0lo -DVOXE_RECON_FORK=0 ...
OUT=gpurun_out/${1:-r06y}; mkdir -p $OUT
F=$OUT/ab_recon_prefetch.txt; : > $F
run() { echo "== $1" >> $F; shift; for i in 1 2; do env "$@" python tools/recon_bench.py 80 2>&1 | grep -E 'reconstruction iteration|kernel phases' >> $F; done; }
run "no hint" RECON_NO_PREFETCH=1
run "hint: shipped (prefetch forks behind the forward, low-priority side stream)" X=1
for v in f0 f2 f3; do [ -f variants/libvoxe_hip_$v.so ] && run "hint: variant $v" VOXE_HIP_LIB=variants/libvoxe_hip_$v.so; done
cat $F
