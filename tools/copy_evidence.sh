#!/bin/bash
# copy what is to be judged from gpurun_out/<tag> (tools/gpu_round.sh) to profiles/<round>_*:   bash tools/copy_evidence.sh r06f r06
T=$1; R=$2; O=gpurun_out/$T
for f in $O/bench_*.json; do cp $f profiles/${R}_$(basename $f); done
for f in pytest_gpu.log smoke.log host.txt recon_bench.txt recon_bench_python_iteration.txt refine_bench.txt sh_bench.txt grid_passes.txt band_probe.txt fuzz_soak.txt fwd_identity_sweep.txt two_ranks_one_gpu_gloo.jsonl eight_ranks_one_gpu_gloo.json two_ranks_one_gpu_gloo_strong.json; do [ -f $O/$f ] && cp $O/$f profiles/${R}_$f; done
cp $O/prof/${T}_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
for k in headline recon refine grid; do cp $O/prof_$k/${T}_${k}_kernel_stats.csv profiles/${R}_${k}_kernel_stats.csv; done
ls profiles/${R}_* | wc -l
