export TMPDIR=/tmp
O=gpurun_out/r04_check6; mkdir -p $O
VOXE_HIP_LIB=variants/libvoxe_hip_seg16.so timeout 600 python tools/band_probe.py 2>/dev/null > $O/band_probe_seg16.txt; head -7 $O/band_probe_seg16.txt
for rep in 1 2; do
for lib in "" variants/libvoxe_hip_seg16.so; do
for cam in 3 12 26; do
r=$(VOXE_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --camera $cam --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step_median'], d['ms_per_step'])")
echo "lib=${lib:-tree} cam$cam fwd/bwd/median/mean $r" | tee -a $O/ab.txt
done
r=$(VOXE_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --image 266 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step_median'], d['ms_per_step'])")
echo "lib=${lib:-tree} image266 fwd/bwd/median/mean $r" | tee -a $O/ab.txt
done; done
