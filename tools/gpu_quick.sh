export TMPDIR=/tmp
O=gpurun_out/r04_check15; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest_gpu.log
for rep in 1 2; do
for lib in "" variants/libvoxe_hip_head.so; do
line="lib=${lib:-tree}:"
for cam in 3 12 26; do
r=$(VOXE_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --camera $cam --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step_median'])")
line="$line  cam$cam fwd/bwd/step $r"
done
echo "$line" | tee -a $O/ab.txt
done; done
