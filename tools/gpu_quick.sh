export TMPDIR=/tmp
O=gpurun_out/r04_check2; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_two_ranks_gpu.py -q -m gpu 2>&1 | tail -8 | tee $O/pytest_gpu.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_400_driver$i.json
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_400_long$i.json
done
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2),'Mrays/s', d['ms_per_step'],'ms median',d.get('ms_per_step_median'),'min',d.get('ms_per_step_min'), d['ms_first_steps'], d['roofline']['phases_ms']['bwd'])" 2>&1)"; done
