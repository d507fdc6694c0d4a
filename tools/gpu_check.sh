#!/bin/bash
# quick GPU validation: the GPU test suite + the driver's bench command line and the long one (same lease)
export TMPDIR=/tmp
TAG=${1:-check}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_err.log | tail -1 > $O/bench_400_driver.json
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_400_long.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_400_driver2.json
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2),'Mrays/s', d['ms_per_step'],'ms median',d.get('ms_per_step_median'),'min',d.get('ms_per_step_min'), d['roofline']['phases_ms'], d['config'].get('untimed_warmup_ms'))" 2>&1)"; done
python - <<PY
import json
d=json.load(open("$O/bench_400_driver.json"))
s=d["secondary"]
print("100x100", s["value"]/1e6, s["ms_per_step"], "multi", s["multi_view"]["value"]/1e6)
for k,v in s["views"]["cameras"].items(): print("view",k,round(v["value"]/1e6,1),v["ms_per_step"],v["fwd_ms"],v["bwd_ms"])
print("views mean", s["views"]["mean_rays_per_s_incl_headline_camera"]/1e6)
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("passes_s"), "gpu", d["gpu_baseline"].get("value"))
PY
