export TMPDIR=/tmp
O=gpurun_out/r04_check9; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_hygiene_r03.py tests/test_hip_fuzz.py tests/test_hip_fullsize.py tests/test_hip_fused_step.py -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest_gpu.log
for rep in 1 2; do
for lib in "" variants/libvoxe_hip_nopcb.so; do
line="lib=${lib:-tree}:"
for cam in 3 12 26 40 0; do
r=$(VOXE_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --camera $cam --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['bwd'], d['ms_per_step_median'])")
line="$line  cam$cam bwd/step $r"
done
echo "$line" | tee -a $O/ab.txt
done; done
for lib in "" variants/libvoxe_hip_nopcb.so; do
for hw in 266 200 100; do
r=$(VOXE_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --image $hw --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step_median'], round(d['value']/1e6,1))")
echo "lib=${lib:-tree} image $hw fwd/bwd/step/Mrays $r" | tee -a $O/ab.txt
done; done
cd /tmp
for lib in "" variants/libvoxe_hip_nopcb.so; do
tag=pcb; [ -n "$lib" ] && tag=nopcb
VOXE_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-secondary > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$GRAFT_REPO_ROOT/$O/pmc_$tag/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:70]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if "bwd_tile" in k: print("$tag", k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
PY
done
