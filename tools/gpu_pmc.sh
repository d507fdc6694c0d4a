#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only): HBM fetch / write sizes and SQ/LDS counters
# for the bench kernels.   gpurun --timeout 1200 -- bash tools/gpu_pmc.sh
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
run() { # name, counters...
  local name=$1; shift
  # r06: the timed steps cycle through the bench's camera set (bench.py --views, default 20): exactly one launch per view, no
  # clock-settling run, so that the per-launch averages below are averages over the SAME views the driver's line times
  VOXE_BENCH_PRE_WARM_MS=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/$name.log 2>&1
  echo "== $name rc=$? =="; ls $OUT/$name | head
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run tccea TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
# r06: dynamic instruction mix for the issue model (tools/isa_issue_model.py prices the classes; bench.py: issue_model())
run mix1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64
run mix2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES
run sq3 SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INSTS_GDS
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ_|TCC_|TCP_|TA_|GRBM_)[A-Za-z0-9_]+" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
python - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc"
for f in sorted(glob.glob(out + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("##", os.path.basename(f))
    for k, d in agg.items():
        if "voxe" not in k: continue
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
