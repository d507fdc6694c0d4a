#!/bin/bash
# two PMC passes (SQ counters) of bench.py under the caller's environment, printed per voxe kernel:  [ENV=..] bash tools/pmc_quick.sh [bench flags]
export TMPDIR=/tmp
OUT=/tmp/pmcq; rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-secondary $BENCH_FLAGS > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmcq/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("voxe::", "")[:60]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "render_" in k or "region_" in k:
        print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
PY
