#!/bin/bash
# SQ counter passes of the reconstruction iteration (tools/recon_bench.py), printed per region kernel (millions per launch):  gpurun -- bash tools/pmc_recon.sh
export TMPDIR=/tmp
OUT=/tmp/pmcr; rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/tools/recon_bench.py 6 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
run sq3 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmcr/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("voxe::", "")[:40]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "region_" in k:
        print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
PY
