#!/bin/bash
# two PMC passes (SQ counters) of an arbitrary command, printed per voxe kernel (millions per launch):
#   bash tools/pmc_cmd.sh python tools/sh_fwd_window.py 2 3
export TMPDIR=/tmp
OUT=/tmp/pmcc; rm -rf $OUT; mkdir -p $OUT
CMD=("$@")
for i in "${!CMD[@]}"; do case "${CMD[$i]}" in tools/*|bench.py) CMD[$i]="$GRAFT_REPO_ROOT/${CMD[$i]}";; esac; done
cd /tmp
run() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- "${CMD[@]}" > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmcc/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("voxe::", "").replace("(anonymous namespace)::", "")[:48]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "render_fwd" in k or "region_" in k or "render_bwd" in k:
        print(k, {c.replace("SQ_", ""): round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
PY
