#!/bin/bash
# kernel timeline (last N kernels, gaps between them) of any command:  gpurun -- bash tools/trace_cmd.sh <outdir> <N> <command ...>
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; N=$2; shift; shift; mkdir -p $OUT
rm -rf /tmp/trace_cmd
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_cmd -o t -- "$@" > $OUT/trace_cmd.log 2>&1)
python - "$OUT" "$N" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/trace_cmd/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
with open(sys.argv[1] + "/trace.csv", "w") as o:
    o.write("kernel,queue,start_us,end_us,dur_us,gap_before_us(same queue)\n")
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = r.get("Queue_Id", "")
        gap = "" if q not in last_end else f"{(s - last_end[q]) / 1e3:.1f}"
        last_end[q] = e
        o.write(f'{r["Kernel_Name"].replace("void ", "").replace("voxe::", "").split("(")[0][:48]},{q},{s / 1e3:.1f},{e / 1e3:.1f},{(e - s) / 1e3:.1f},{gap}\n')
PY
