#!/bin/bash
# kernel timeline of the reconstruction iteration (who overlaps whom):  gpurun -- bash tools/trace_recon.sh <outdir>
# -> gpurun_out/<outdir>/recon_trace.csv (kernel, queue, start, end of the last iterations)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06y}; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_recon -o recon -- python $GRAFT_REPO_ROOT/tools/recon_bench.py 20 > $OUT/trace_recon.log 2>&1)
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/trace_recon/**/recon_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-150:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[1] + "/recon_trace.csv", "w") as o:
    o.write("kernel,queue,stream,start_us,end_us,dur_us\n")
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        o.write(f'{r["Kernel_Name"].replace("void ", "").replace("voxe::", "").split("(")[0][:40]},{r.get("Queue_Id", "")},{r.get("Stream_Id", "")},{s / 1e3:.1f},{e / 1e3:.1f},{(e - s) / 1e3:.1f}\n')
PY
