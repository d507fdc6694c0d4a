#!/bin/bash
# kernel timeline of the headline step (gaps between the kernels of a step):  gpurun -- bash tools/trace_bench.sh <outdir>
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06y}; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gpu-baseline --no-secondary --steps 20 --warmup 3 $TRACE_FLAGS > $OUT/trace_bench.log 2>&1)
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/trace_bench/**/bench_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: the last run of (fwd, combine, bwd, adam) quadruples
idx = [i for i, r in enumerate(rows) if "render_bwd_tile4_kernel" in r["Kernel_Name"]]
lo = max(0, idx[-20] - 3)
rows = rows[lo: idx[-1] + 3]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[1] + "/bench_trace.csv", "w") as o:
    o.write("kernel,queue,start_us,end_us,dur_us,gap_before_us\n")
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        gap = "" if prev is None else f"{(s - prev) / 1e3:.1f}"
        prev = e
        o.write(f'{r["Kernel_Name"].replace("void ", "").replace("voxe::", "").split("(")[0][:40]},{r.get("Queue_Id", "")},{s / 1e3:.1f},{e / 1e3:.1f},{(e - s) / 1e3:.1f},{gap}\n')
PY
