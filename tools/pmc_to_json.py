"""gpurun_out/pmc/*/*counter_collection.csv (written by tools/gpu_pmc.sh) -> profiles/<tag>_pmc_summary.json
(per-kernel averages per launch; bench.py reads FETCH_SIZE / WRITE_SIZE of the dominant kernel from it)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vox-e_amd", "voxe_hip"))
import build as _build  # noqa: E402
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if "voxe" in name:
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {
    # the kernels these counters belong to: bench.py only uses the summary when this equals the hash of ITS tree
    "source_hash": _build.source_hash(),
    "source": "tools/gpu_pmc.sh: rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group) -- python bench.py "
              "--steps 5 --warmup 2; averages per launch; FETCH_SIZE / WRITE_SIZE in KiB. MI355X_MICROARCH.md: FETCH_SIZE "
              "reports 1/2 of the bytes of wide coalesced reads on gfx950 -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
    "config": "160^3 random grid, 400x400, S=256, jitter on",
    "kernels": {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in agg.items()},
}
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print(path, len(out["kernels"]), "kernels")
