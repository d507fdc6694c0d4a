#!/bin/bash
# A/B of library variants on the 100x100 step (secondary line of bench.py):  bash tools/ab_small.sh base tag ...
for tag in "$@"; do
  lib=""; [ "$tag" != "base" ] && lib=variants/libvoxe_hip_$tag.so
  VOXE_HIP_LIB=$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab_small.json
  python - "$tag" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_small.json"))["secondary"]
print(sys.argv[1], "100x100:", round(d["value"] / 1e6, 2), "M rays/s", d["ms_per_step"], "ms fwd", d["fwd_ms"], "bwd", d["bwd_ms"],
      "| 8 cameras:", round(d["multi_view"]["value"] / 1e6, 2), "M rays/s")
PY
done
