#!/bin/bash
# A/B of the traversal structures on bench config 2: ART_BVH=<primary><shadow> with 1 = quantised binary, 2 = binary, 4 = wide quantised
for k in "$@"; do
  ART_BVH=$k python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$k.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$k.log").read().strip().splitlines()[-1]); print("ART_BVH=$k", round(d["value"]), "Mray/s", round(d["ms_per_step"],4), {a:round(b,4) for a,b in d["stage_ms"].items()}, round(d["build_ms"],1))
except Exception as e: print("$k", e, open("gpurun_out/ab_$k.log").read()[-800:])
PY
done
