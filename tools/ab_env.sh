#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ... [-- bench args]: one default bench.py run per value of the environment variable VAR
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_${VAR}_$v.log 2>&1
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${VAR}_$v.log").read().strip().splitlines()[-1])
print("$VAR=$v", round(d["value"], 1), "Mray/s", round(d["ms_per_step"], 4), "ms/frame", "alone", round(d["stage_ms_one_frame_alone"]["frame_ms"], 3))
PY
done
