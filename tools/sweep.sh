#!/bin/bash
# usage: tools/sweep.sh "VAR=val VAR2=val" ...   (each argument is one environment for bench.py config 2)
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 40 --warmup 5 --no-cpu-baseline $SWEEP_ARGS > gpurun_out/sweep_$i.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sweep_$i.log").read().strip().splitlines()[-1]); print("$envs", round(d["value"]), "Mray/s", round(d["ms_per_step"],4), {a:round(b,4) for a,b in d["stage_ms"].items()})
except Exception as e: print("$envs", e, open("gpurun_out/sweep_$i.log").read()[-600:])
PY
done
