#!/bin/bash
# the driver's protocol (--steps 20 --warmup 5) and a long run for several ring depths (frames in flight), three short runs each: bash tools/ring_sweep.sh 2 3 4 6 16
mkdir -p gpurun_out
for f in "$@"; do
  for k in "20 5" "20 5" "20 5" "20 5" "1000 50"; do set -- $k
    python bench.py --plain --steps $1 --warmup $2 --frames-in-flight $f > gpurun_out/rs.json 2> gpurun_out/rs.err
    python -c "
import json; d=json.load(open('gpurun_out/rs.json')); print('ring $f', 'steps', d['steps'], round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms')"
  done
done
