#!/bin/bash
# the driver's protocol (--steps 20 --warmup 5) and the long run for several ring depths: bash tools/short_run_f.sh 4 8 12 16
for f in "$@"; do
  for k in "20 5" "20 5" "1000 50"; do set -- $k
    python bench.py --plain --steps $1 --warmup $2 --frames-in-flight $f > gpurun_out/sr.json 2> gpurun_out/sr.err
    python -c "
import json; d=json.load(open('gpurun_out/sr.json')); print('F=$f', 'steps', d['steps'], round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms')"
  done
done
