#!/bin/bash
# bench.py --plain at the driver's protocol (twice) and over 1 000 steps for a list of ArtTuning settings: bash tools/tuning_ab.sh "wave_plan=1" "wave_plan=1,packet_budget=192" ...
mkdir -p gpurun_out
for t in "$@"; do
  for k in "20 5" "20 5" "1000 50"; do set -- $k
    python bench.py --plain --steps $1 --warmup $2 ${t:+--tuning $t} > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('${t:-default}', 'steps', d['steps'], round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms', flush=True)"
  done
done
