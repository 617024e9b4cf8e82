#!/bin/bash
# the driver's protocol (--steps 20 --warmup 5) and the long run, for a list of ArtTuning settings: bash tools/short_run.sh "" "split_alpha=0.2" ...
for t in "$@"; do
  for k in "20 5" "20 5" "20 5" "1000 50"; do set -- $k
    python bench.py --plain --steps $1 --warmup $2 ${t:+--tuning $t} > gpurun_out/sr.json 2> gpurun_out/sr.err
    python -c "
import json; d=json.load(open('gpurun_out/sr.json')); print('${t:-default}', 'steps', d['steps'], round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms')"
  done
done
