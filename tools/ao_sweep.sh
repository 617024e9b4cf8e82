#!/bin/bash
# config 5 (4K, 16 AO rays per hit pixel) for a list of ArtTuning settings: bash tools/ao_sweep.sh "" "trace_leaf_batch=16" "trace_refill=16,trace_leaf_batch=24" ...
for t in "$@"; do
  python bench.py --plain --width 3840 --height 2160 --ao 16 --steps 40 --warmup 8 ${t:+--tuning $t} > gpurun_out/ao_sweep.json 2> gpurun_out/ao_sweep.err
  python -c "
import json; d=json.load(open('gpurun_out/ao_sweep.json')); print(d['tuning'], round(d['value']), 'Mray/s', round(d['ms_per_step'], 3), 'ms')"
done
