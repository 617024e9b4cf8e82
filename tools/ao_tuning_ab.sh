#!/bin/bash
# config 5 (4K, 16-spp AO) for a list of ArtTuning settings, alternating twice: bash tools/ao_tuning_ab.sh "ao_walk=6" "" "trace_refill=4" ...
mkdir -p gpurun_out
for i in 1 2; do for t in "$@"; do
  python bench.py --plain --steps 60 --warmup 30 --width 3840 --height 2160 --ao 16 ${t:+--tuning $t} > gpurun_out/aoab.json 2> gpurun_out/aoab.err || { tail -5 gpurun_out/aoab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/aoab.json')); print('${t:-default}', round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms', flush=True)"
done; done
