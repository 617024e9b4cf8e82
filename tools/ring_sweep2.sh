#!/bin/bash
# ring depth 8 against 16 on the other configs (long runs), alternating
mkdir -p gpurun_out
run() { python bench.py --plain "$@" > gpurun_out/rs.json 2> gpurun_out/rs.err; python -c "
import json; d=json.load(open('gpurun_out/rs.json')); print('$*', round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms')"; }
for i in 1 2; do for f in 8 16; do
  run --steps 300 --warmup 30 --width 3840 --height 2160 --lights 4 --frames-in-flight $f
  run --steps 600 --warmup 50 --scene bistro --frames-in-flight $f
  run --steps 60 --warmup 30 --width 3840 --height 2160 --ao 16 --frames-in-flight $f
done; done
