#!/bin/bash
# the shader clock rocm-smi reports while bench.py keeps the GPU busy (config 2, a long timed region), and idle before / after:
#   bash tools/clock_under_load.sh [steps=30000] [bench.py arguments ...]
N=${1:-30000}; shift
mkdir -p gpurun_out
echo "idle:"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
python bench.py --plain --steps $N --warmup 50 --settle-seconds 0 "$@" > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
P=$!
sleep 6   # scene build + settle
for i in 1 2 3 4 5 6 7 8; do
  kill -0 $P 2>/dev/null || break
  echo "under load ($i):"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -3
  sleep 0.4
done
wait $P
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('bench:', round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms per frame')"
