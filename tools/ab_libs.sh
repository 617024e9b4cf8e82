#!/bin/bash
# A/B of two builds of libart on one box, alternating: bash tools/ab_libs.sh <base.so> <new.so> [bench.py arguments ...]   (default: config 2, 1000 steps)
A=$1; B=$2; shift 2
ARGS=${@:---steps 1000 --warmup 50}
mkdir -p gpurun_out
for i in 1 2; do for L in $A $B; do
  ART_LIB_PATH=$PWD/$L python bench.py --plain $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$L', round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms')"
done; done
