#!/bin/bash
# config 5 (4K, 16-spp AO) for several builds of libart under one ArtTuning setting, alternating twice: bash tools/ao_libs_ab.sh "<tuning>" libart.so libart_x.so ...
T=$1; shift
mkdir -p gpurun_out
for i in 1 2; do for L in "$@"; do
  ART_LIB_PATH=$PWD/araytracingjourney_amd/$L python bench.py --plain --steps 60 --warmup 30 --width 3840 --height 2160 --ao 16 ${T:+--tuning $T} > gpurun_out/aoab.json 2> gpurun_out/aoab.err || { tail -5 gpurun_out/aoab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/aoab.json')); print('$L', '${T:-default}', round(d['value']), 'Mray/s', round(d['ms_per_step'], 4), 'ms', flush=True)"
done; done
