#!/bin/bash
# the moving-model leg of bench.py for several builds of libart under one ArtTuning setting, alternating: bash tools/versions_libs_ab.sh <steps> "<tuning>" libart.so libart_A.so ...
S=$1; T=$2; shift 2
mkdir -p gpurun_out
for i in 1 2 3; do for L in "$@"; do
  ART_LIB_PATH=$PWD/araytracingjourney_amd/$L python bench.py --steps $S --warmup 5 --no-cpu-baseline ${T:+--tuning $T} > gpurun_out/vab.json 2> gpurun_out/vab.err || { tail -5 gpurun_out/vab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/vab.json')); m=d['moving_model']; print('$L', '${T:-default}', 'static', round(d['value']), 'moving', round(m['value']), 'Mray/s', round(m['ms_per_step'], 4), 'ms', flush=True)"
done; done
