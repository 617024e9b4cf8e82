#!/bin/bash
# the moving-model leg of bench.py (a model moved before every frame) for several ArtTuning settings, e.g. the depth of the ring of structure versions:
#   bash tools/versions_ab.sh <steps> "as_versions=3" "as_versions=4" "" ...
S=$1; shift
mkdir -p gpurun_out
for i in 1 2; do for t in "$@"; do
  python bench.py --steps $S --warmup 5 --no-cpu-baseline ${t:+--tuning $t} > gpurun_out/vab.json 2> gpurun_out/vab.err || { tail -5 gpurun_out/vab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/vab.json')); m=d['moving_model']; print('${t:-default}', 'static', round(d['value']), 'moving', round(m['value']), 'Mray/s', round(m['ms_per_step'], 4), 'ms  refit_ms', m.get('refit_ms'), flush=True)"
done; done
