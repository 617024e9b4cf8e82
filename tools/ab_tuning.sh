#!/bin/bash
# usage: tools/ab_tuning.sh KEY v1 v2 ... [-- bench args]: one plain bench.py run per value of an ArtTuning field (include/art.h: frame_waves,
# block_order, tree_builder, frame_form, packet_wide, split_alpha ...), e.g.  tools/ab_tuning.sh frame_waves 6 7 8 -- --steps 400
KEY=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
mkdir -p gpurun_out
for v in "${VALS[@]}"; do
  python bench.py --plain --tuning $KEY=$v "$@" > gpurun_out/ab_${KEY}_$v.log 2>&1
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/ab_${KEY}_$v.log") if l.startswith("{")][-1])
    print("$KEY=$v", round(d["value"], 1), "Mray/s", round(d["ms_per_step"], 4), "ms/frame")
except Exception as e:
    print("$KEY=$v", e, open("gpurun_out/ab_${KEY}_$v.log").read()[-600:])
PY
done
