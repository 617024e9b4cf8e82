#!/bin/bash
# one GPU call of the beam-walk study: step counts on the profiling build (tools/packet_prof.py), then frame times of the product build (tools/beam_probe.py)
#   tools/beam_round.sh TAG   ->  gpurun_out/beam_TAG.log
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/beam_$1.log; mkdir -p gpurun_out; : > $out
for t in "packet_wide=1" "packet_wide=4" "packet_wide=3,beam_fat=0.02" "packet_wide=3,beam_fat=0.05" "packet_wide=3,beam_fat=0.25" "packet_wide=3,beam_fat=-1"; do
  echo "== prof $t" >> $out
  ART_LIB_PATH=$PWD/araytracingjourney_amd/libart_prof.so timeout -k 10 120 python tools/packet_prof.py --tuning "$t" > gpurun_out/_pp.json 2>gpurun_out/_pp.err || { tail -5 gpurun_out/_pp.err >> $out; exit 1; }
  python - >> $out <<'PY'
import json
t = open("gpurun_out/_pp.json").read(); d = json.loads(t[t.index("{"):])
for k in ("primary", "shadow"):
    v = d[k]
    print(k, "walks", v["walks"], "node/walk %.2f tri/walk %.2f boxes/node %.2f hit-share %.2f fallback-or-mixed %d cyc/node %.0f cyc/tri %.0f setup %.0f" % (v["node_steps_per_walk"], v["triangle_steps_per_walk"], v["child_boxes_hit_per_node_step"], v["share_of_triangle_steps_with_a_hit"], v["mixed_octant_walks_and_fat_beams"], v["cycles_per_node_step"], v["cycles_per_triangle_step"], v["setup_cycles_per_walk"]))
PY
done
echo "== times" >> $out
timeout -k 10 300 python tools/beam_probe.py --frames 400 --tunings "packet_wide=1;packet_wide=3;packet_wide=4;packet_wide=5;packet_wide=3,beam_fat=0.05;packet_wide=3,beam_fat=0.02;packet_wide=3,beam_fat=-1;packet_wide=1" >> $out 2>&1
tail -40 $out
