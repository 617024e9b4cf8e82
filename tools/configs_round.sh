#!/bin/bash
# usage (GPU box): bash tools/configs_round.sh <tag>  -- the other BASELINE configs on one GPU, plain runs: 3 (4K, 4 lights), 4 (bistro 1080p), 5 (4K + 16-spp AO)
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
run() { name=$1; shift; python3 $R/bench.py --plain "$@" > $OUT/$name.log 2>&1; grep '^{' $OUT/$name.log | tail -1 > $OUT/${TAG}_$name.json; python3 -c "import json; d=json.load(open('$OUT/${TAG}_$name.json')); print('$name', round(d['value']), 'Mray/s', round(d['ms_per_step'],4), 'ms/frame', 'build', round(d['build_ms'],1))"; }
run config3 --width 3840 --height 2160 --lights 4 --steps 300 --warmup 30
run config4 --scene bistro --steps 600 --warmup 50
run config5 --width 3840 --height 2160 --ao 16 --steps 60 --warmup 8
