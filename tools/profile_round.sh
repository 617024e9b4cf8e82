#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh <tag>
# Everything tools/roofline.py needs for one build, under gpurun_out/prof_<tag>/ (copy the <tag>_* files into profiles/ afterwards):
#   <tag>_bench_line.json                 the un-profiled `bench.py --steps 1000` line (with the CPU baseline and the extra legs)
#   <tag>_kernel_stats_bench_c2.csv       rocprofv3 --kernel-trace --stats of `bench.py --steps 1000 --plain`
#   <tag>_bench_line_under_rocprof.json   that run's own line
#   <tag>_pmc.txt                         tools/pmc.sh: PMC passes (kernel-trace only, one counter group per pass), means per kernel
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1000 --warmup 50 > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/${TAG}_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 1000 --warmup 50 --plain > $OUT/stats.log 2>&1
cp "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_bench_c2.csv
grep '^{' $OUT/stats.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
bash $R/tools/pmc.sh $TAG > $OUT/${TAG}_pmc.txt 2> $OUT/pmc.err
rm -rf $OUT/stats $R/gpurun_out/pmc_$TAG/p*/   # the raw traces are large; the summaries above are what is kept
echo "profile_round $TAG done"
