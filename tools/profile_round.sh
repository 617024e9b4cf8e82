#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh <tag>
# Everything tools/roofline.py needs for one build, under gpurun_out/prof_<tag>/ (copy the <tag>_* files into profiles/ afterwards):
#   <tag>_bench_line.json                 the un-profiled `bench.py --steps 1000` line (with the CPU baseline and the extra legs)
#   <tag>_kernel_stats_bench_c2.csv       rocprofv3 --kernel-trace --stats of `bench.py --steps 1000 --plain`
#   <tag>_bench_line_under_rocprof.json   that run's own line
#   <tag>_pmc.txt                         tools/pmc.sh: PMC passes (kernel-trace only, one counter group per pass), means per kernel
#   <tag>_config{3,4,5}.json, <tag>_pmc_c{3,4,5}.txt   the other BASELINE configs: plain bench line + the counter passes their rooflines need
# then: python tools/roofline.py --tag <tag> --set-current (profiles/current_pmc.json: what bench.py's roofline object reads, keyed by workload)
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PARTS="c2" or "others" runs one half only (a gpurun call is at most 20 minutes)
if [ "${PARTS:-c2 others}" != "others" ]; then
python3 $R/bench.py --steps 1000 --warmup 50 > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/${TAG}_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 1000 --warmup 50 --plain > $OUT/stats.log 2>&1
cp "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_bench_c2.csv
grep '^{' $OUT/stats.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
bash $R/tools/pmc.sh $TAG > $OUT/${TAG}_pmc.txt 2> $OUT/pmc.err
rm -rf $OUT/stats $R/gpurun_out/pmc_$TAG/p*/   # the raw traces are large; the summaries above are what is kept
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench20.log 2>&1; grep '^{' $OUT/bench20.log | tail -1 > $OUT/${TAG}_bench_line_steps20.json   # the driver's protocol
echo "config 2 done"
fi
if [ "${PARTS:-c2 others}" == "c2" ]; then exit 0; fi
# the other BASELINE configs on one GPU: a plain bench line and the counter passes a roofline needs, each (<tag>_configN.json, <tag>_pmc_cN.txt)
other() { n=$1; steps=$2; shift 2; python3 $R/bench.py --plain --steps $steps --warmup 30 "$@" > $OUT/config$n.log 2>&1; grep '^{' $OUT/config$n.log | tail -1 > $OUT/${TAG}_config$n.json
          PMC_SHORT=1 bash $R/tools/pmc.sh ${TAG}_c$n "$@" > $OUT/${TAG}_pmc_c$n.txt 2> $OUT/pmc_c$n.err; rm -rf $R/gpurun_out/pmc_${TAG}_c$n/p*/; echo "config $n done"; }
other 3 300 --width 3840 --height 2160 --lights 4
other 4 600 --scene bistro
other 5 60 --width 3840 --height 2160 --ao 16
echo "profile_round $TAG done"
