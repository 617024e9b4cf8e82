#!/bin/bash
# one GPU call: what a build costs and where (phase log to stderr, then a kernel trace of five builds)   tools/build_round.sh TAG
out=gpurun_out/build_$1.log; mkdir -p gpurun_out; : > $out
for s in sponza bistro; do
  echo "== $s" >> $out
  timeout -k 10 200 python tools/build_probe.py --scene $s --n 4 --tuning log=1 >> $out 2>&1 || exit 1
done
cd /tmp && export TMPDIR=/tmp
for s in sponza bistro; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/buildtrace_$1_$s -- python3 $GRAFT_REPO_ROOT/tools/build_probe.py --scene $s --n 4 > $GRAFT_REPO_ROOT/gpurun_out/buildtrace_$1_$s.log 2>&1 || exit 1
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/buildtrace_$1_$s/*/*kernel_stats.csv | head -1)
  echo "== kernel stats $s" >> $GRAFT_REPO_ROOT/$out; head -30 $f | cut -c1-160 >> $GRAFT_REPO_ROOT/$out
done
tail -90 $GRAFT_REPO_ROOT/$out
