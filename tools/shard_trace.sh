#!/bin/bash
# kernel trace (concurrency, launch durations) and a few counters of one GPU's share of a sharded frame against the whole frame
# usage: tools/shard_trace.sh   (on the GPU box; writes gpurun_out/shard_trace.txt)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
OUT=$R/gpurun_out/shard_trace; mkdir -p $OUT
{
for cfg in "8 3 20 22" "1 0 16 16"; do
  set -- $cfg; export PG=$1 PK=$2 PF=$3 GPU_MAX_HW_QUEUES=$4
  echo "== shards $PG, shard $PK, $PF frames in flight, $GPU_MAX_HW_QUEUES queues"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$PG -- python3 $R/tools/shard_probe3.py 2>/dev/null | tail -1
  python3 $R/tools/overlap.py $OUT/kt_$PG
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${PG}_$i -- python3 $R/tools/shard_probe3.py > /dev/null 2>&1 || echo "pmc pass $i failed"
    python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_${PG}_$i/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_frame" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print({c: round(sum(v) / len(v), 1) for c, v in agg.items()}, "launches", len(next(iter(agg.values()))) if agg else 0)
PY
  done
done
} > $R/gpurun_out/shard_trace.txt 2>&1
