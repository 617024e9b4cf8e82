#!/bin/bash
# usage: tools/pmc_ta.sh <tag> [bench.py arguments ...] -- the vector-memory path's own counters (texture addresser, L1) for the kernels of a bench.py run:
# how busy the addresser is, what it waits for, the L1's latency and stalls.  Separate passes, kernel-trace only (as gpurun requires); few counters a pass
# (a block has two or four: a request for more aborts the profiled program) and every pass under its own timeout.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcta_$TAG
mkdir -p $OUT
SETS=("GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
      "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
      "TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
      "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum")
# PMC_SETS="A B;C D": other counter sets, one pass each (e.g. the scalar / instruction caches: "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES;SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES")
if [ -n "$PMC_SETS" ]; then IFS=";" read -ra SETS <<< "$PMC_SETS"; fi
i=0
for C in "${SETS[@]}"; do
  i=$((i+1))
  echo "pass $i: $C"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --plain --settle-seconds 0 --frames-in-flight 8 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "art::" not in k: continue
            k = k.split("(")[0][-64:]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            if len(next(iter(cs.values()))) >= 4: print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
