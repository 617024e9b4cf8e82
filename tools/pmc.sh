#!/bin/bash
# usage: tools/pmc.sh <tag> [bench.py arguments ...] -- collects PMC counters for bench.py (separate passes; kernel-trace only, as gpurun requires)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
# PMC_SHORT=1: the four passes a roofline needs (wave / lane cycles, instruction counts, HBM reads, HBM writes), without the cache passes
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT")
if [ -z "$PMC_SHORT" ]; then SETS+=("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"); fi
SETS+=("FETCH_SIZE" "WRITE_SIZE")
for C in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --plain --settle-seconds 0 --frames-in-flight 8 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
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
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
