#!/bin/bash
# usage: tools/pmc_ab.sh <tag> "<tuning A>" "<tuning B>" ... -- the SQ counters of the frame kernel for several ArtTuning settings, one rocprofv3 pass per counter set and setting
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcab_$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
      "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_FLAT")
n=0
for T in "$@"; do
  n=$((n+1)); i=0
  for C in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/t${n}p$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --plain --settle-seconds 0 --frames-in-flight 8 --tuning "$T" > $OUT/t${n}p$i.log 2>&1 || echo "setting $n pass $i failed"
  done
done
python3 - "$@" <<PY
import csv, glob, collections, sys
for n, t in enumerate(sys.argv[1:], 1):
    tot = {}
    for d in sorted(glob.glob("$OUT/t%dp*/" % n)):
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "k_frame<" not in k: continue
                agg[k.split("(")[0][-48:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            best = max(agg.items(), key=lambda kv: len(next(iter(kv[1].values()))), default=None)
            if best: tot.update({c: round(sum(v) / len(v)) for c, v in best[1].items()}); name = best[0]
    print(t, name if tot else "", tot)
PY
