#!/bin/bash
# what a refit is made of (rocprofv3 kernel trace of tools/refit_probe.py) and how the moving-model rate depends on the number of versions
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/refit_prof -o rp -- python3 $GRAFT_REPO_ROOT/tools/refit_probe.py --steps 100 > $out/rp.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $out/rp.log
for v in 1 2 3 4 6 8; do python tools/refit_probe.py --versions $v; done 2>&1 | tee $out/refit_versions.log
python tools/refit_probe.py --scene bistro --versions 3 2>&1 | tee -a $out/refit_versions.log
