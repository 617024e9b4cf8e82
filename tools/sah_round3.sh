#!/bin/bash
# build times only (both scenes) of candidate libraries   tools/sah_round3.sh TAG lib...
TAG=$1; shift
out=gpurun_out/sah_$TAG.log; mkdir -p gpurun_out; : > $out
for L in "$@"; do for s in sponza bistro; do
  echo "== build $s $L" >> $out
  ART_LIB_PATH=$PWD/araytracingjourney_amd/$L timeout -k 10 200 python tools/build_probe.py --scene $s --n 3 --tuning log=1 2>&1 | grep -v amdgpu.ids | tail -2 >> $out || exit 1
done; done
tail -40 $out
