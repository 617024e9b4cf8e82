#!/bin/bash
# ray rates (configs 2 and 4) and build times of candidate libraries against the base   tools/sah_round4.sh TAG lib...
export GPU_MAX_HW_QUEUES=16
TAG=$1; shift
out=gpurun_out/sah_$TAG.log; mkdir -p gpurun_out; : > $out
for L in "$@"; do
  for s in sponza bistro; do
    echo "== build $s $L" >> $out
    ART_LIB_PATH=$PWD/araytracingjourney_amd/$L timeout -k 10 200 python tools/build_probe.py --scene $s --n 3 2>&1 | grep -v amdgpu.ids | tail -1 >> $out || exit 1
  done
  echo "== config 2 $L" >> $out; bash tools/ab_libs.sh araytracingjourney_amd/libart_base.so araytracingjourney_amd/$L --steps 1000 --warmup 50 >> $out 2>&1
  echo "== config 4 $L" >> $out; bash tools/ab_libs.sh araytracingjourney_amd/libart_base.so araytracingjourney_amd/$L --steps 600 --warmup 50 --scene bistro >> $out 2>&1
done
tail -40 $out
