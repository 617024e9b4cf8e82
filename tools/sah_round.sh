#!/bin/bash
# one GPU call for a change of the tree builder: the tree tests, then for each candidate library its build times and its ray rates against a base library
#   tools/sah_round.sh TAG [candidate.so ...]     (default candidate: libart.so; base: araytracingjourney_amd/libart_base.so, when it is there)
export GPU_MAX_HW_QUEUES=16
TAG=$1; shift; LIBS=${@:-libart.so}
out=gpurun_out/sah_$TAG.log; mkdir -p gpurun_out; : > $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "traversal_tree or tiny_scenes or frame_forms or sponza_frame or config4_bistro or wide_collapse or moves_every_frame or leaves_and_re_enters" >> $out 2>&1 || { tail -30 $out; exit 1; }
for L in $LIBS; do
  for s in sponza bistro; do
    echo "== build $s $L" >> $out
    ART_LIB_PATH=$PWD/araytracingjourney_amd/$L timeout -k 10 200 python tools/build_probe.py --scene $s --n 3 --tuning log=1 2>&1 | grep -v amdgpu.ids | tail -2 >> $out || exit 1
  done
  if [ -f araytracingjourney_amd/libart_base.so ]; then
  echo "== config 2 $L" >> $out; bash tools/ab_libs.sh araytracingjourney_amd/libart_base.so araytracingjourney_amd/$L --steps 1000 --warmup 50 >> $out 2>&1
  echo "== config 4 $L" >> $out; bash tools/ab_libs.sh araytracingjourney_amd/libart_base.so araytracingjourney_amd/$L --steps 600 --warmup 50 --scene bistro >> $out 2>&1
  fi
done
tail -60 $out
