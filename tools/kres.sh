#!/bin/bash
# Registers, scratch, spills, occupancy and LDS of every kernel of one translation unit of libart (the compiler's own resource remarks):
#   tools/kres.sh araytracingjourney_amd/csrc/art_trace.hip [extra hipcc flags]
# The fused frame and the AO tracer live at the 64-register edge (8 waves per SIMD): run this after touching them.
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | awk '
/remark: Function Name:/ {name=$5}
/ TotalSGPRs:/ {s=$4}
/ VGPRs:/ {v=$4}
/ScratchSize/ {sc=$5}
/Occupancy/ {oc=$5}
/VGPRs Spill/ {sp=$5}
/LDS Size/ {printf "%-14s vgpr %3d sgpr %3d scratch %5d spill %3d occ %d lds %6d  %s\n", "", v, s, sc, sp, oc, $6, name}' | while read -r line; do
  name=$(echo "$line" | awk '{print $NF}'); echo "$(echo "$line" | sed 's/ [^ ]*$//')  $(echo "$name" | c++filt | sed 's/art:://; s/(.*//; s/void //')"; done
