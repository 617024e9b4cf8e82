#!/bin/bash
# config 5 (4K, 16-spp AO) for candidate libraries against the product build, alternating   tools/ao_ab.sh lib...
export GPU_MAX_HW_QUEUES=16
for L in "$@"; do bash tools/ab_libs.sh araytracingjourney_amd/libart.so araytracingjourney_amd/$L --steps 60 --warmup 30 --width 3840 --height 2160 --ao 16; done
