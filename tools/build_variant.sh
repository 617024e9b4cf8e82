#!/bin/bash
# an A/B build of libart beside the product library: tools/build_variant.sh NAME "-DFLAG ..." [file.hip]  ->  araytracingjourney_amd/libart_NAME.so
# (one translation unit -- art_trace.hip unless named -- compiled with the flags, the other objects shared with the product build)
F=${3:-art_trace.hip}; O=${F%.hip}.o
cd "$(dirname "$0")/.." && mkdir -p /tmp/variant_$1 && make -C araytracingjourney_amd/csrc > /dev/null &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 $2 -c araytracingjourney_amd/csrc/$F -o /tmp/variant_$1/$O &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o araytracingjourney_amd/libart_$1.so /tmp/variant_$1/$O $(ls araytracingjourney_amd/csrc/*.o | grep -v "/$O") -lz -ldl
