#!/bin/bash
# an A/B build of libart beside the product library: tools/build_variant.sh NAME "-DFLAG ..."  ->  araytracingjourney_amd/libart_NAME.so (art_trace.hip compiled with the flags, the other objects shared)
cd "$(dirname "$0")/.." && mkdir -p /tmp/variant_$1 && make -C araytracingjourney_amd/csrc > /dev/null &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 $2 -c araytracingjourney_amd/csrc/art_trace.hip -o /tmp/variant_$1/art_trace.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o araytracingjourney_amd/libart_$1.so /tmp/variant_$1/art_trace.o $(ls araytracingjourney_amd/csrc/*.o | grep -v art_trace.o) -lz -ldl
