#!/bin/bash
# the counting build of the packet walks beside the product library: araytracingjourney_amd/libart_prof.so (tools/packet_prof.py loads it through ART_LIB_PATH)
cd "$(dirname "$0")/.." && mkdir -p /tmp/prof && make -C araytracingjourney_amd/csrc > /dev/null &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -DART_PACKET_PROF -c araytracingjourney_amd/csrc/art_trace.hip -o /tmp/prof/art_trace.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o araytracingjourney_amd/libart_prof.so /tmp/prof/art_trace.o $(ls araytracingjourney_amd/csrc/*.o | grep -v art_trace.o) -lz -ldl
