#!/bin/bash
# tools/build_variant.sh NAME -DMACRO... : raytracing_amd/libsrx_NAME.so = the library with srx_trace.hip rebuilt with the macros
set -e
cd "$(dirname "$0")/../raytracing_amd"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
  -Wall -Wno-unused-function -mllvm -structurizecfg-skip-uniform-regions=1 "$@" -c csrc/srx_trace.hip -o _obj/srx_trace_$name.o
objs=$(ls _obj/*.o | grep -v srx_trace)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj/srx_trace_$name.o -o libsrx_$name.so
echo built libsrx_$name.so
