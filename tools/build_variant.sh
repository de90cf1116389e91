#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO...] : raytracing_amd/libsrx_NAME.so = the library with one translation unit
# (SRC=srx_trace.hip by default) rebuilt with the macros; select it at run time with SRX_LIB=<path>. Experiments only.
set -e
cd "$(dirname "$0")/../raytracing_amd"
name=$1; shift
src=${SRC:-srx_trace.hip}
extra=""
[ "$src" = srx_trace.hip ] && extra="-mllvm -structurizecfg-skip-uniform-regions=1"
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
  -Wall -Wno-unused-function $extra "$@" -c csrc/$src -o _obj/variant_$name.o
objs=$(ls _obj/*.o | grep -v "_obj/variant_" | grep -v "_obj/srx_trace_" | grep -v "_obj/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj/variant_$name.o -ldl -o libsrx_$name.so
echo built libsrx_$name.so
