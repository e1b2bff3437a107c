#!/bin/bash
# A/B build of the one-launch kernel only: tools/ab_fast_build.sh NAME [-DFLAG ...]  ->  _ab/lib_NAME.so
# (pendulum instantiation only: seconds instead of minutes; the other objects come from the in-tree build)
set -e
cd "$(dirname "$0")/../linearmpc.jl_amd/csrc"
name=$1; shift
mkdir -p ../../_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mfma -Wno-unused-value -DLMPC_FAST_ONLY_PENDULUM "$@" \
    -c -o ../../_ab/fast_$name.o lmpc_fast_inst.hip
objs=$(ls ../lib/obj/*.o | grep -v lmpc_fast.o)
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../_ab/lib_$name.so $objs ../../_ab/fast_$name.o -ldl
echo "_ab/lib_$name.so"
