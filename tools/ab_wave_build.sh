#!/bin/bash
# A/B build of the binary64 wavefront kernel (no branch and bound): tools/ab_wave_build.sh NAME [-DFLAG ...] -> _ab/lib_NAME.so
set -e
cd "$(dirname "$0")/../linearmpc.jl_amd/csrc"
name=$1; shift
mkdir -p ../../_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mfma -Wno-unused-value -DLMPC_WV_REAL=double -DLMPC_WV_BNB=0 "$@" \
    -c -o ../../_ab/wave_$name.o lmpc_wave_inst.hip
objs=$(ls ../lib/obj/*.o | grep -v "wave_f64.o")
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../_ab/lib_$name.so $objs ../../_ab/wave_$name.o -ldl
echo "_ab/lib_$name.so"
