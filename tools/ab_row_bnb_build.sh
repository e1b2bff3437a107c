#!/bin/bash
# A/B build of the row kernel's branch-and-bound unit (binary32): tools/ab_row_bnb_build.sh NAME [-DFLAG ...]
# -> linearmpc.jl_amd/lib/ab/lib_NAME.so (use with LMPC_HIP_LIB=...; run from the repository root)
set -e
cd "$(dirname "$0")/../linearmpc.jl_amd/csrc"
name=$1; shift
mkdir -p ../lib/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mfma -Wno-unused-value -DLMPC_ROW_REAL=float -DLMPC_ROW_BNB=1 "$@" \
    -c -o ../lib/ab/rowb_$name.o lmpc_row_inst.hip
objs=$(ls ../lib/obj/*.o | grep -v "row_f32_bnb.o")
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../lib/ab/lib_$name.so $objs ../lib/ab/rowb_$name.o -ldl
rm -f ../lib/ab/rowb_$name.o
echo "linearmpc.jl_amd/lib/ab/lib_$name.so"
