#!/bin/bash
# same-box comparison of one-launch-kernel builds: tools/ab_fast_run.sh NAME ...   (libraries _ab/lib_NAME.so from
# tools/ab_fast_build.sh; "cur" = a copy of the in-tree build).  Prints the stream-only call (no point needs
# iterations), the headline call and three batches in flight, all cold.
for lib in "$@"; do
  echo "== $lib"
  LMPC_HIP_LIB=$PWD/_ab/lib_$lib.so timeout -k 10 200 python tools/stream_floor.py 2>&1 | grep -v amdgpu.ids | grep "0.50\|1.00" || exit 1
  LMPC_HIP_LIB=$PWD/_ab/lib_$lib.so timeout -k 10 200 python tools/hot_ab.py --reps 1 2>&1 | grep -v amdgpu.ids | grep cold || exit 1
done
