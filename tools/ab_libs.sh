#!/bin/bash
# same-box A/B of two builds of the library (LMPC_HIP_LIB): usage tools/ab_libs.sh libA.so libB.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    for st in 3 1; do
      echo -n "$(basename $lib) streams=$st  "
      LMPC_HIP_LIB=$PWD/$lib python bench.py --streams $st --no-cpu-baseline --steps 800 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.4g  %.5f  screen %.4f iterate %.4f' % (d['value'], d['ms_per_step'], r['screen_kernel_ms'], r['iterate_kernel_ms']))"
    done
  done
done
