#!/bin/bash
# same-box comparison of several library builds: usage tools/ab_many.sh lib1.so lib2.so ... (bench args via $EXTRA)
for rep in 1 2; do
  for lib in "$@"; do
    for st in 3 1; do
      echo -n "$(basename $lib) streams=$st  "
      LMPC_HIP_LIB=$PWD/$lib python bench.py --streams $st --no-cpu-baseline --steps 800 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.4g  %.5f  screen %.4f iterate %.4f' % (d['value'], d['ms_per_step'], r['screen_kernel_ms'], r['iterate_kernel_ms']))"
    done
  done
done
