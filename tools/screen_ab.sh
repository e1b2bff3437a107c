#!/bin/bash
# A/B of the screening kernel's memory access variants (bench.py --ablate: 8 = plain instead of
# nontemporal loads/stores, 16 = outputs only for finished problems (lines with holes))
for ab in 0 8 16 24; do
  for st in 3 1; do
    echo -n "ablate=$ab streams=$st  "
    python bench.py --ablate $ab --streams $st --no-cpu-baseline --steps 600 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.4g  ms/step %.5f  frac %.3f  screen %.4f  iterate %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['screen_kernel_ms'], r['iterate_kernel_ms']))"
  done
done
