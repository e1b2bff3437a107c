#!/bin/bash
# lane-kernel workgroup size (bench.py --lane-block), three batches in flight and one
for rep in 1 2; do
for lb in 0 64 128 256; do
  for st in 3 1; do
    echo -n "lane_block=$lb streams=$st  "
    python bench.py --lane-block $lb --streams $st --no-cpu-baseline --steps 800 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.4g  %.5f  screen %.4f iterate %.4f' % (d['value'], d['ms_per_step'], r['screen_kernel_ms'], r['iterate_kernel_ms']))"
  done
done
done
