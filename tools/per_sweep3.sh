#!/bin/bash
# work-list workgroups per shard of the iterating kernel (grid = per * 64 workgroups of 64 lanes),
# three batches in flight and one
for per in 8 16 24 30 32 40 48; do
  for st in 3 1; do
    echo -n "lane_per=$per streams=$st  "
    python bench.py --lane-per $per --streams $st --no-cpu-baseline --steps 600 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.4g  ms/step %.5f  frac %.3f  screen %.4f  iterate %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['screen_kernel_ms'], r['iterate_kernel_ms']))"
  done
done
