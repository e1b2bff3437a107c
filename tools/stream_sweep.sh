#!/bin/bash
# batches in flight (bench.py --streams) vs throughput on the headline batch
for st in 1 2 3 4 5 6 8; do
  echo -n "streams=$st  "
  python bench.py --streams $st --no-cpu-baseline --no-single-launch --steps 600 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.4g  ms/step %.5f  frac %.3f  enqueue %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['host_enqueue_ms_per_step']))"
done
