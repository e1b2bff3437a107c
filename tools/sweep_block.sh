#!/bin/bash
for blk in 64 128 256; do for per in 12 24 48; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --streams 3 --lane-block $blk --lane-per $per 2>&1 | tail -1 > /tmp/sp.json
  python -c "import json; d=json.load(open('/tmp/sp.json')); print($blk, $per, '%.3e'%d['value'], d['ms_per_step'], d['roofline']['iterate_kernel_ms'])"
done; done
