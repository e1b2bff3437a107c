#!/bin/bash
# timing-only ablation of the screening kernel (diagnostic)
for ab in ${@:-0 1 2 3 4 7 8 15}; do
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --streams 1 --ablate $ab 2>&1 | tail -1 > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print($ab, d['roofline']['screen_kernel_ms'], d['roofline']['iterate_kernel_ms'], '%.3e'%d['value'])"
done
