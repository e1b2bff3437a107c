#!/bin/bash
for per in 4 8 12 16 32; do
  for st in 1 3; do
    python bench.py --steps 300 --warmup 10 --no-cpu-baseline --streams $st --lane-per $per 2>&1 | tail -1 > /tmp/sp.json
    python -c "import json; d=json.load(open('/tmp/sp.json')); print($per, $st, '%.3e'%d['value'], d['ms_per_step'], d['roofline']['screen_kernel_ms'], d['roofline']['iterate_kernel_ms'])"
  done
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload pendulum_hard | tail -1 | cut -c 60-160
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload pendulum_hard --lane-per 32 | tail -1 | cut -c 60-160
