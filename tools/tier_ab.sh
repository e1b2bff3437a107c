#!/bin/bash
# A/B of the lane kernel's first-tier capacity (lmpc_set_option "lane_tier"): headline batch and the
# +-20 "hard" batch, three batches in flight and one.
set -e
mkdir -p gpurun_out
for wl in pendulum pendulum_hard; do
  for tier in 1 0; do
    for st in 3 1; do
      echo "== $wl tier=$tier streams=$st"
      python bench.py --workload $wl --lane-tier $tier --streams $st --no-cpu-baseline --steps 500 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.4g  ms/step %.5f  frac %.3f  screen %.4f  iterate %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['screen_kernel_ms'], r['iterate_kernel_ms']))"
    done
  done
done
