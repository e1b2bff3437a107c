#!/bin/bash
# sweep of the wave kernel's workgroup shape / LDS staging level for one workload
# usage: wave_sweep.sh <workload> <batch> [extra bench args]
W=$1; B=$2; shift 2
for LV in 0 1 2 3; do for NW in 4 8 16; do
  python bench.py --no-cpu-baseline --workload $W --batch $B --steps 5 --warmup 1 --wave-level $LV --wave-nwv $NW "$@" 2>&1 | tail -1 > /tmp/fn.json
  python -c "
import json; d=json.load(open('/tmp/fn.json')); print('level $LV nwv $NW  %.3e solves/s' % d['value'])" 2>/dev/null || echo "level $LV nwv $NW failed"
done; done
