#!/bin/bash
# wavefront-kernel workloads, one batch in flight, both forms: solves/s (quick A/B after a kernel change)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo -n "$1 $3: "; timeout -k 5 120 python bench.py --workload $1 --batch $2 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-single-launch --no-configs $3 2>/dev/null > /tmp/wq.json; python -c "import json; d=json.loads(open('/tmp/wq.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'])" || echo failed; }
for w in "mass_spring 1000000" "mass_spring_3in 1000000" "mass_spring_3in_feasible 1000000" "pendulum_N50 200000" "pendulum_N75 200000" "pendulum_N100 200000" "pendulum_N125 200000"; do
  set -- $w; run $1 $2 ""; run $1 $2 "--opt gram_scan=1"
done
