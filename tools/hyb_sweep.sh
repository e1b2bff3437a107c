#!/bin/bash
# hybrid B&B: layout / form sweep (one batch in flight).  usage: tools/hyb_sweep.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --workload hybrid --batch 100000 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-launch --no-configs"
for o in "" "--opt wave_packed=0" "--opt wave_packed=1" "--opt gram_scan=1" "--opt gram_scan=1 --opt wave_packed=0" "--f32" "--f32 --opt wave_packed=1"; do
  echo "== $o"; $B $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'])"
done
