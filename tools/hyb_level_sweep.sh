cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --workload hybrid --batch 100000 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-launch --no-configs --f32"
for o in "" "--wave-level 1" "--wave-level 1 --wave-nwv 4" "--opt gram_scan=1" "--opt gram_scan=1 --wave-level 1" "--opt gram_scan=1 --wave-level 0"; do
  echo -n "== $o : "; timeout 100 $B $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'])"
done
