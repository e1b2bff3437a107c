cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --workload mass_spring --batch 1000000 --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --no-single-launch --no-configs"
for o in "" "--opt wave=0" "--opt wave=0 --no-screen" "--opt gram_scan=1" "--opt screen_wave=0"; do
  echo -n "== $o : "; timeout 100 $B $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'], d['config']['kernel'])"
done
