#!/bin/bash
# config 3 (3-input masses, 1e6 points): resident wavefronts per CU x factor layout x staging level (VERDICT r3 #5a)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="timeout -k 5 90 python bench.py --workload mass_spring_3in --batch 1000000 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-single-launch --no-configs"
run() { echo -n "$1 : "; $B $2 2>/dev/null > /tmp/cfg3.json; python -c "import sys,json; d=json.loads(open('/tmp/cfg3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'])" || echo failed; }
run "default" ""
run "default gram" "--opt gram_scan=1"
for w in 16 20 24 28 32; do run "packed level0 waves=$w" "--opt wave_packed=1 --wave-level 0 --wave-cap $w"; done
for w in 16 24; do run "packed level1 nwv8 waves=$w" "--opt wave_packed=1 --wave-level 1 --wave-nwv 8 --wave-cap $w"; done
for w in 12 16 19; do run "square level0 nwv1 waves=$w" "--opt wave_packed=0 --wave-level 0 --wave-nwv 1 --wave-cap $w"; done
for w in 24 32; do run "gram packed level0 waves=$w" "--opt gram_scan=1 --opt wave_packed=1 --wave-level 0 --wave-cap $w"; done
