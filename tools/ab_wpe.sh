#!/bin/bash
# same-box A/B: library builds with 4 (default) / 5 / 6 wavefronts per SIMD for the 1-2 slot binary64 wavefront kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in "mass_spring 1000000" "mass_spring_3in 1000000" "mass_spring_3in_feasible 1000000"; do
  set -- $w
  for lib in "" linearmpc.jl_amd/lib/ab/lib_wpe5.so linearmpc.jl_amd/lib/ab/lib_wpe6.so; do
    echo -n "$1 ${lib:-default} : "
    LMPC_HIP_LIB=${lib:+$PWD/$lib} timeout -k 5 120 python bench.py --workload $1 --batch $2 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-single-launch --no-configs 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'])" || echo failed
  done
done
