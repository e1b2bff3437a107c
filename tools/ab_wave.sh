#!/bin/bash
# same-box comparison of library builds on the wavefront-kernel workloads: usage tools/ab_wave.sh lib1.so lib2.so ...
run() { LMPC_HIP_LIB=$PWD/$LIB python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g' % d['value'], end='  ')"; }
for rep in 1 2; do
  for LIB in "$@"; do
    echo -n "$(basename $LIB): "
    run --workload mass_spring_3in --steps 5 --warmup 1 --batch 100000
    run --workload soft_doc --steps 10 --warmup 2 --batch 200000
    run --workload mass_spring --steps 10 --warmup 2 --batch 200000
    run --workload mass_spring_3in --steps 5 --warmup 1 --batch 100000 --f32
    run --workload hybrid --steps 3 --warmup 1 --batch 20000
    run --workload hybrid --steps 3 --warmup 1 --batch 20000 --f32
    echo
  done
done
