#!/bin/bash
# one line per workload/mode for the DESIGN.md table
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > /tmp/fn.json; python -c "
import json,sys; d=json.load(open('/tmp/fn.json')); r=d['roofline']
print(' '.join(sys.argv[1:]), '| %.3e solves/s | %.4f ms/step | screen %.1f us iterate %.1f us | %.0f GB/s | frac %.3f' % (d['value'], d['ms_per_step'], 1e3*r['screen_kernel_ms'], 1e3*r['iterate_kernel_ms'], r['achieved'], r['frac']))" "$@"; }
run --streams 3
run --streams 1
run --streams 3 --workload pendulum_hard --steps 200
run --streams 1 --workload pendulum_hard --steps 200
run --streams 3 --workload mass_spring --steps 30 --warmup 3
run --streams 3 --workload soft_doc --steps 10 --warmup 2 --batch 200000
run --streams 3 --workload mass_spring --wave --steps 10 --warmup 2 --batch 200000
