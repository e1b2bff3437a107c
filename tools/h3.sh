run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > /tmp/fn.json; python -c "
import json,sys; d=json.load(open('/tmp/fn.json')); r=d['roofline']
print(' '.join(sys.argv[1:]), '| %.3e solves/s | screen %.1f us iterate %.1f us' % (d['value'], 1e3*r['screen_kernel_ms'], 1e3*r['iterate_kernel_ms']))" "$@"; }
run --streams 3
run --streams 1
run --streams 3 --workload pendulum_hard --steps 200
run --streams 1 --workload pendulum_hard --steps 200
