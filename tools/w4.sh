run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > /tmp/fn.json; python -c "
import json,sys; d=json.load(open('/tmp/fn.json')); c=d['config']
print(' '.join(sys.argv[1:]), '| %.3e solves/s' % (d['value']))" "$@"; }
run --workload mass_spring_3in --steps 5 --warmup 1 --batch 100000
run --workload soft_doc --steps 10 --warmup 2 --batch 200000
run --workload mass_spring --wave --steps 10 --warmup 2 --batch 200000
run --workload pendulum_hard --wave --steps 10 --warmup 2
