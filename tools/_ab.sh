#!/bin/bash
cd /root/repo
run() { python bench.py "$@" --no-latency --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*:', round(d['value']/1e6,3), 'M QP/s; solved', round(d['config'].get('solved_frac'),5), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"; }
for rep in 1 2; do
run --config 3 --rho-restart -1
run --config 3
run --streams 2
run --streams 3
run --streams 4
run --streams 3 --rho-restart -1
run --streams 3 --steps 20 --warmup 5
run --streams 2 --steps 20 --warmup 5
done
