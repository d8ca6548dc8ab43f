#!/bin/bash
# the bench's batch-1 latency legs (10,000 calls each) on several builds of the library inside ONE GPU-box call: tools/ab_lib_latency.sh <lib> ... ("-" = in-tree; two rounds)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$R/$L; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['config']['latency_batch1_us']
print('[%s] rep $rep ' % '$L' + '  '.join('%s %.2f/%.2f' % (k, d[k]['p50'], d[k]['p99']) for k in ('c_abi', 'c_abi_double_support', 'mpc_update_double_support', 'closed_loop_cold')))"
done; done
