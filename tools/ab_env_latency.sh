#!/bin/bash
# the bench's batch-1 latency legs (10,000 calls each) under different environment settings inside ONE GPU-box call:
#   tools/ab_env_latency.sh "" "SRBDQP_DONE_FENCE=1" "SRBDQP_NO_AQL=1" ...   (two rounds)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for E in "$@"; do
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['config']['latency_batch1_us']
print('[%s] rep $rep ' % '$E' + '  '.join('%s %.2f/%.2f' % (k, d[k]['p50'], d[k]['p99']) for k in ('c_abi', 'c_abi_double_support', 'mpc_update_double_support', 'closed_loop_cold')))"
done; done
