#!/bin/bash
# configs[1] (500 steps) and the single-support batch-1 call on several builds inside ONE GPU-box call: tools/ab_single.sh <lib> ... ("-" = in-tree; two rounds)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$R/$L; fi
  python bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); l = d['config']['latency_batch1_us']
print('[%s] rep $rep value %.2f M plain %.2f in_place %.2f  ' % ('$L', d['value']/1e6, d['value_plain']/1e6, d['in_place']['value']/1e6) + '  '.join('%s %.2f/%.2f' % (k, l[k]['p50'], l[k]['p99']) for k in ('c_abi', 'cold', 'c_abi_double_support')))"
done; done
