#!/bin/bash
# A/B of library builds on configs[1] in ONE gpurun call (boxes differ by several per cent): tools/ab_config1.sh <lib A> <lib B> ...  (paths relative to the repo root;
# "-" = the in-tree library).  Prints value / value_plain / in_place.value for the driver's flags (20 steps) and for 500 steps, two rounds.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$R/$L; fi
  for K in 20 500; do
    python bench.py --steps $K --warmup 10 --no-latency --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$L rep $rep K=$K', 'value %.2f M' % (d['value']/1e6), 'plain %.2f' % (d['value_plain']/1e6), 'stale %.2f' % (d['value_stale_hint']/1e6), 'in_place %.2f / %.2f' % (d['in_place']['value']/1e6, d['in_place']['value_plain']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'], d['config']['kernel'], 'solved %.5f' % d['config']['solved_frac'])"
  done
done; done
