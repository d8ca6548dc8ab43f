#!/bin/bash
# configs[2] and configs[4] on several builds of the library inside ONE GPU-box call: tools/ab_config24.sh <lib> ... ("-" = in-tree; two rounds)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$R/$L; fi
  for C in 2 4; do
    python bench.py --config $C --no-cpu-baseline --no-latency --no-also 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[%s] rep $rep config $C: %.3f M QP/s  %.3f ms/step  solved %.4f  kernel_ms %.3f' % ('$L', d['value'] / 1e6, d['ms_per_step'], d['config']['solved_frac'], d['roofline']['kernel_ms']))"
  done
done; done
