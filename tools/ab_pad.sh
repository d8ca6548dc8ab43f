#!/bin/bash
# the batch-1 double-support call on several builds inside ONE GPU-box call: tools/ab_pad.sh <lib> ...  (staged stamps + p50 of the C-ABI call, two rounds)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$R/$L; fi
  echo "== $L (rep $rep)"
  python tools/latency_patterns.py 3000 2>/dev/null | grep -E "^(double|mixed) auto"
  SCHED=double python tools/wrench_stamps_staged.py 2>/dev/null | grep -E "ADMM|total"
done; done
