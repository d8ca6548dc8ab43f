#!/bin/bash
# A/B of library builds on the batch-1 calls in ONE gpurun call: tools/ab_latency.sh <calls> <lib A> <lib B> ...  ("-" = the in-tree library)
# per library: tools/latency_patterns.py (staged C-ABI call per contact pattern and kernel) and the per-phase stamps of the low-latency general kernel
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
CALLS=$1; shift
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$R/$L; fi
  echo "== $L (rep $rep)"
  python tools/latency_patterns.py $CALLS 2>/dev/null | grep -E "^(single|double|mixed) (auto|wrench)"
  if [ $rep = 1 ]; then SCHED=double python tools/wrench_stamps_staged.py 2>/dev/null; SCHED=mixed python tools/wrench_stamps_staged.py 2>/dev/null | head -12; fi
done; done
