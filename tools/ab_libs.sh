#!/bin/bash
# A/B of builds of the library in ONE gpurun call (boxes differ by up to 10 %): tools/ab_libs.sh <lib A> <lib B> ...   (paths relative to the repo root)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do
for L in "$@"; do
  export SRBDQP_LIB=$R/$L
  echo "== $L (rep $rep)"
  for c in 1 2; do python bench.py --config $c --no-also --no-latency --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[$c]', round(d['value']/1e6,3), 'M QP/s', round(d['ms_per_step'],4), 'ms')"; done
  python tools/ragged_bench.py 2>/dev/null | grep "ragged B"
  for a in "mixed 8 4096" "double 10 4096" "mixed 10 4096" "mixed 12 16384" "mixed 16 16384" "mixed 24 16384" "mixed 10 4096 auto 1" "double 16 16384 auto 1"; do python tools/schedule_bench.py $a 2>/dev/null; done
  python tools/latency_patterns.py 1500 2>/dev/null | grep "auto"
done; done
