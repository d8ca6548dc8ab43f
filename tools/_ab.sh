#!/bin/bash
cd /root/repo
run() { python bench.py --no-latency --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SRBDQP_LIB:', round(d['value']/1e6,3), 'M QP/s; solved', round(d['config'].get('solved_frac'),5), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"; }
for rep in 1 2 3; do
export SRBDQP_LIB=/root/repo/tools/lib_head.so; run
export SRBDQP_LIB=/root/repo/g1_locomotion_amd/libsrbdqp.so; run
done
unset SRBDQP_LIB
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
