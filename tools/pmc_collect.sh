#!/bin/bash
# PMC passes over the benchmark's dominant kernel (one rocprofv3 run per counter group; --pmc only with --kernel-trace).
#   tools/pmc_collect.sh <outdir under gpurun_out>     then: python tools/pmc_parse.py <outdir>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $OUT/g$i.log; }
    echo "group $i done: $grp"
done
