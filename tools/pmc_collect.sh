#!/bin/bash
# PMC passes over the benchmark's dominant kernel (one rocprofv3 run per counter group; --pmc only with --kernel-trace).
#   tools/pmc_collect.sh <outdir under gpurun_out> [bench.py arguments ...]     then: python tools/pmc_summary.py ...
# A failed pass stops the collection (no partial sets in profiles/); every pass keeps its stderr in <outdir>/gN.log.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-pmc}
shift || true
BENCH_ARGS=("$@")
[ ${#BENCH_ARGS[@]} -eq 0 ] && BENCH_ARGS=(--streams 1 --steps 5 --warmup 2)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    i=$((i+1))
    if ! timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" "${BENCH_ARGS[@]}" --no-cpu-baseline --no-latency --no-also > "$OUT/g$i.log" 2>&1; then
        echo "group $i FAILED ($grp)"; tail -5 "$OUT/g$i.log"; exit 1
    fi
    echo "group $i done: $grp"
done
