#!/bin/bash
# Re-measure everything kept under profiles/ for round RR (default r01) on the GPU box; run through gpurun, e.g.
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02'
# then copy the files it lists from gpurun_out/refresh/ into profiles/ (gpurun_out/ is scratch).
set -u
RR=${1:-r01}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
T="timeout -k 10"
$T 400 python bench.py > $O/${RR}_bench_wave_n10_s2.json 2> $O/bench.err
$T 200 python bench.py --kernel compact --no-cpu-baseline --no-latency > $O/${RR}_bench_fused_compact_n10_s2.json 2>/dev/null
$T 500 python tools/latency_probe.py 10000 > $O/${RR}_latency_batch1.json 2> $O/latency.err
$T 100 python tools/floor_probe.py > $O/${RR}_floor_probe.txt 2>/dev/null
$T 120 python tools/phase_stamps.py > $O/${RR}_phase_stamps.txt 2>/dev/null
$T 100 python tools/phase_stamps_staged.py > $O/${RR}_phase_stamps_staged.txt 2>/dev/null
[ -x tools/pcie_probe ] && PCIE_PROBE_TOUCH=1 $T 60 tools/pcie_probe > $O/${RR}_pcie_probe.txt 2>&1
(echo "# tools/schedule_bench.py on 1 x MI355X (2 streams + longest-first hint, device-resident inputs)"
 for a in "double 10" "mixed 10" "single 8" "single 16" "single 20" "double 4" "single 4" "double 8"; do $T 100 python tools/schedule_bench.py $a 4096 2>/dev/null; done) > $O/${RR}_other_schedules.txt
rm -f $O/batch_sweep.jsonl
for B in 1 8 64 512 4096 32768 65536 262144; do
    S=200; [ $B -ge 32768 ] && S=20
    $T 200 python bench.py --batch $B --steps $S --warmup 3 --no-cpu-baseline --no-latency >> $O/batch_sweep.jsonl 2>/dev/null
done
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/batch_sweep.jsonl")]
with open("$O/${RR}_batch_sweep.txt", "w") as f:
    f.write("# bench.py --batch B (1 x MI355X, device-resident inputs, 2 streams + longest-first hint, N=10 single support, fp64)\n")
    f.write("# batch/step   QP/s          ms/step    kernel\n")
    for d in rows:
        f.write("%8d   %12.0f   %8.4f   %s\n" % (d["config"]["batch_per_gpu"], d["value"], d["ms_per_step"], d["config"]["kernel"]))
PY
$T 100 python tools/cascade_bench.py > $O/${RR}_cascade_kernels_hbm.json 2>/dev/null
# rocprofv3: kernel trace + stats, then the PMC passes (one counter group per run)
(cd /tmp && export TMPDIR=/tmp && $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency > $O/stats.log 2>&1)
bash tools/pmc_collect.sh refresh/pmc > $O/pmc_collect.log 2>&1
python tools/pmc_summary.py $O/pmc $O/stats $O/${RR}_wave_n10_s2_pmc_summary.json > /dev/null 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/${RR}_wave_n10_s2_kernel_stats.csv \;
ls -la $O | grep ${RR}_
