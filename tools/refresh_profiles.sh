#!/bin/bash
# Re-measure everything kept under profiles/ for round RR (default r02) on the GPU box; run through gpurun, e.g.
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02'
# then copy the files it lists from gpurun_out/refresh/ into profiles/ (gpurun_out/ is scratch).  Every step keeps its
# stderr in gpurun_out/refresh/<step>.err; a step that fails is reported and its output file removed (no partial evidence).
set -u
RR=${1:-r02}
PART=${2:-all}      # all | bench | sweeps | profile  (a whole refresh is longer than one gpurun call may last)
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
T="timeout -k 10"
FAILED=""
run() {   # run <output file> <seconds> <command ...>
    local out=$1 secs=$2; shift 2
    if ! $T $secs "$@" > "$O/$out" 2> "$O/$out.err"; then echo "FAILED: $out ($*)"; tail -3 "$O/$out.err"; rm -f "$O/$out"; FAILED="$FAILED $out"; fi
}
if [ "$PART" = all ] || [ "$PART" = bench ]; then
run ${RR}_bench_config1_wave_n10_s2.json 500 python bench.py
run ${RR}_bench_config2_wrench_f32_n20.json 400 python bench.py --config 2
run ${RR}_bench_config1_compact_n10_s2.json 200 python bench.py --kernel compact --no-cpu-baseline --no-latency
run ${RR}_latency_batch1.json 500 python tools/latency_probe.py 10000
run ${RR}_error_distribution_config1.json 300 python tools/error_distribution.py 1
run ${RR}_error_distribution_config2.json 400 python tools/error_distribution.py 2
fi
if [ "$PART" = all ] || [ "$PART" = sweeps ]; then
{ echo "# tools/schedule_bench.py on 1 x MI355X (4 rotating batches, 2 streams + longest-first hint, device-resident inputs)"
  for a in "double 10 4096" "mixed 10 4096" "double 10 16384" "mixed 8 4096" "single 8 4096" "single 12 16384" "mixed 12 16384" "single 16 16384" "mixed 16 16384" "double 16 16384" "single 20 16384" "mixed 20 16384" "double 20 16384" "mixed 24 16384" \
           "mixed 10 4096 auto 1" "mixed 12 16384 auto 1" "double 16 16384 auto 1" "double 20 65536 auto 1" "single 20 16384 wrench 1" "mixed 24 16384 wrench 1"; do
      $T 100 python tools/schedule_bench.py $a 2>> $O/schedules.err || echo "FAILED: schedule_bench $a"; done; } > $O/${RR}_other_schedules.txt
rm -f $O/batch_sweep.jsonl
for B in 1 8 64 512 4096 32768 65536 262144; do
    S=200; [ $B -ge 32768 ] && S=20
    $T 200 python bench.py --batch $B --steps $S --warmup 3 --no-cpu-baseline --no-latency >> $O/batch_sweep.jsonl 2>> $O/batch_sweep.err || echo "FAILED: batch sweep B=$B"
done
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/batch_sweep.jsonl") if l.startswith("{")]
with open("$O/${RR}_batch_sweep.txt", "w") as f:
    f.write("# bench.py --batch B (1 x MI355X, device-resident inputs, 4 rotating batches, 2 streams + longest-first hint, N=10 single support, fp64)\n")
    f.write("# batch/step   QP/s          ms/step    kernel\n")
    for d in rows:
        f.write("%8d   %12.0f   %8.4f   %s\n" % (d["config"]["batch_per_gpu"], d["value"], d["ms_per_step"], d["config"]["kernel"]))
PY
run ${RR}_cascade_kernels_hbm.json 100 python tools/cascade_bench.py
fi
if [ "$PART" = all ] || [ "$PART" = profile ]; then
# rocprofv3: kernel trace + stats, then the PMC passes (one counter group per run), both configs
# (round 4: configs[1] runs the deferred-tail kernel; configs[2] is profiled with --in-place -- with deferred restart passes on the tail stream the per-kernel durations
#  of rocprofv3 overlap and do not add up; configs[4] counts per CALL: 3 launches of a bucket's kernel per call, mean horizon 15)
bash tools/profile_config.sh $RR 1 wave_defer_f64_n10_s2 4096 10 8 1 > $O/profile_c1.log 2>&1 || { echo "FAILED: profile config 1"; FAILED="$FAILED profile1"; }
bash tools/profile_config.sh $RR 2 wrench_f32_n20 65536 20 4 1 --in-place > $O/profile_c2.log 2>&1 || { echo "FAILED: profile config 2"; FAILED="$FAILED profile2"; }
bash tools/profile_config.sh $RR 4 ragged_wrench_f64_n8_n12_n16_n24 16384 15 8 3 > $O/profile_c4.log 2>&1 || { echo "FAILED: profile config 4"; FAILED="$FAILED profile4"; }
bash tools/profile_config.sh $RR 1 compact_f64_n10_s2 4096 10 8 1 --kernel compact > $O/profile_c1_compact.log 2>&1 || { echo "FAILED: profile config 1 (compact)"; FAILED="$FAILED profile1c"; }
cp $R/gpurun_out/prof_${RR}_c1/${RR}_* $R/gpurun_out/prof_${RR}_c2_wrench_f32_n20/${RR}_* $R/gpurun_out/prof_${RR}_c4/${RR}_* $R/gpurun_out/prof_${RR}_c1_compact_f64_n10_s2/${RR}_* $O/ 2>/dev/null
fi
ls -la $O | grep ${RR}_
[ -z "$FAILED" ] || { echo "steps that failed:$FAILED"; exit 1; }
