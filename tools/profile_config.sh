#!/bin/bash
# rocprofv3 evidence for one bench configuration, on the GPU box (through gpurun):
#   bash tools/profile_config.sh <round tag, e.g. r02> <config 1|2> <bench kernel name> <B> <N> <esz> [launches per solve] [extra bench.py args ...]
# writes gpurun_out/prof_<tag>_c<config>/{<tag>_<kernel>_kernel_stats.csv, <tag>_<kernel>_pmc_summary.json, bench json}
set -eu
RR=$1; CFG=$2; KN=$3; B=$4; N=$5; ESZ=$6; LPS=${7:-1}
shift 6; [ $# -gt 0 ] && shift
EXTRA="$*"
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_${RR}_c${CFG}${EXTRA:+_${KN}}
mkdir -p "$O"
cd "$R"
STEPS=20; [ "$CFG" = "2" ] && STEPS=6; [ "$CFG" = "4" ] && STEPS=5
STREAMS=${STREAMS:-1}      # STREAMS=2: the counters and the trace in the as-benchmarked two-stream mode (consecutive launches overlap)
[ "$STREAMS" != "1" ] && { O=${O}_s${STREAMS}; mkdir -p "$O"; STEPS=40; export PMC_WORKLOAD="bench.py --streams $STREAMS ...: as benchmarked, consecutive solves overlap on $STREAMS streams; per-kernel durations overlap in time; see \`command\`"; }
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --config $CFG --streams $STREAMS --steps $STEPS --warmup 4 --no-cpu-baseline --no-latency --no-also $EXTRA > "$O/stats.log" 2>&1)
find "$O/stats" -name "*kernel_stats.csv" -exec cp {} "$O/${RR}_${KN}_kernel_stats.csv" \;
bash tools/pmc_collect.sh "$(basename "$O")/pmc" --config $CFG --streams $STREAMS --steps $( [ "$STREAMS" = 1 ] && echo 4 || echo 8 ) --warmup 4 $EXTRA > "$O/pmc_collect.log" 2>&1
python tools/pmc_summary.py "$O/pmc" "$O/stats" "$O/${RR}_${KN}_pmc_summary.json" "$KN" $B $N $ESZ ${RR#r} $LPS > "$O/pmc_summary_print.txt" 2> "$O/pmc_summary.err"
if [ "$STREAMS" != "1" ]; then    # the timeline of the overlapping launches (start, duration, queue of every kernel)
  TR=$(find "$O/stats" -name "*kernel_trace.csv" | head -1)
  { echo "# rocprofv3 --kernel-trace of: bench.py --config $CFG --streams $STREAMS --steps $STEPS --warmup 4 --no-cpu-baseline --no-latency --no-also $EXTRA"; echo "# start (us since the first row)  + duration   queue   kernel  -- the last 60 launches of the run"; python tools/trace_timeline.py "$TR" 60; } > "$O/${RR}_two_streams_timeline.txt"
  for f in "$O/${RR}_${KN}_kernel_stats.csv" "$O/${RR}_${KN}_pmc_summary.json"; do [ -f "$f" ] && mv "$f" "${f%.*}_s${STREAMS}.${f##*.}"; done
fi
ls -la "$O" | grep "${RR}_"
