#!/bin/bash
# rocprofv3 --kernel-trace of the AS-BENCHMARKED mode of configs[1] (2 streams + longest-first hint): the timeline shows the launches
# of the two streams overlapping, which is how a step takes less wall time than one launch lasts.  Run on the GPU box:
#   bash tools/trace_two_streams.sh r03        -> gpurun_out/trace_r03/{r03_two_streams_kernel_trace.csv, r03_two_streams_timeline.txt}
set -eu
RR=${1:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/trace_$RR
mkdir -p "$O"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/raw" -- python3 "$R/bench.py" --steps 40 --warmup 8 --no-cpu-baseline --no-latency --no-also > "$O/bench.log" 2>&1)
find "$O/raw" -name "*kernel_trace.csv" -exec cp {} "$O/${RR}_two_streams_kernel_trace.csv" \;
find "$O/raw" -name "*kernel_stats.csv" -exec cp {} "$O/${RR}_two_streams_kernel_stats.csv" \;
python3 "$R/tools/trace_timeline.py" "$O/${RR}_two_streams_kernel_trace.csv" 60 > "$O/${RR}_two_streams_timeline.txt"
python3 - "$O/${RR}_two_streams_kernel_trace.csv" >> "$O/${RR}_two_streams_timeline.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "setup1_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
s = [int(r["Start_Timestamp"]) for r in rows]; e = [int(r["End_Timestamp"]) for r in rows]
dur = [(b - a) / 1e3 for a, b in zip(s, e)]
span = (max(e) - min(s)) / 1e3
ov = sum(max(0, min(e[i], e[i + 1]) - s[i + 1]) for i in range(len(rows) - 1)) / 1e3
print("\n# last %d solver launches: mean duration %.1f us, wall span %.1f us = %.1f us per launch; consecutive launches overlap %.1f us on average"
      % (len(rows), sum(dur) / len(dur), span, span / len(rows), ov / (len(rows) - 1)))
PY
tail -3 "$O/${RR}_two_streams_timeline.txt"
