#!/bin/bash
# rocprofv3 --kernel-trace of a configs[4] call (tools/ragged_bench.py): when the four bucket kernels (and their restart passes) run.
#   bash tools/trace_ragged.sh r03   -> gpurun_out/trace_ragged_r03/{..._kernel_trace.csv, ..._timeline.txt}
set -eu
RR=${1:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/trace_ragged_$RR
mkdir -p "$O"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/raw" -- python3 "$R/tools/ragged_bench.py" > "$O/bench.log" 2>&1)
find "$O/raw" -name "*kernel_trace.csv" -exec cp {} "$O/${RR}_ragged_kernel_trace.csv" \;
python3 - "$O/${RR}_ragged_kernel_trace.csv" > "$O/${RR}_ragged_timeline.txt" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wrench_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the ragged calls are the first 12 x 7 launches (2 warm-up + 10 timed calls); print the last call of them
calls = rows[:12 * 7]
last = calls[-7:]
t0 = min(int(r["Start_Timestamp"]) for r in last)
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    n = re.search(r"wrench_kernel<(\d+)", r["Kernel_Name"]).group(1)
    print("N=%-2s start %8.1f us  end %8.1f us  (+%7.1f)  grid %s  queue %s" % (n, s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?"), r.get("Queue_Id", "?")))
PY
cat "$O/${RR}_ragged_timeline.txt"
