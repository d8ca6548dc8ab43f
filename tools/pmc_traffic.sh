#!/bin/bash
# HBM-side traffic of the benchmark's kernels only (FETCH_SIZE, WRITE_SIZE; one rocprofv3 run each):
#   tools/pmc_traffic.sh <outdir under gpurun_out> [bench args]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-pmc_tr}; shift || true
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-latency > "$OUT/g$i.log" 2>&1 || echo "FAILED group $i"
    i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "srbdqp" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    f = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1) * 2048
    w = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [1])), 1) * 1024
    print("%-72s fetch(x2) %.1f MB  write %.1f MB per launch (n=%d)" % (k, f / 1e6, w / 1e6, len(d.get("FETCH_SIZE", []))))
PY
