#!/bin/bash
# instruction-fetch counters of the benchmark's kernels (one rocprofv3 run per group):  tools/pmc_icache.sh <outdir> [bench args]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-pmc_ic}; shift || true
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-latency > "$OUT/g$i.log" 2>&1 || echo "FAILED group $i"
    i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s mean/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
