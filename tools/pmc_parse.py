"""Average per-dispatch PMC counters of the dominant kernel from tools/pmc_collect.sh output.
    python tools/pmc_parse.py gpurun_out/pmc [kernel-substring]"""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "srbdqp_compact_kernel"
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if key in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    out = {k: acc[k] / cnt[k] for k in sorted(acc)}
    out["_dispatches_per_counter"] = {k: cnt[k] for k in sorted(cnt)}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
