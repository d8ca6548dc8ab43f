"""Average per-dispatch PMC counters of the solver kernels from tools/pmc_collect.sh output.
    python tools/pmc_parse.py gpurun_out/pmc [kernel-substring ...]
Prints {kernel-substring: {counter: mean per dispatch}}; default kernels: the compact / set-up kernel and the ADMM kernel."""
import csv, glob, json, os, sys
from collections import defaultdict


def parse(d, keys):
    out = {}
    for key in keys:
        acc, cnt = defaultdict(float), defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if key in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
        if acc:
            out[key] = {k: acc[k] / cnt[k] for k in sorted(acc)}
            out[key]["_dispatches"] = min(cnt.values())
    return out


if __name__ == "__main__":
    keys = sys.argv[2:] or ["srbdqp_setup1_kernel", "srbdqp_compact_kernel", "srbdqp_admm_kernel"]
    print(json.dumps(parse(sys.argv[1], keys), indent=1))
