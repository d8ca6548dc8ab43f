"""Timeline of a rocprofv3 --kernel-trace csv: per kernel launch start / duration / queue, for the last N rows.
    python tools/trace_timeline.py <kernel_trace.csv> [rows=60]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f us  +%8.1f us  q%-3s grid %-8s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), r["Kernel_Name"][:70]))
