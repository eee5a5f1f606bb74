"""Print one generation's kernel timeline from a rocprofv3 --kernel-trace CSV:
start offset, duration and the idle gap before each kernel.  Development aid.
Usage: trace_timeline.py kernel_trace.csv [anchor-kernel-substring] [which-occurrence]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_assemble"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.2f us  dur %8.2f  gap %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e)
    busy += e - s
print("generation: %.2f us wall, %.2f us in kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3))
