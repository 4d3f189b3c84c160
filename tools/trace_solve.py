#!/usr/bin/env python
"""Print the kernel timeline of the last CG solve in a rocprofv3 --kernel-trace CSV (start offset, duration, gap)."""
import csv, glob, os, sys
base = sys.argv[1]
f = max(glob.glob(base + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "cg_init" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[max(0, idx - 3):idx + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print("%8.2f us  dur %6.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e
