#!/usr/bin/env python
"""Print the kernel timeline of the last CG solve in a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the
previous kernel, name.  The solve is located by its closing decision launch (the deciding update, or cg_decide in older traces)."""
import csv, glob, os, re, sys
base = sys.argv[1]
f = max(glob.glob(base + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# (round 4: a solve's first graph ends in the deciding update launch, cg_update_c1_kernel<true, ..>; older traces: cg_decide)
dec = [i for i, r in enumerate(rows) if "cg_update_c1_kernel<true" in r["Kernel_Name"] or "cg_decide" in r["Kernel_Name"]]
cands = dec[::-1] if dec else [max(i for i, r in enumerate(rows) if "cg_update" in r["Kernel_Name"])]
# walk back to the start of that solve: the gap before its first kernel is a host round trip (>= 8 us).  Under the profiler a
# replay now and then shows such a gap INSIDE a solve: among the last 200 solves take the last one of the most common length
import collections
spans = []
for end in cands[:200]:
    start = end
    while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 8000:
        start -= 1
    spans.append((start, end))
common = collections.Counter(e - s_ + 1 for s_, e in spans).most_common(1)[0][0]
start, end = next((s_, e) for s_, e in spans if e - s_ + 1 == common)
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
total = 0
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    print("%8.2f us  dur %6.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e
    total += e - s
print("solve: %d kernels, %.2f us of kernel time, %.2f us first start -> last end" % (end - start + 1, total / 1e3, (prev_end - t0) / 1e3))
