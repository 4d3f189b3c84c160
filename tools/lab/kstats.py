#!/usr/bin/env python
"""Compact view of a rocprofv3 *_kernel_stats.csv: kstats.py <dir or csv> [rows]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(p)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-72s calls %5s  avg %10.1f us  min %10.1f  total %9.3f ms  %5.1f %%" % (
        r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
