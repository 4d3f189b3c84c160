#!/usr/bin/env python
"""Idle time of the device inside the last training epoch of a rocprofv3 --kernel-trace CSV (tools/profile_training.py run):
epochs are delimited by the optimizer's step kernel (multi_tensor_apply).  Prints kernel time, idle time, the idle time by
gap size, and the (previous kernel -> next kernel) pairs that carry most of it.  epoch_gaps.py <trace dir>"""
import collections, csv, glob, os, re, sys
base = sys.argv[1]
f = max(glob.glob(base + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:56]
marks = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
# one optimizer step may launch several multi_tensor kernels back to back: keep the last of each burst
steps = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 50]
if len(steps) < 2:
    sys.exit("fewer than two optimizer steps in the trace")
a, b = steps[-2] + 1, steps[-1] + 1
ep = rows[a:b]
t0, t1 = int(ep[0]["Start_Timestamp"]), int(ep[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ep)
gaps, pairs, pair_n = collections.Counter(), collections.Counter(), collections.Counter()
idle = 0
for p, q in zip(ep[:-1], ep[1:]):
    g = int(q["Start_Timestamp"]) - int(p["End_Timestamp"])
    if g <= 0:
        continue
    idle += g
    bucket = "<2us" if g < 2000 else "2-10us" if g < 10000 else "10-50us" if g < 50000 else "50-200us" if g < 200000 else ">200us"
    gaps[bucket] += g
    if g >= 10000:
        pairs[(nm(p), nm(q))] += g; pair_n[(nm(p), nm(q))] += 1
print("last epoch: %d kernels, wall %.2f ms, kernel time %.2f ms, idle %.2f ms" % (len(ep), (t1 - t0) / 1e6, busy / 1e6, idle / 1e6))
for k in ("<2us", "2-10us", "10-50us", "50-200us", ">200us"):
    print("  idle in gaps %-9s %7.2f ms" % (k, gaps[k] / 1e6))
print("gaps >= 10 us by (previous kernel -> next kernel):")
for k, v in pairs.most_common(24):
    print("  %7.2f ms in %4d gaps (%6.1f us each)  %s -> %s" % (v / 1e6, pair_n[k], v / 1e3 / pair_n[k], k[0], k[1]))
