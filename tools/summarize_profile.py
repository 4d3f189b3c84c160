#!/usr/bin/env python
"""gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) -> profiles/<tag>_*.{csv,json}."""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime)


def main(tag):
    base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out = os.environ.get("MGP_PROFILE_OUT") or os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = list(csv.DictReader(open(newest(base + "/trace/*/*_kernel_stats.csv"))))
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in stats[:32]:
            w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    trace = list(csv.DictReader(open(newest(base + "/trace/*/*_kernel_trace.csv"))))

    import re

    def hit(sub, name):
        return re.search(sub, name) is not None

    def durs(sub, lo=2500):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if hit(sub, r["Kernel_Name"])]
        return [x for x in d if x > lo]
    f = list(csv.DictReader(open(newest(base + "/pmc_fetch/*/*_counter_collection.csv"))))
    wv = list(csv.DictReader(open(newest(base + "/pmc_write/*/*_counter_collection.csv"))))

    def med(rows, name, sub):
        v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name and hit(sub, r["Kernel_Name"])]
        v = [x for x in v if x > 1.0]
        return (statistics.median(v), len(v)) if v else (0.0, 0)
    res = {}
    pats = {"spmv_kernel (no pre-scaling)": r"spmv_kernel<\d+, \d+, false|spmv_tile_kernel<false",
            "spmv_kernel (pre-scaled input)": r"spmv_kernel<\d+, \d+, true|spmv_tile_kernel<true",
            "cg_update_kernel": "cg_update_(c1_|q_)?kernel", "cg_init_kernel": "cg_init_kernel"}
    for label, k in pats.items():
        fs, n1 = med(f, "FETCH_SIZE", k)
        ws, _ = med(wv, "WRITE_SIZE", k)
        d = durs(k)
        res[label] = dict(launches=n1, FETCH_SIZE_KB_median=fs, WRITE_SIZE_KB_median=ws,
                      hbm_bytes_per_launch_corrected=int((2 * fs + ws) * 1024),
                      kernel_trace_median_ns=statistics.median(d) if d else None,
                      kernel_trace_launches=len(d))
    res["note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); "
                   "WRITE_SIZE as is; durations: non-skipped launches (> 2.5 us) from rocprofv3 --kernel-trace")
    res["spmv_hbm_bytes_per_launch"] = res["spmv_kernel (no pre-scaling)"]["hbm_bytes_per_launch_corrected"]
    # mean duration of the dominant kernel over every non-skipped launch of the profiled command (the solves' graph
    # launches outnumber everything else): the figure bench.py's roofline quotes as "in_graph_profile"
    d = durs(pats["spmv_kernel (no pre-scaling)"])
    res["spmv_kernel_trace_mean_ns"] = (sum(d) / len(d)) if d else None
    res["spmv_kernel_trace_launches"] = len(d)
    # which source tree was profiled: bench.py quotes this summary only while the tree's hash is the same
    sys.path.insert(0, ROOT)
    import bench
    res["source_hash"] = bench.source_hash()
    json.dump(res, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)
    for name in ("bench_trace.json", "bench_fetch.json"):
        src = os.path.join(base, name)
        if os.path.exists(src):
            open(os.path.join(out, tag + "_" + name), "w").write(open(src).read())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
