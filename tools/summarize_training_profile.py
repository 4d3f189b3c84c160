#!/usr/bin/env python
"""Two `rocprofv3 --kernel-trace --stats` runs of tools/profile_training.py (E1 and E2 epochs) -> profiles/<tag>_training_<mode>.json:
kernel time and launches per epoch = the difference of the two runs / (E2 - E1), the top kernels of that difference.
Usage: summarize_training_profile.py <tag> <mode> <dir E1> <E1> <dir E2> <E2>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats(d):
    f = max(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}


def main(tag, mode, d1, e1, d2, e2):
    import bench
    a, b = stats(d1), stats(d2)
    e1, e2 = int(e1), int(e2)
    rows = []
    for name in b:
        c2, t2 = b[name]
        c1, t1 = a.get(name, (0, 0.0))
        if c2 - c1 > 0:
            rows.append((name, (c2 - c1) / (e2 - e1), (t2 - t1) / (e2 - e1)))
    rows.sort(key=lambda r: -r[2])
    out = dict(mode=mode, epochs_difference=e2 - e1, kernel_ms_per_epoch=round(sum(r[2] for r in rows) / 1e6, 3),
               launches_per_epoch=round(sum(r[1] for r in rows), 1), source_hash=bench.source_hash(),
               top_kernels=[dict(name=r[0].replace("(anonymous namespace)::", "")[:100], launches_per_epoch=round(r[1], 1),
                                 ms_per_epoch=round(r[2] / 1e6, 3)) for r in rows[:14]],
               recipe="rocprofv3 --kernel-trace --stats -- python3 tools/profile_training.py %s {%d,%d}; per-epoch figures = difference "
                      "of the two runs / %d (setup cancels)" % (mode, e1, e2, e2 - e1))
    dst = os.environ.get("MGP_PROFILE_OUT") or os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    name = os.path.join(dst, "%s_training_%s.json" % (tag, "supervised" if mode == "sup" else "semisupervised"))
    json.dump(out, open(name, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main(*sys.argv[1:7])
