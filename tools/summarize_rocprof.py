"""Reduce rocprofv3 CSV output (run on the GPU box) to small per-kernel summaries that can be committed
under profiles/.  Usage: summarize_rocprof.py <dir> <out.csv>   (searches *_kernel_stats.csv / *_counter_collection.csv)"""
import csv, glob, os, sys, collections

d, out = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(("stats", r))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out, "w") as fo:
    w = csv.writer(fo)
    if rows:
        keys = list(rows[0][1].keys())
        w.writerow(keys)
        for _, r in rows:
            w.writerow([r[k] for k in keys])
    if agg:
        w.writerow(["Kernel_Name", "Counter_Name", "launches", "mean_per_launch", "sum"])
        for k, cs in sorted(agg.items()):
            for c, v in sorted(cs.items()):
                w.writerow([k, c, len(v), sum(v) / len(v), sum(v)])
print(open(out).read()[:6000])
