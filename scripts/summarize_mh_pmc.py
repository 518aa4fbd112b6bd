#!/usr/bin/env python3
"""Per-kernel, per-launch-shape means of every counter found under DIR (rocprofv3 counter_collection.csv files), with the
kernel durations of the same dispatches:  summarize_mh_pmc.py DIR [name-substring ...]   (default: metropolis hiword)"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
names = sys.argv[2:] or ["metropolis", "hiword"]
# template arguments are part of the identity of the matrix-core kernels: keep the name up to the argument list
short = lambda n: n.split("(")[0].replace("void cusmc::", "")
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not any(n in k for n in names):
            continue
        key = (k, r.get("Grid_Size", "?"))
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(n in k for n in names):
            dur[(k, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
counters = sorted({c for v in vals.values() for c in v})
print("| kernel | grid (threads) | us (mean over the PMC passes) | " + " | ".join(counters) + " |")
print("|---|---|---|" + "---|" * len(counters))
for key in sorted(vals):
    du = dur.get(key, [])
    print("| `%s` | %s | %s | " % (key[0], key[1], "%.1f" % (sum(du) / len(du)) if du else "?") +
          " | ".join("%.4g" % (sum(vals[key][c]) / len(vals[key][c])) if vals[key][c] else "-" for c in counters) + " |")
