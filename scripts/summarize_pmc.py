#!/usr/bin/env python3
"""Mean of one PMC counter per kernel from a rocprofv3 counter_collection.csv:  summarize_pmc.py DIR COUNTER"""
import csv
import glob
import sys

d, counter = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "cusmc" in r["Kernel_Name"]:
            k = r["Kernel_Name"]
            k = k if len(k) < 100 else k[:97] + "..."
            vals.setdefault(k, []).append(float(r["Counter_Value"]))
print("| kernel | dispatches | mean %s |\n|---|---|---|" % counter)
for k, v in vals.items():
    print("| `%s` | %d | %.1f |" % (k, len(v), sum(v) / len(v)))
