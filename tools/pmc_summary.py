#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (mean per dispatch).  Usage: pmc_summary.py DIR [substr]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {sum(x)/len(x):16.1f}  (n={len(x)})")
