#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly.  Usage: kstats.py DIR [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-58s calls=%5s total_ms=%9.3f avg_us=%10.1f %s%%" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
