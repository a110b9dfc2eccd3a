#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of one bench workload -> per-step sum of kernel durations, idle gaps between consecutive
kernels inside bench.py's region markers, and the per-kernel table.  Usage: kernel_gaps.py DIR steps"""
import csv, glob, sys, collections
d, steps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "gpupoly_marker_kernel" in r["Kernel_Name"]]
lo, hi = marks[0], marks[1]
win = rows[lo + 1:hi]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(win, win[1:])]
print(f"{len(win)} dispatches in the timed region ({len(win) / steps:.1f} per step); span {span / steps / 1e3:.1f} us per step, "
      f"kernels {busy / steps / 1e3:.1f} us per step, idle between kernels {sum(g for g in gaps if g > 0) / steps / 1e3:.1f} us per step "
      f"(median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us)")
agg = collections.OrderedDict()
for r in win:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:60s} x{c / steps:5.1f}  {t / steps / 1e3:9.1f} us per step")
