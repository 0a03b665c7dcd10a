#!/usr/bin/env python3
"""Warm-only per-kernel statistics from a rocprofv3 --kernel-trace CSV: for every kernel name the launches of the warm-up steps
(the first WARMUP/(WARMUP+STEPS) of its launches, in dispatch order) are dropped, so a cold first launch cannot move the average.
usage: warm_kernel_stats.py TRACE_DIR WARMUP STEPS OUT.csv"""
import collections, csv, glob, os, statistics, sys

root, warm, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
f = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
per = collections.OrderedDict()
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows, total = [], 0
for name, d in per.items():
    drop = len(d) * warm // (warm + steps) if len(d) >= warm + steps else 0
    d = d[drop:]
    rows.append((name, len(d), sum(d), sum(d) / len(d), min(d), max(d), statistics.pstdev(d) if len(d) > 1 else 0.0))
    total += sum(d)
rows.sort(key=lambda r: -r[2])
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "CallsPerStep", "UsPerStep"])
    for n, c, t, a, mn, mx, sd in rows:
        w.writerow([n, c, t, round(a, 1), round(100.0 * t / total, 3), mn, mx, round(sd, 1), round(c / steps, 3), round(t / steps / 1e3, 2)])
print(f"warm kernel time per step: {total / steps / 1e6:.4f} ms over {steps} steps ({f})")
for n, c, t, a, mn, mx, sd in rows[:12]:
    print(f"  {n[:90]:90s} calls/step {c / steps:6.2f}  avg {a / 1e3:9.2f} us  {100.0 * t / total:5.1f} %")
