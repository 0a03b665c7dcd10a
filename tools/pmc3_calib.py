#!/usr/bin/env python3
"""Counters of tools/gpu_pmc3.sh for one kernel family, as a fraction of the SIMD-cycles of its launches
(sum of launch durations x clock x 256 CUs x 4 SIMDs).  usage: pmc3_calib.py [root] [name substring] [GHz]"""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
pat = sys.argv[2] if len(sys.argv) > 2 else "wino3x3"
ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.35
for tag in ("s1", "s2", "s3"):
    f = glob.glob(f"{root}/pmc_{tag}/runc/*counter_collection.csv")
    if not f: continue
    rows = list(csv.DictReader(open(max(f, key=os.path.getmtime))))
    seen = set(); t = 0; cnt = collections.OrderedDict()
    for r in rows:
        if pat not in r["Kernel_Name"]: continue
        cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); t += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    simd = t * 1e-9 * ghz * 1e9 * 256 * 4
    print(f"{tag}: {len(seen)} launches, {t / 1e3:.0f} us")
    for k, v in cnt.items(): print(f"    {k:32s} {v:16.0f}  {v / simd:8.3f} per SIMD-cycle")
