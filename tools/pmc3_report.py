#!/usr/bin/env python3
"""Per-kernel-family totals of the stall-reason counters collected by tools/gpu_pmc3.sh (gpurun_out/pmc_s1..s3).
Each counter is printed raw (summed over the launches of the family) and relative to SQ_WAVE_CYCLES of its own pass."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
pat = sys.argv[2] if len(sys.argv) > 2 else "wino"
for tag in ("s1", "s2", "s3"):
    f = glob.glob(f"{root}/pmc_{tag}/runc/*counter_collection.csv")
    if not f: continue
    rows = list(csv.DictReader(open(max(f, key=os.path.getmtime))))
    agg = collections.OrderedDict()
    for r in rows:
        n = r["Kernel_Name"]
        if pat not in n: continue
        key = n.split("(")[0].replace("mgu::", "").replace("void ", "")[:40]
        agg.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, a in agg.items():
        wc = max(a.get("SQ_WAVE_CYCLES", 1), 1)
        print(tag, k)
        for c, v in a.items():
            print(f"    {c:32s} {v:16.0f}  {v / wc * 100:8.2f} % of wave-cycles")
