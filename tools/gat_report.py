#!/usr/bin/env python3
import csv, glob, os, sys, collections
def newest(pat):
    f = glob.glob(pat)
    return [max(f, key=os.path.getmtime)] if f else []
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for kind in ("patch", "stress"):
    f = newest(f"{root}/gat_{kind}_trace/runc/*kernel_stats.csv")
    if not f: continue
    print(f"== {kind}")
    dur = {}
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        if "mgu::" in n: dur[n.split("(")[0].replace("void ", "").replace("mgu::", "")[:40]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    by = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for tag, idx in (("fetch", 0), ("write", 1)):
        g = newest(f"{root}/gat_{kind}_{tag}/runc/*counter_collection.csv")
        if not g: continue
        for r in csv.DictReader(open(g[0])):
            n = r["Kernel_Name"]
            if "mgu::" not in n: continue
            k = n.split("(")[0].replace("void ", "").replace("mgu::", "")[:40]
            by[k][idx] += float(r["Counter_Value"]) * 1024 * (2 if tag == "fetch" else 1)
            if tag == "fetch": by[k][2] += 1
    for k, (us, calls) in dur.items():
        rd, wr, n = by[k]
        n = max(n, 1)
        print(f"  {k:40s} {us:8.1f} us  read {rd/n/1e6:8.2f} MB  write {wr/n/1e6:8.2f} MB  -> {(rd+wr)/n/us/1e6:6.2f} TB/s measured traffic")
