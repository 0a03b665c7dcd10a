#!/usr/bin/env python3
"""Per-kernel HBM bytes of the stage benchmarks from tools/gpu_pmc_stage.sh (FETCH_SIZE doubled: gfx950 tallies 128-byte
requests at 64 B, MI355X_MICROARCH.md HBM section; WRITE_SIZE exact; both in KiB)."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for tag in ("ncut", "region", "det"):
    acc = collections.OrderedDict()
    for ctr, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        f = glob.glob(f"{root}/pmcs_{ctr}_{tag}/runc/*counter_collection.csv")
        if not f: continue
        for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
            n = r["Kernel_Name"]
            if "mgu::" not in n: continue
            k = n.split("(")[0].replace("void ", "").replace("mgu::", "")
            d = acc.setdefault(k, collections.defaultdict(float))
            d[ctr] += float(r["Counter_Value"]) * 1024 * mul
            d[ctr + "_n"] += 1
    print(f"[{tag}]")
    for k, d in acc.items():
        rd = d["FETCH_SIZE"] / max(d["FETCH_SIZE_n"], 1) / 1e6
        wr = d["WRITE_SIZE"] / max(d["WRITE_SIZE_n"], 1) / 1e6
        if rd + wr < 0.5: continue
        print(f"  {k:48s} read {rd:9.1f} MB  write {wr:9.1f} MB per launch ({int(d['FETCH_SIZE_n'])} launches)")
