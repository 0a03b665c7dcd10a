#!/usr/bin/env python3
"""profiles/r02_traffic_<tag>.json from the FETCH_SIZE / WRITE_SIZE PMC passes of tools/gpu_pmc.sh: HBM bytes per launch of every
kernel family, averaged over all launches of the run (bytes do not depend on warm-up).  FETCH_SIZE is doubled (gfx950 tallies
128-B requests at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores.
usage: traffic_kernels.py ROOT OUT.json "command that was profiled" """
import collections, csv, glob, json, os, re, sys

root, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""


def per_dispatch(tag, counter):
    f = max(glob.glob(f"{root}/pmc_{tag}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        acc.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    return [acc[k] for k in sorted(acc)]


def short(n):
    n = n.replace("void ", "").replace("mgu::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else n[:60]


agg = collections.OrderedDict()
for tag, counter, scale in (("fetch", "FETCH_SIZE", 2 * 1024.0), ("write", "WRITE_SIZE", 1024.0)):
    for name, v in per_dispatch(tag, counter):
        e = agg.setdefault(short(name), {"fetch_n": 0, "write_n": 0, "read_bytes": 0.0, "write_bytes": 0.0})
        e[tag + "_n"] += 1
        e["read_bytes" if tag == "fetch" else "write_bytes"] += v * scale
kern = {}
for k, e in agg.items():
    if not e["fetch_n"] or k.startswith("at::") or "elementwise" in k and "mgu" not in k:
        continue
    rd, wr = e["read_bytes"] / e["fetch_n"], e["write_bytes"] / max(e["write_n"], 1)
    kern[k] = {"launches": e["fetch_n"], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
def lib_build_id():
    """first 16 hex digits of the SHA-256 of the library the passes ran (bench.py drops `roofline.traffic` when it differs from the
    library it is running)"""
    import hashlib
    lib = os.environ.get("MGU_LIB_PATH") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mingraph-unet_amd", "lib",
                                                         "libmgunet.so")
    return hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None


j = {"lib_build_id": lib_build_id(),
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/gpu_pmc.sh + tools/traffic_kernels.py) over `"
               + cmd + "`",
     "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md HBM section), WRITE_SIZE exact",
     "kernels": kern,
     "hbm_bytes_all_launches": round(sum(e["read_bytes"] + e["write_bytes"] for e in agg.values()))}
json.dump(j, open(out, "w"), indent=1)
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:24]:
    print(f"{k[:70]:70s} n={v['launches']:4d} rd {v['read_bytes_per_launch']/1e6:9.1f} MB  wr {v['write_bytes_per_launch']/1e6:9.1f} MB")
