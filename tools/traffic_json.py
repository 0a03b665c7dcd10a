#!/usr/bin/env python3
"""profiles/r01_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes of tools/gpu_pmc.sh: HBM bytes of the 23
conv launches (3x3 Winograd / direct, ConvTranspose, 1x1 head) of the LAST U-Net forward of the run."""
import csv, glob, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_traffic.json"
summary = sys.argv[3] if len(sys.argv) > 3 else "profiles/r01_j_pmc_summary.txt"

def per_dispatch(tag, counter):
    f = max(glob.glob(f"{root}/pmc_{tag}/runc/*counter_collection.csv"), key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        acc.setdefault(k, [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    return [acc[k] for k in sorted(acc)]

def conv(name):
    return any(t in name for t in ("wino3x3", "conv3x3_halo", "conv3x3_first", "igemm_kernel", "conv1x1_head", "patch_mean_kernel"))

def last_step(rows):
    c = [r for r in rows if conv(r[0])]
    return c[-23:]                 # the 23 conv launches of the last U-Net forward (the GAT no longer launches a GEMM)

rd = last_step(per_dispatch("fetch", "FETCH_SIZE"))
wr = last_step(per_dispatch("write", "WRITE_SIZE"))
assert len(rd) == 23 and len(wr) == 23, (len(rd), len(wr))
read_b = sum(v for _, v in rd) * 1024 * 2      # KiB, and the gfx950 x2 correction
write_b = sum(v for _, v in wr) * 1024
B, S = 8, 512
# algorithmic (unfused, every conv reads its input once and writes its output once), fp32
alg = 6144000000.0
j = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline --no-profile-pass`; tools/gpu_pmc.sh + tools/traffic_json.py; summary " + summary,
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md HBM section; checked: the 1x1 head reads "
                  "268.5 MB corrected vs 268.4 MB algorithmic); WRITE_SIZE exact",
    "kernels": "conv3x3_first_kernel + wino3x3_f32_kernel (17) + igemm_kernel (4 ConvTranspose) + patch_mean_kernel<float,2> (1x1 head fused with the patch means): the 23 conv launches of one U-Net forward, B=8 3x512x512",
    "launches_per_step": 23,
    "read_bytes_per_step": read_b,
    "write_bytes_per_step": write_b,
    "hbm_bytes_per_launch": (read_b + write_b) / 23,
    "algorithmic_unfused_bytes_per_step": alg,
}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j, indent=1))
