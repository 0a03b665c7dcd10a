#!/usr/bin/env python3
"""Per-layer conv timings of the last bench step from a rocprofv3 kernel trace (B=8, 512x512)."""
import csv, glob, sys
root = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
import os
rows = sorted(csv.DictReader(open(max(glob.glob(root + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime))),
              key=lambda r: int(r["Start_Timestamp"]))   # (the trace file is not always in dispatch order)
ig = [r for r in rows if any(t in r["Kernel_Name"] for t in ("igemm", "convt2x2", "conv3x3_halo", "conv3x3_first", "wino3x3", "mgu_wino_cp", "head_kernel", "patch_mean_kernel"))]
last = ig[-23:]   # the 23 conv launches of the last U-Net forward
def convf(h, cin, cout): return 2 * B * h * h * 9 * cin * cout
layers = []
h, cin, f = S, 3, 32
for i in range(4):
    layers += [("enc%d.c1" % i, convf(h, cin, f)), ("enc%d.c2" % i, convf(h, f, f))]; cin = f; f *= 2; h //= 2
layers += [("bott.c1", convf(h, cin, f)), ("bott.c2", convf(h, f, f))]
prev = f
for b in range(4):
    c = 32 << (3 - b)
    layers.append(("dec%d.up" % b, 2 * B * h * h * prev * (prev // 2) * 4)); h *= 2
    layers += [("dec%d.c1" % b, convf(h, 2 * c, c)), ("dec%d.c2" % b, convf(h, c, c))]; prev = c
layers.append(("final", 2 * B * S * S * 32 * 2))
tot = 0
for (name, fl), r in zip(layers, last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    kname = r["Kernel_Name"]
    tag = "head<" if ("head_kernel" in kname or "patch_mean" in kname) else "first<" if "conv3x3_first" in kname else "wino<" if "wino3x3" in kname else "wino_asm<" + kname.split("_gfx950")[0][-5:] + " " if "mgu_wino_cp" in kname else "halo<" if "halo" in kname else "convt_x3<" if "convt2x2" in kname else "igemm<"
    if "mgu_wino_cp" in kname:      # hand-written code objects: mgu_wino_cp2_gfx950 (wide), mgu_wino_cp1r2/4_gfx950 (narrow, 2 / 4 resident chunks)
        kn = "wino_asm<" + kname.split("mgu_wino_")[1].split("_gfx950")[0] + ">"
    else:
        kn = tag + (kname.split("<")[1].split(">")[0].replace(" ", "") if "<" in kname else "?") + ">"
    print(f"{name:9s} {kn:22s} grid={r['Grid_Size_X']:>9s}x{r['Grid_Size_Y']:>4s} {d:8.1f} us {fl/d/1e6:7.1f} TF/s vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']} lds={r['LDS_Block_Size']}")
print("total us", round(tot, 1))
oth = [r for r in rows if r not in ig]

# idle time between consecutive kernels of the last step (dispatch gaps), including the kernels that are not convolutions
allk = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
i1 = allk.index(last[-1])
i0 = allk.index(last[0])
seg = allk[i0:i1 + 1]                  # first conv .. head of the last forward
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(seg, seg[1:])]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg)
print(f"step span {span:.1f} us over {len(seg)} kernels: busy {busy:.1f} us, gaps {sum(gaps):.1f} us (mean {sum(gaps)/max(1,len(gaps)):.2f}, max {max(gaps):.1f})")
