#!/usr/bin/env python3
"""The kernels of ONE warm step in dispatch order, from a rocprofv3 --kernel-trace CSV: name (shortened), duration, gap to the
previous kernel's end.  The last step is taken: the trace is cut at the last launch of MARKER (default pack_input_kernel, the first
kernel of a forward).  usage: step_sequence.py TRACE_DIR [MARKER]"""
import csv, glob, os, re, sys

root = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "pack_input_kernel"
f = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit(f"marker {marker} seen {len(starts)} times")
seq = rows[starts[-2]:starts[-1]]
t0 = int(seq[0]["Start_Timestamp"])
prev_end, busy = None, 0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mgu::", "").replace("(anonymous namespace)::", "")
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.2f} us  gap {gap:6.2f}  {name[:80]}")
    prev_end, busy = e, busy + (e - s)
print(f"step span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, {len(seq)} kernels")
