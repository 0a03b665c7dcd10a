#!/usr/bin/env python3
"""Phase timeline of one workgroup of a Winograd component-pair layer, from a DIAGNOSTIC build of the library
(make CXXFLAGS+=' -DMGU_DIAG=20 -DMGU_DIAG_H=512 -DMGU_DIAG_CP=32 -DMGU_DIAG_N=32'; never shipped): every wave's lane 0 stamps
s_memtime at the phase boundaries of every patch (wino_f32.hip: DIAG_T).  Prints, per phase, the mean over patches and waves of the
time spent in it, in counter ticks and as a share of the patch time.
    python tools/diag_timeline.py [steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mgunet  # noqa: E402
import mgunet_oracle as O  # noqa: E402
from mgunet import _lib  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 8, 512, 512
model = mgunet.UNet(3, 2, 32, 4)
model.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
model = model.to(dev).eval()
x = torch.from_numpy(O.formula_normal("bench/x", (B, 3, H, W), seed=1)).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    with torch.no_grad():
        model(x)
torch.cuda.synchronize()
L = _lib.lib()
L.mgu_diag_read.restype = C.c_int
L.mgu_diag_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((8, 64, 32), np.uint64)
rc = L.mgu_diag_read(buf.ctypes.data, buf.size)
assert rc == 0, rc
ts = buf.astype(np.int64)
npatch = int((ts[0, :, 0] != 0).sum())
print(f"patches recorded: {npatch}")
names = {0: "patch start", 1: "B0 c0", 2: "->B1 c0", 3: "B1 c0", 4: "B0 c1", 5: "->B1 c1", 6: "B1 c1", 7: "B0 c2", 8: "->B1 c2", 9: "B1 c2",
         10: "B0 c3", 11: "->B1 c3", 12: "B1 c3", 13: "->epilogue", 14: "epi barrier 0", 15: "->shares written", 16: "shares barrier",
         17: "->finish issued", 18: "final barrier"}
used = [k for k in sorted(names) if (ts[0, :npatch, k] != 0).all()]
t = ts[:, :npatch, :]
per_patch = (t[:, 1:, 0] - t[:, :-1, 0])
print(f"patch period: mean {per_patch.mean():.0f} ticks (min {per_patch.min()}, max {per_patch.max()})")
prev = used[0]
tot = 0.0
for k in used[1:]:
    dlt = (t[:, :, k] - t[:, :, prev]).astype(np.float64)
    print(f"  {names[prev]:>18s} -> {names[k]:<18s} mean {dlt.mean():8.0f}  per wave {np.round(dlt.mean(axis=1)).astype(int).tolist()}")
    tot += dlt.mean()
    prev = k
last = (t[:, 1:, 0] - t[:, :-1, used[-1]]).astype(np.float64)
print(f"  {names[used[-1]]:>18s} -> next patch start    mean {last.mean():8.0f}")
print(f"sum of phases {tot + last.mean():.0f}")
