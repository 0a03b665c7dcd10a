#!/usr/bin/env python3
"""Step timeline of one workgroup of the bf16 halo convolution kernel from a DIAGNOSTIC build (make CXXFLAGS+=' -DMGU_DIAG=23
-DMGU_DIAG_H=.. -DMGU_DIAG_CP=.. -DMGU_DIAG_N=..'; igemm.hip: HALO_T): per step (one tap of one 64-channel chunk) every wave stamps
s_memtime before / after the step's barrier, after the MFMA block and at the end of the step."""
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
model = mgunet.UNet(3, 2, 32, 4, compute_dtype=torch.bfloat16)
model.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
model = model.to(dev).eval()
x = torch.from_numpy(O.formula_normal("bench/x", (8, 3, 512, 512), seed=1)).to(dev)
for _ in range(3):
    with torch.no_grad():
        model(x)
torch.cuda.synchronize()
L = _lib.lib()
L.mgu_diag_read.restype = C.c_int
L.mgu_diag_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((4, 256, 4), np.uint64)
assert L.mgu_diag_read(buf.ctypes.data, buf.size) == 0
t = buf.astype(np.int64)
n = int((t[0, :, 0] != 0).sum())
print("steps recorded:", n)
t = t[:, :n, :]
per = t[:, 1:, 0] - t[:, :-1, 0]
print(f"step period: mean {per.mean():.0f}  median {np.median(per):.0f}  (16 MFMAs = 512 matrix-pipe cycles per wave; two workgroups share the CU)")
print(f"  barrier wait        mean {(t[:, :, 1] - t[:, :, 0]).mean():7.0f}  per wave {np.round((t[:, :, 1] - t[:, :, 0]).mean(axis=1)).astype(int).tolist()}")
print(f"  barrier -> MFMAs issued  {(t[:, :, 2] - t[:, :, 1]).mean():7.0f}  per wave {np.round((t[:, :, 2] - t[:, :, 1]).mean(axis=1)).astype(int).tolist()}")
print(f"  tail of the step    mean {(t[:, :, 3] - t[:, :, 2]).mean():7.0f}")
long = np.argsort(-per[0])[:8]
print("  longest steps (wave 0):", [(int(i), int(per[0, i])) for i in sorted(long)])
# per tap of an item (9 steps per 64-channel chunk): where the waits land
nt = int(os.environ.get('NT', '9'))   # steps per item: 9 (one tap per step) or 3 (the N <= 32 tile: three taps per step)
m = (n // nt) * nt
if m >= 2 * nt:
    tt = t[:, nt:m, :].reshape(4, -1, nt, 4)   # skip the first item (cold)
    bw = (tt[..., 1] - tt[..., 0]).mean(axis=(0, 1))
    mf = (tt[..., 2] - tt[..., 1]).mean(axis=(0, 1))
    tl = (tt[..., 3] - tt[..., 2]).mean(axis=(0, 1))
    print("  per tap: barrier wait   ", np.round(bw).astype(int).tolist())
    print("           barrier->MFMAs ", np.round(mf).astype(int).tolist())
    print("           tail           ", np.round(tl).astype(int).tolist())
