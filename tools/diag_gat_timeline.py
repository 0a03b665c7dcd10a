#!/usr/bin/env python3
"""Phase latencies of one workgroup of gat_fused2_kernel from a DIAGNOSTIC build (hipcc -DMGU_DIAG=40 ... gat_fused.hip linked into a copy
of the library, MGU_LIB_PATH; never shipped): every stamp drains the wave's memory operations, so the intervals are phase LATENCIES.
    MGU_LIB_PATH=.../libmgunet_diag_g.so python tools/diag_gat_timeline.py [graphs]"""
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
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rowptr, col, gp, N, E = mgunet.PatchGraphConstructor(16).batched_csr(512, 512, G, dev)
gat = mgunet.GATNetwork(32, 128, 64, 4, 1)
gat.load_state_dict(O.make_gat_params(32, 128, 64, 4, 1, seed=0))
gat = gat.to(dev).eval()
X = torch.randn(N * G, 32, device=dev)
for _ in range(3):
    mgunet.gat_forward_csr(gat, X, rowptr, col, gp)
torch.cuda.synchronize()
L = _lib.lib()
L.mgu_diag_gat_read.restype = C.c_int
L.mgu_diag_gat_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((4, 8, 16), np.uint64)
assert L.mgu_diag_gat_read(buf.ctypes.data, buf.size) == 0
t = buf.astype(np.int64)
nt = int((t[0, :, 0] != 0).sum())
names = ["tile start", "(gmax)", "-", "rows in registers", "weights + aggregate + prefetch", "barrier A", "GEMM", "ELU + mean + stores"]
print(f"tiles recorded: {nt}")
used = [k for k in range(8) if (t[0, :nt, k] != 0).all()]
for i in range(1, len(used)):
    k, kp = used[i], used[i - 1]
    d = (t[:, :nt, k] - t[:, :nt, kp]).astype(np.float64)
    print(f"  {names[kp]:>30s} -> {names[k]:<30s} mean {d.mean():7.0f}   per tile (wave 0) {d[0].astype(int).tolist()}  (wave 3) {d[3].astype(int).tolist()}")
for k in []:
    d = (t[:, :nt, k] - t[:, :nt, k - 1]).astype(np.float64)
    print(f"  {names[k - 1]:>26s} -> {names[k]:<26s} mean {d.mean():7.0f}   per tile (wave 0) {d[0].astype(int).tolist()}")
if nt > 1:
    print(f"  tile period (wave 0): {(t[0, 1:nt, 0] - t[0, :nt - 1, 0]).tolist()}")
