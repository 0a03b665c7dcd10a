#!/usr/bin/env python3
"""The aggregate-first GAT layer (gat_stmax + gat_fused) over a range of batch sizes: per-kernel time by HIP events against the
number of 32-node tiles -- is the fused kernel's time a function of the tiles per resident workgroup (rounds) or of the bytes?
    python tools/gat_tiles_sweep.py [graphs ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import mgunet  # noqa: E402
import mgunet_oracle as O  # noqa: E402
from mgunet import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
heads, Fh, Fin = 4, 64, 32
graph = mgunet.PatchGraphConstructor(16)
ctx = _lib.Context(0)
gp_ = O.make_gat_params(Fin, 128, 64, 4, 1, seed=0)
W = torch.cat([gp_[f"gat_layers.0.heads.{h}.W.weight"] for h in range(heads)], 0).contiguous().to(dev)
a = torch.cat([gp_[f"gat_layers.0.heads.{h}.a.weight"] for h in range(heads)], 0).contiguous().to(dev)
hnd = C.c_void_p()
s = _lib.current_stream_ptr(dev)
_lib.check(L.mgu_gat_prepare(ctx.handle, W.data_ptr(), a.data_ptr(), heads, Fh, Fin, 1, C.byref(hnd), s), ctx.handle)
for G in [int(v) for v in sys.argv[1:]] or [8, 16, 24, 32, 48, 64, 96, 128, 192, 256]:
    rowptr, col, gp, N1, E1 = graph.batched_csr(512, 512, G, dev)
    N, E = N1 * G, E1 * G
    X = torch.randn((N, Fin), device=dev)
    y = torch.empty((N, Fh), device=dev)

    def run():
        _lib.check(L.mgu_gat_layer_forward_prepared(ctx.handle, hnd, X.data_ptr(), N, rowptr.data_ptr(), col.data_ptr(), E, gp.data_ptr(), G, 0, 0.2,
                                                    y.data_ptr(), s), ctx.handle)
    for _ in range(5):
        run()
    torch.cuda.synchronize(dev)
    reps = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    layer = e0.elapsed_time(e1) * 1e3 / reps
    L.mgu_profile_enable(ctx.handle, 1)
    for _ in range(reps):
        run()
    ks = _lib.read_kernel_stats(ctx)
    L.mgu_profile_enable(ctx.handle, 0)
    kern = {k["name"]: round(k["ms"] * 1e3 / reps, 2) for k in ks}
    print(f"graphs {G:4d} tiles {N // 32:6d} layer {layer:7.2f} us  {kern}")
