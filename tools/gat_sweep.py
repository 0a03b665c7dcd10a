#!/usr/bin/env python3
"""Wh-row gather schedule of the GAT layer (gat_aggregate_kernel) over a sweep of batch sizes: does the time per node depend on where
the (N, 1 KiB) node table lives (L2 / Infinity Cache / HBM)?   MGU_NO_GAT_FUSED=1 python tools/gat_sweep.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["MGU_NO_GAT_FUSED"] = "1"
import torch
import mgunet, mgunet_oracle as O
from mgunet import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
graph = mgunet.PatchGraphConstructor(16)
heads, Fh, Fin = 4, 64, 32
gp_ = O.make_gat_params(Fin, 128, 64, 4, 1, seed=0)
W = torch.cat([gp_[f"gat_layers.0.heads.{h}.W.weight"] for h in range(heads)], 0).contiguous().to(dev)
a = torch.cat([gp_[f"gat_layers.0.heads.{h}.a.weight"] for h in range(heads)], 0).contiguous().to(dev)
for G in (8, 16, 32, 64, 128, 256, 512):
    rowptr, col, gp, N1, E1 = graph.batched_csr(512, 512, G, dev)
    N, E = N1 * G, E1 * G
    X = torch.randn((N, Fin), device=dev)
    ctx = _lib.Context(0)
    hnd = C.c_void_p()
    s = _lib.current_stream_ptr(dev)
    _lib.check(L.mgu_gat_prepare(ctx.handle, W.data_ptr(), a.data_ptr(), heads, Fh, Fin, 1, C.byref(hnd), s), ctx.handle)
    y = torch.empty((N, Fh), device=dev)
    def run():
        _lib.check(L.mgu_gat_layer_forward_prepared(ctx.handle, hnd, X.data_ptr(), N, rowptr.data_ptr(), col.data_ptr(), E, gp.data_ptr(), G, 0, 0.2,
                                                    y.data_ptr(), s), ctx.handle)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    L.mgu_profile_enable(ctx.handle, 1)
    reps = 20
    for _ in range(reps):
        run()
    ks = {k["name"]: k["ms"] * 1e3 / reps for k in _lib.read_kernel_stats(ctx)}
    L.mgu_profile_enable(ctx.handle, 0)
    us = ks.get("gat_aggregate_kernel", float("nan"))
    table_mb = N * heads * Fh * 4 / 1e6
    comp = N * heads * Fh * 4 + N * 2 * heads * 4 + (N + 1 + E) * 4 + N * 4 + N * Fh * 4
    print(f"graphs {G:4d}  nodes {N:7d}  Wh table {table_mb:7.1f} MB  aggregate {us:8.2f} us  {us * 1e3 / N:6.3f} ns/node  "
          f"compulsory {comp / us / 1e6:6.2f} TB/s  logical {E * 1032 / us / 1e6:6.2f} TB/s   all: " + " ".join(f"{k.split('_kernel')[0]}={v:.1f}" for k, v in ks.items()), flush=True)
    L.mgu_gat_release(ctx.handle, hnd)
    del ctx
