"""Soak run of the aggregate-first GAT layer (gat_stmax_kernel + gat_fused2_kernel): REPS consecutive calls per case must give the same
bytes, with and without a second stream streaming through HBM/L2 beside them (timing perturbation).  Cases: 64 patch graphs (32 -> 4 x 64)
and configs[3]'s 32 stress graphs (64 -> 4 x 64, in-degree 8), head mean and concat.  Prints one line per case; exit code 1 on a mismatch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
import mgunet  # noqa: E402
from mgunet import _lib  # noqa: E402

REPS = int(os.environ.get("REPS", "2000"))
cuda = torch.device("cuda:0")
rng = np.random.default_rng(5)


def case(kind, concat, noise):
    ctx = _lib.Context(0)
    heads, Fh = 4, 64
    if kind == "patch":
        Fin, G = 32, 64
        rowptr, col, gp, N1, E1 = mgunet.PatchGraphConstructor(16).batched_csr(512, 512, G, cuda)
    else:
        Fin, G, N1, deg = 64, 32, 2048, 8
        src = rng.integers(0, N1, size=(G, N1 * deg))
        col = torch.from_numpy((src + (np.arange(G) * N1)[:, None]).reshape(-1).astype(np.int32)).to(cuda)
        rowptr = torch.from_numpy((np.arange(G * N1 + 1) * deg).astype(np.int32)).to(cuda)
        gp = torch.from_numpy((np.arange(G + 1) * N1).astype(np.int32)).to(cuda)
    N = N1 * G
    X = torch.from_numpy(rng.standard_normal((N, Fin)).astype(np.float32)).to(cuda)
    W = torch.from_numpy(rng.uniform(-0.4, 0.4, (heads * Fh, Fin)).astype(np.float32)).to(cuda)
    a = torch.from_numpy(rng.uniform(-0.4, 0.4, (heads, 2 * Fh)).astype(np.float32)).to(cuda)
    L = _lib.lib()
    h = C.c_void_p()
    s = _lib.current_stream_ptr(cuda)
    _lib.check(L.mgu_gat_prepare(ctx.handle, W.data_ptr(), a.data_ptr(), heads, Fh, Fin, 1, C.byref(h), s), ctx.handle)
    Fo = heads * Fh if concat else Fh
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device=cuda) if noise else None
    first, nbad = None, 0
    outs = [torch.empty((N, Fo), device=cuda) for _ in range(2)]
    for i in range(REPS):
        if noise and i % 3 == 0:
            with torch.cuda.stream(side):
                big.mul_(1.0001)
        out = outs[i & 1]
        out.fill_(float("nan"))
        _lib.check(L.mgu_gat_layer_forward_prepared(ctx.handle, h, X.data_ptr(), N, rowptr.data_ptr(), col.data_ptr(), col.numel(), gp.data_ptr(), G,
                                                    concat, 0.2, out.data_ptr(), s), ctx.handle)
        if first is None:
            first = out.clone()
        else:
            bad = int((out != first).any(1).sum())
            if bad:
                nbad += 1
                if nbad <= 3:
                    print(f"  mismatch at call {i}: {bad} rows", flush=True)
    torch.cuda.synchronize()
    L.mgu_gat_release(ctx.handle, h)
    print(f"{kind:7s} concat={concat} noise={noise}: {nbad} of {REPS - 1} calls differ from the first", flush=True)
    return nbad


total = 0
for kind in ("patch", "stress"):
    for concat in (0, 1):
        for noise in (0, 1):
            total += case(kind, concat, noise)
sys.exit(1 if total else 0)
