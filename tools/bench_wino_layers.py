"""Per-layer timing of the wide 3x3 layers of UNet(3,2,32,4) at 8 x 512^2 (the headline shapes): C++ kernel vs assembly kernel,
interleaved rounds in ONE process (HIP events).  usage: bench_wino_layers.py [rounds]"""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
from mgunet import _lib

cuda = torch.device("cuda:0")
LAYERS = [("enc1.c1", 256, 32, 64), ("enc1.c2", 256, 64, 64), ("enc2.c1", 128, 64, 128), ("enc2.c2", 128, 128, 128),
          ("enc3.c1", 64, 128, 256), ("enc3.c2", 64, 256, 256), ("bott.c1", 32, 256, 512), ("bott.c2", 32, 512, 512),
          ("dec0.c1", 64, 512, 256), ("dec0.c2", 64, 256, 256), ("dec1.c1", 128, 256, 128), ("dec1.c2", 128, 128, 128),
          ("dec2.c1", 256, 128, 64), ("dec2.c2", 256, 64, 64),
          ("enc0.c2", 512, 32, 32), ("dec3.c1", 512, 64, 32), ("dec3.c2", 512, 32, 32)]
B = 8
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FLAGS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1"]
L = _lib.lib()
ctxs = {}
for flag in FLAGS:
    os.environ["MGU_WINO_ASM"] = flag
    ctxs[flag] = _lib.Context(0)
stream = _lib.current_stream_ptr(cuda)
g = torch.Generator().manual_seed(0)
tot = {f: 0.0 for f in FLAGS}
print(f"{'layer':8s} {'GF':>6s} " + " ".join(f"{'v' + f:>8s}" for f in FLAGS))
for name, S, Cin, Cout in LAYERS:
    x = torch.randn(B, S, S, Cin, generator=g).to(cuda)
    w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.2).to(cuda)
    sc = torch.ones(Cout, device=cuda); sh = torch.zeros(Cout, device=cuda)
    out = torch.empty(B, S, S, Cout, device=cuda)
    hs = {}
    for flag, ctx in ctxs.items():
        h = C.c_void_p()
        _lib.check(L.mgu_conv2d_prepare(ctx.handle, w.data_ptr(), Cout, Cin, 3, C.byref(h), stream), ctx.handle)
        hs[flag] = h
    def run(flag, n):
        ctx = ctxs[flag]
        for _ in range(n):
            L.mgu_conv2d_prepared_nhwc(ctx.handle, hs[flag], x.data_ptr(), B, S, S, None, sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), Cout, 0, stream)
    best = {f: 1e9 for f in FLAGS}
    for flag in FLAGS:
        run(flag, 3)
    torch.cuda.synchronize()
    for r in range(rounds):
        for flag in FLAGS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(flag, 20); e1.record(); torch.cuda.synchronize()
            best[flag] = min(best[flag], e0.elapsed_time(e1) / 20 * 1000)
    gf = 2 * B * S * S * Cin * Cout * 9 / 1e9
    for f in FLAGS:
        tot[f] += best[f]
    print(f"{name:8s} {gf:6.1f} " + " ".join(f"{best[f]:8.1f}" for f in FLAGS))
    for flag, ctx in ctxs.items():
        L.mgu_conv2d_release(ctx.handle, hs[flag])
print(f"{'total':8s} {'':6s} " + " ".join(f"{tot[f]:8.1f}" for f in FLAGS))
