"""Per-step timeline of chunk 8 of the asm Winograd kernel (variant _v15 of a GEN_WINO_VARIANTS=1 build): cycles between stamps."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
from mgunet import _lib
cuda = torch.device("cuda:0")
L = _lib.lib()
B, S, Cin, Cout = 8, 32, 512, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(B, S, S, Cin, generator=g).to(cuda)
w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.2).to(cuda)
sc = torch.ones(Cout, device=cuda); sh = torch.zeros(Cout, device=cuda)
out = torch.zeros(B, S, S, Cout, device=cuda)
stream = _lib.current_stream_ptr(cuda)
os.environ["MGU_WINO_ASM"] = sys.argv[1] if len(sys.argv) > 1 else "16"
ctx = _lib.Context(0)
h = C.c_void_p()
_lib.check(L.mgu_conv2d_prepare(ctx.handle, w.data_ptr(), Cout, Cin, 3, C.byref(h), stream), ctx.handle)
for _ in range(500):
    L.mgu_conv2d_prepared_nhwc(ctx.handle, h, x.data_ptr(), B, S, S, None, sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), Cout, 0, stream)
torch.cuda.synchronize()
st = out.view(-1)[:256 * 8 * 8].view(torch.int32).view(256, 8, 8).cpu().long()
names = ["B0 wait", "stage", "step0", "step1", "step2", "B1 wait", "step3a", "step3b"]
d = (st[:, :, 1:] - st[:, :, :-1])              # 7 intervals
tot = st[:, :, 7] - st[:, :, 0]
print("intervals (cycles), median over workgroups, per wave (rows) x [B0wait, stage, step0, step1, (spread: B1wait, step2 | pingpong: step2, B1wait), step3] ; total")
for wv in range(8):
    print(f"wave {wv} (wi {wv & 3}, jp {wv >> 2}):", [int(d[:, wv, k].median()) for k in range(7)], int(tot[:, wv].median()))
print("all waves median:", [int(d[:, :, k].median()) for k in range(7)], int(tot.median()))
for wg in (0, 100):
    print(f"wg {wg} absolute (relative to wave 0's t0):")
    for wv in range(8):
        print("   wave", wv, (st[wg, wv] - st[wg, 0, 0]).tolist())
