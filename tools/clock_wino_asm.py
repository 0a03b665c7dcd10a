"""In-kernel clock of the stamped timing variants of the asm Winograd kernel: cycles / 100 MHz ticks per workgroup."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
from mgunet import _lib
cuda = torch.device("cuda:0")
L = _lib.lib()
B, S, Cin, Cout = 8, 32, 512, 512
name = "bott.c2"
if len(sys.argv) > 1 and sys.argv[1] == "enc1":
    B, S, Cin, Cout, name = 8, 256, 64, 64, "enc1.c2"
g = torch.Generator().manual_seed(0)
x = torch.randn(B, S, S, Cin, generator=g).to(cuda)
w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.2).to(cuda)
sc = torch.ones(Cout, device=cuda); sh = torch.zeros(Cout, device=cuda)
out = torch.zeros(B, S, S, Cout, device=cuda)
stream = _lib.current_stream_ptr(cuda)
for var in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("9", "10", "11")):
    os.environ["MGU_WINO_ASM"] = var
    ctx = _lib.Context(0)
    h = C.c_void_p()
    _lib.check(L.mgu_conv2d_prepare(ctx.handle, w.data_ptr(), Cout, Cin, 3, C.byref(h), stream), ctx.handle)
    def run(n):
        for _ in range(n):
            L.mgu_conv2d_prepared_nhwc(ctx.handle, h, x.data_ptr(), B, S, S, None, sc.data_ptr(), sh.data_ptr(), 1, out.data_ptr(), Cout, 0, stream)
    run(2000)      # sustained load before the measured launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(200); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1000
    st = out.view(-1)[:512].view(torch.int32).view(256, 2).cpu()
    cyc, ticks = st[:, 0].float(), st[:, 1].float()
    ghz = (cyc / ticks * 0.1)
    print(f"{name} variant {var}: {us:.1f} us/launch; workgroup life cycles median {cyc.median():.0f}, 100MHz ticks median {ticks.median():.0f}; "
          f"clock median {ghz.median():.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f}); cycles/us {cyc.median() / us:.0f}")
    L.mgu_conv2d_release(ctx.handle, h)
