"""Debug: run the asm kernel built with GEN_WINO_DEBUG=dump_*: out holds 32 registers per thread."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
from mgunet import _lib
from mgunet import gat as G
cuda = torch.device("cuda:0")
os.environ["MGU_WINO_ASM"] = "1"
B, H, W, Cin, Cout = 1, 8, 32, 32, 64
y = torch.arange(H).view(1, H, 1, 1).float(); x = torch.arange(W).view(1, 1, W, 1).float(); c = torch.arange(Cin).view(1, 1, 1, Cin).float()
xin = (y * 10000 + x * 100 + c + 1).expand(B, H, W, Cin).contiguous().to(cuda)      # value encodes (y, x, channel)
w = torch.zeros(Cout, Cin, 3, 3); w[:, :, 1, 1] = 1.0
sc = torch.ones(Cout, device=cuda); sh = torch.zeros(Cout, device=cuda)
out = torch.full((B, H, W, Cout), -7.0, device=cuda)
ctx = G._context(cuda)
rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, xin.data_ptr(), B, H, W, Cin, w.to(cuda).data_ptr(), None, sc.data_ptr(), sh.data_ptr(),
                                Cout, 3, 0, out.data_ptr(), Cout, 0, _lib.current_stream_ptr(cuda))
torch.cuda.synchronize()
d = out.cpu().view(-1)[:512 * 32].view(512, 32)
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
for tid in (0, 1, 15, 16, 31, 32, 33, 63, 64, 65, 128, 192, 256, 257, 320, 511):
    print("tid", tid, d[tid].tolist())
import numpy as np
np.save("gpurun_out/dump.npy", d.numpy())
