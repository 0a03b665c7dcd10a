"""Debug driver: asm vs C++ Winograd kernel on structured inputs (prints where they differ)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
from mgunet import _lib
from mgunet import gat as G

cuda = torch.device("cuda:0")

ASM_FLAG = os.environ.get("DBG_ASM_FLAG", "1")
def conv(xin, wd, sc, sh, Cout, relu, flag):
    os.environ["MGU_WINO_ASM"] = ASM_FLAG if flag == "1" else flag
    G._CTX.clear()
    B, H, W, Cin = xin.shape
    out = torch.full((B, H, W, Cout), -7.0, device=cuda)
    ctx = G._context(cuda)
    rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, xin.data_ptr(), B, H, W, Cin, wd.data_ptr(), None, sc.data_ptr(), sh.data_ptr(),
                                    Cout, 3, relu, out.data_ptr(), Cout, 0, _lib.current_stream_ptr(cuda))
    _lib.check(rc, ctx.handle)
    torch.cuda.synchronize()
    return out.cpu()

def report(name, a, c):
    d = (a - c).abs()
    print(f"== {name}: max diff {float(d.max()):.3e}, wrong {int((a != c).sum())} / {a.numel()}")
    if float(d.max()) == 0:
        return
    B, H, W, N = a.shape
    wrong = (a != c)
    print("  wrong per channel (first 64):", wrong.sum((0, 1, 2))[:64].tolist())
    print("  wrong per row y:", wrong.sum((0, 2, 3))[:16].tolist())
    print("  wrong per col x:", wrong.sum((0, 1, 3))[:40].tolist())
    idx = wrong.nonzero()[:6]
    for i in idx:
        t = tuple(i.tolist())
        print("   ", t, float(a[t]), float(c[t]))

def case(name, B, H, W, Cin, Cout, xf, wf, relu=0):
    x = xf(B, H, W, Cin)
    w = wf(Cout, Cin)
    sc = torch.ones(Cout, device=cuda)
    sh = torch.zeros(Cout, device=cuda)
    a = conv(x.to(cuda), w.to(cuda), sc, sh, Cout, relu, "1")
    c = conv(x.to(cuda), w.to(cuda), sc, sh, Cout, relu, "0")
    report(name, a, c)
    return a, c

g = torch.Generator().manual_seed(0)
def xr(B, H, W, C): return torch.randn(B, H, W, C, generator=g)
def xcoord(B, H, W, C):
    y = torch.arange(H).view(1, H, 1, 1).float()
    x = torch.arange(W).view(1, 1, W, 1).float()
    c = torch.arange(C).view(1, 1, 1, C).float()
    return (y * 100 + x + c * 0.0 + 1.0).expand(B, H, W, C).contiguous()
def w_center_id(Co, Ci):
    w = torch.zeros(Co, Ci, 3, 3)
    for n in range(min(Co, Ci)):
        w[n, n, 1, 1] = 1.0
    return w
def w_center_c0(Co, Ci):
    w = torch.zeros(Co, Ci, 3, 3)
    w[:, 0, 1, 1] = 1.0
    return w
def wr(Co, Ci): return (torch.rand(Co, Ci, 3, 3, generator=g) - 0.5) * 0.4

which = sys.argv[1:] or ["all"]
a, c = case("1patch coord/center-c0", 1, 8, 32, 32, 64, xcoord, w_center_c0)
print("asm row0:", a[0, 0, :8, 0].tolist()); print("cpp row0:", c[0, 0, :8, 0].tolist())
print("asm col0:", a[0, :8, 0, 0].tolist()); print("cpp col0:", c[0, :8, 0, 0].tolist())
print("asm ch at (3,5):", a[0, 3, 5, :8].tolist(), a[0, 3, 5, 32:40].tolist())
case("1patch rand/center-id", 1, 8, 32, 32, 64, xr, w_center_id)
case("1patch rand/rand", 1, 8, 32, 32, 64, xr, wr)
case("2x2 patches rand", 1, 16, 64, 32, 64, xr, wr)
case("4 chunks", 1, 8, 32, 64, 64, xr, wr)
case("2 nblocks", 1, 8, 32, 32, 128, xr, wr)
case("many patches", 2, 128, 128, 32, 64, xr, wr)
case("N1 1patch 2ch", 1, 8, 32, 32, 32, xr, wr)
case("N1 1patch 4ch", 1, 8, 32, 64, 32, xr, wr)
case("N1 many 2ch", 2, 128, 128, 32, 32, xr, wr)
case("N1 many 4ch", 2, 128, 128, 64, 32, xr, wr, relu=1)
print("######## probes")
def x0(B, H, W, C): return torch.zeros(B, H, W, C)
def x1(B, H, W, C): return torch.ones(B, H, W, C)
def w0(Co, Ci): return torch.zeros(Co, Ci, 3, 3)
def probe(name, xf, wf, sh=0.25):
    B, H, W, Cin, Cout = 1, 8, 32, 32, 64
    x = xf(B, H, W, Cin); w = wf(Cout, Cin)
    sc = torch.ones(Cout, device=cuda); shv = torch.full((Cout,), sh, device=cuda)
    a = conv(x.to(cuda), w.to(cuda), sc, shv, Cout, 0, "1")
    c = conv(x.to(cuda), w.to(cuda), sc, shv, Cout, 0, "0")
    report(name, a, c)
    print("  asm[0,0,:4,:4]:", a[0, 0, :4, :4].tolist())
    print("  asm[0,3,8:12,30:34]:", a[0, 3, 8:12, 30:34].tolist())
    print("  cpp[0,0,:2,:4]:", c[0, 0, :2, :4].tolist())
probe("x=0, w rand", x0, wr)
probe("x rand, w=0", xr, w0)
probe("x=1, w center c0", x1, w_center_c0)
