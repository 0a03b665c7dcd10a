"""The hand-scheduled assembly form of the wide component-pair Winograd kernel (csrc/asm/gen_wino_cp.py, csrc/wino_asm.hip)
against the C++ kernel it replaces (wino3x3_cp_kernel<2>, csrc/wino_f32.hip): the same arithmetic in the same order, so the
outputs must be EQUAL BIT FOR BIT on every layer shape of the U-Net (model/unet/unet_encoder.py:15-25), and equal to torch
within the Winograd tolerance.  MGU_WINO_ASM is read by mgu_create: each arm runs in a context of its own."""
import pytest
import torch
import torch.nn.functional as F

import mgunet_oracle as O
from mgunet import _lib
from mgunet import gat as G

pytestmark = pytest.mark.gpu


def _conv(cuda, xin, wd, sc, sh, Cout, relu, ld, off):
    B, H, W, Cin = xin.shape
    out = torch.full((B, H, W, ld), -7.0, device=cuda)
    ctx = G._context(cuda)
    rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, xin.data_ptr(), B, H, W, Cin, wd.data_ptr(), None, sc.data_ptr(), sh.data_ptr(),
                                    Cout, 3, relu, out.data_ptr(), ld, off, _lib.current_stream_ptr(cuda))
    _lib.check(rc, ctx.handle)
    torch.cuda.synchronize()
    return out


# (B, H, W, Cin, Cout): the fourteen wide 3x3 layers of UNet(3, 2, 32, 4) at 512^2 scaled to small batches, plus walks of several
# patches per workgroup and of several n blocks
SHAPES = [
    (1, 256, 256, 32, 64),    # enc1.c1: two chunks
    (1, 256, 256, 64, 64),    # enc1.c2 / dec2.c2
    (2, 128, 128, 64, 128),   # enc2.c1: two n blocks
    (2, 128, 128, 128, 128),  # enc2.c2 / dec1.c2
    (4, 64, 64, 128, 256),    # enc3.c1
    (2, 64, 64, 256, 256),    # enc3.c2 / dec0.c2
    (8, 32, 32, 256, 512),    # bott.c1: one tile column, eight n blocks
    (4, 32, 32, 512, 512),    # bott.c2: 32 chunks
    (2, 64, 64, 512, 256),    # dec0.c1
    (1, 128, 128, 256, 128),  # dec1.c1
    (1, 256, 256, 128, 64),   # dec2.c1
    (8, 256, 256, 32, 64),    # eight patches per workgroup (the persistent walk, the cross-patch prefetch)
    (1, 8, 32, 32, 64),       # a single patch: every halo side outside the image
    (3, 40, 96, 96, 192),     # sizes that are not powers of two (five tile rows, three tile columns, six chunks, three n blocks)
    # the narrow kernels (32 output channels; the layer's weight pieces resident, two chunks of halo lead)
    (1, 128, 128, 32, 32),    # enc0.c2 / dec3.c2: two chunks
    (1, 128, 128, 64, 32),    # dec3.c1: four chunks
    (4, 256, 256, 32, 32),    # four patches per workgroup
    (2, 256, 256, 64, 32),
    (1, 8, 32, 32, 32),       # a single patch
    (3, 40, 96, 64, 32),      # sizes that are not powers of two
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
@pytest.mark.parametrize("relu", [1, 0])
def test_assembly_kernel_equals_cpp_kernel_bit_for_bit(cuda, B, H, W, Cin, Cout, relu, monkeypatch):
    if relu == 0 and B * H * W > 70000:
        pytest.skip("the no-ReLU epilogue is covered on the smaller shapes")
    x = torch.from_numpy(O.formula_normal("wa/x", (B, Cin, H, W), seed=H + Cin))
    w = torch.from_numpy(O.formula_uniform("wa/w", (Cout, Cin, 3, 3), -0.2, 0.2, seed=W + Cout))
    sc = torch.from_numpy(O.formula_uniform("wa/sc", (Cout,), 0.5, 1.5, seed=1)).to(cuda)
    sh = torch.from_numpy(O.formula_uniform("wa/sh", (Cout,), -0.5, 0.5, seed=2)).to(cuda)
    xin, wd = x.permute(0, 2, 3, 1).contiguous().to(cuda), w.contiguous().to(cuda)
    ld, off = Cout + 8, 4
    monkeypatch.setenv("MGU_WINO_ASM", "0")
    G._CTX.clear()
    cpp = _conv(cuda, xin, wd, sc, sh, Cout, relu, ld, off)
    monkeypatch.setenv("MGU_WINO_ASM", "1")
    G._CTX.clear()
    asm = _conv(cuda, xin, wd, sc, sh, Cout, relu, ld, off)
    monkeypatch.delenv("MGU_WINO_ASM")
    G._CTX.clear()
    assert torch.all(asm[..., :off] == -7.0) and torch.all(asm[..., off + Cout:] == -7.0)
    if not torch.equal(asm, cpp):
        bad = (asm != cpp).nonzero()
        raise AssertionError(f"{bad.shape[0]} of {asm.numel()} values differ; first at (b, y, x, c) = {bad[0].tolist()}: "
                             f"{asm[tuple(bad[0])].item()!r} vs {cpp[tuple(bad[0])].item()!r}; max |diff| "
                             f"{float((asm - cpp).abs().max()):.3e}")
    if B * H * W * Cin * Cout <= 2 ** 31:
        ref = F.conv2d(x, w, None, padding=1) * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)
        ref = F.relu(ref) if relu else ref
        got = asm[..., off:off + Cout].permute(0, 3, 1, 2).cpu()
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_assembly_kernel_in_the_forward_with_fused_pooling(cuda, monkeypatch):
    """Whole UNet forward (pooling rides in the encoder conv2 epilogue, skips land in the concat halves): both arms equal."""
    import mgunet
    cfg = (3, 2, 32, 4)
    p = O.make_unet_params(*cfg, seed=11)
    x = torch.from_numpy(O.formula_normal("wa/img", (2, 3, 128, 160), seed=3)).to(cuda)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MGU_WINO_ASM", flag)
        G._CTX.clear()
        unet = mgunet.UNet(*cfg)
        unet.load_state_dict(p)
        unet = unet.to(cuda).eval()
        lg, sk, ft = unet(x)
        torch.cuda.synchronize()
        outs.append([lg.clone()] + [t.clone() for t in sk] + [t.clone() for t in ft])
        del unet
    monkeypatch.delenv("MGU_WINO_ASM")
    G._CTX.clear()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_headline_forward_on_the_assembly_kernels_is_bitwise_repeatable(cuda):
    """The assembly kernels wait on hand-counted vmcnt / lgkmcnt values: an under-counted wait reads a register before its load has
    landed only when the memory system is slow enough, i.e. not in every launch.  40 forwards of the headline shape (8 x 3 x 512 x 512,
    every one of the 17 Winograd layers on its assembly kernel, 256 persistent workgroups each) must give the same bytes."""
    import mgunet
    cfg = (3, 2, 32, 4)
    p = O.make_unet_params(*cfg, seed=5)
    x = torch.from_numpy(O.formula_normal("wa/rep", (8, 3, 512, 512), seed=4)).to(cuda)
    unet = mgunet.UNet(*cfg)
    unet.load_state_dict(p)
    unet = unet.to(cuda).eval()
    lg, sk, ft = unet(x)
    torch.cuda.synchronize()
    first = [lg.clone()] + [t.clone() for t in ft]
    assert all(bool(torch.isfinite(t).all()) for t in first)
    for _ in range(40):
        lg, sk, ft = unet(x)
        torch.cuda.synchronize()
        for a, b in zip(first, [lg] + list(ft)):
            assert torch.equal(a, b)
