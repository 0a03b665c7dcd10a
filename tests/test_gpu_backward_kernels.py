"""Every backward kernel in isolation against FLOAT64 torch on fixed (x, dz): the whole-step record (test_gpu_train.py) sits
behind train-mode BatchNorm + ReLU + MaxPool, whose conditioning hides a 1 % kernel error; here nothing is ill-conditioned and
the bar is 2e-5 of the result's max (the forward kernels' bar).  Entry points: include/mgunet.h "backward building blocks" --
they run the launchers mgu_unet_backward uses.  Kernel variants are reached through the per-context MGU_* switches.

Replaces the autograd nodes of scripts/train_segmentation.py:133 for model/unet/unet_encoder.py:7-25, unet_decoder.py:25-55."""
import contextlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mgunet_oracle as O
from mgunet import _lib

pytestmark = pytest.mark.gpu

TOL = 2e-5


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@contextlib.contextmanager
def context(cuda, **env):
    """A fresh mgu_ctx created under the given MGU_* switches (they are read at mgu_create and live in the context)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        ctx = _lib.Context(cuda.index or 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield ctx
    torch.cuda.synchronize()


def rel(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


VARIANTS = {"default": {}, "no_wino_wgrad": {"MGU_NO_WINO_WGRAD": 1}, "tiles": {"MGU_NO_WINO_WGRAD": 1, "MGU_NO_WGRAD_HALO": 1, "MGU_NO_THIN_WGRAD": 1},
            "fp32_mfma": {"MGU_WINO_PREC": 0}, "wgrad_fp32_mfma": {"MGU_NO_WGRAD_X3": 1}, "convt_dgrad_tiles": {"MGU_NO_CONVT_DGRAD_X3": 1}}

WGRAD_CASES = [  # B, H, W, Cin, Cout, k
    (2, 13, 17, 8, 32, 3),      # generic tile kernel, ragged
    (1, 40, 40, 12, 20, 3),     # Cp = 12
    (2, 24, 40, 32, 32, 3),     # Winograd F(3x3,2x2) / halo, one tile each way
    (2, 35, 18, 96, 64, 3),     # ragged patches, three channel chunks
    (1, 17, 50, 128, 64, 3),    # Cin tile pair
    (1, 16, 16, 256, 128, 3),   # deep-layer shape
    (2, 64, 64, 3, 32, 3),      # first conv (Cin 3 on the packed NHWC4 input): streaming kernel
    (2, 64, 64, 32, 2, 1),      # 1x1 head: streaming kernel
    (3, 7, 5, 16, 8, 1),        # 1x1, tile kernel
    (2, 37, 45, 64, 64, 3),     # three-piece Winograd weight gradient, 64 x 32 tile, ragged patches on both axes
    (2, 35, 18, 64, 32, 3),     # ... 32 x 64 tile (N = 32)
    (1, 20, 50, 32, 96, 3),     # ... 32 x 32 tile (N = 96 is not a multiple of 64)
]


@pytest.mark.parametrize("variant", ["default", "no_wino_wgrad", "tiles", "fp32_mfma", "wgrad_fp32_mfma"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", WGRAD_CASES)
def test_conv_wgrad_vs_float64(cuda, variant, B, H, W, Cin, Cout, k):
    x = torch.from_numpy(O.formula_normal("bk/x", (B, Cin, H, W), seed=Cin + H))
    dz = torch.from_numpy(O.formula_normal("bk/dz", (B, Cout, H, W), seed=Cout + W))
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dz.double(), padding=k // 2)
    Cp, N = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xin = torch.zeros((B, H, W, Cp), device=cuda)
    xin[..., :Cin] = nhwc(x).to(cuda)
    dzn = torch.zeros((B, H, W, N), device=cuda)
    dzn[..., :Cout] = nhwc(dz).to(cuda)
    dw = torch.full((Cout, Cin, k, k), float("nan"), device=cuda)
    with context(cuda, **VARIANTS[variant]) as ctx:
        _lib.check(_lib.lib().mgu_conv2d_wgrad_nhwc(ctx.handle, xin.data_ptr(), Cp, dzn.data_ptr(), B, H, W, Cin, Cout, k, dw.data_ptr(),
                                                    _lib.current_stream_ptr(cuda)), ctx.handle)
        e = rel(dw, ref)
    assert e <= TOL, e


@pytest.mark.parametrize("variant", ["default", "no_wino_wgrad", "fp32_mfma"])
@pytest.mark.parametrize("H,Cin,Cout", [(512, 32, 32), (256, 64, 64), (64, 256, 256)])
def test_conv_wgrad_c5_layer_shapes(cuda, variant, H, Cin, Cout):
    """BASELINE configs[4] shard (4 images): a 2^20-pixel reduction per weight at the top level."""
    B = 4
    g = torch.Generator().manual_seed(H)
    x = torch.randn((B, Cin, H, H), generator=g)
    dz = torch.randn((B, Cout, H, H), generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dz.double(), padding=1)
    xin, dzn = nhwc(x).to(cuda), nhwc(dz).to(cuda)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=cuda)
    with context(cuda, **VARIANTS[variant]) as ctx:
        _lib.check(_lib.lib().mgu_conv2d_wgrad_nhwc(ctx.handle, xin.data_ptr(), Cin, dzn.data_ptr(), B, H, H, Cin, Cout, 3, dw.data_ptr(),
                                                    _lib.current_stream_ptr(cuda)), ctx.handle)
        e = rel(dw, ref)
    assert e <= TOL, e


@pytest.mark.parametrize("variant", ["default", "fp32_mfma", "no_wino_dgrad"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 13, 17, 8, 32, 3), (1, 40, 33, 64, 32, 3), (2, 35, 18, 96, 48, 3), (1, 16, 16, 256, 128, 3), (1, 17, 50, 136, 256, 3),
    (2, 20, 24, 32, 2, 1), (4, 128, 128, 64, 64, 3)])
def test_conv_dgrad_vs_float64(cuda, variant, B, H, W, Cin, Cout, k):
    env = {"no_wino_dgrad": {"MGU_NO_WINO_DGRAD": 1}}.get(variant, VARIANTS.get(variant, {}))
    w = torch.from_numpy(O.formula_uniform("bk/w", (Cout, Cin, k, k), -0.2, 0.2, seed=Cout))
    dz = torch.from_numpy(O.formula_normal("bk/dz", (B, Cout, H, W), seed=Cout + W))
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dz.double(), padding=k // 2)
    N = (Cout + 3) // 4 * 4
    dzn = torch.zeros((B, H, W, N), device=cuda)
    dzn[..., :Cout] = nhwc(dz).to(cuda)
    wd = w.to(cuda)
    din = torch.full((B, H, W, Cin), float("nan"), device=cuda)
    with context(cuda, **env) as ctx:
        _lib.check(_lib.lib().mgu_conv2d_dgrad_nhwc(ctx.handle, dzn.data_ptr(), wd.data_ptr(), B, H, W, Cin, Cout, k, din.data_ptr(), Cin,
                                                    _lib.current_stream_ptr(cuda)), ctx.handle)
        e = rel(din.permute(0, 3, 1, 2), ref)
    assert e <= TOL, e


@pytest.mark.parametrize("variant", ["default", "convt_dgrad_tiles"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 9, 64, 32), (1, 16, 16, 512, 256), (2, 33, 20, 128, 64), (4, 256, 256, 64, 32), (1, 9, 7, 16, 32)])
def test_conv_transpose_backward_vs_float64(cuda, variant, B, H, W, Cin, Cout):
    """ConvTranspose2d(k2,s2) weight, bias and data gradients (unet_decoder.py:25,36) with the gradient arriving in the upper
    channel half of a concat-shaped buffer, as in mgu_unet_backward."""
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = (torch.rand((Cin, Cout, 2, 2), generator=g) - 0.5) * 0.4
    dout = torch.randn((B, Cout, 2 * H, 2 * W), generator=g)
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    bd = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(xd, wd, bd, stride=2).backward(dout.double())
    ld, coff = 2 * Cout, Cout
    dcat = torch.randn((B, 2 * H, 2 * W, ld), generator=g).to(cuda)     # lower half: unrelated data that must not be read
    dcat[..., coff:] = nhwc(dout).to(cuda)
    xin, wdev = nhwc(x).to(cuda), w.to(cuda)
    dw = torch.full((Cin, Cout, 2, 2), float("nan"), device=cuda)
    db = torch.full((Cout,), float("nan"), device=cuda)
    din = torch.full((B, H, W, Cin), float("nan"), device=cuda)
    with context(cuda, **VARIANTS[variant]) as ctx:
        L, s = _lib.lib(), _lib.current_stream_ptr(cuda)
        _lib.check(L.mgu_conv_transpose2x2_wgrad_nhwc(ctx.handle, xin.data_ptr(), dcat.data_ptr(), ld, coff, B, H, W, Cin, Cout, dw.data_ptr(),
                                                      db.data_ptr(), s), ctx.handle)
        _lib.check(L.mgu_conv_transpose2x2_dgrad_nhwc(ctx.handle, dcat.data_ptr(), ld, coff, wdev.data_ptr(), B, H, W, Cin, Cout, din.data_ptr(), s),
                   ctx.handle)
        e = (rel(dw, wd.grad), rel(db, bd.grad), rel(din.permute(0, 3, 1, 2), xd.grad))
    assert max(e) <= TOL, e


@pytest.mark.parametrize("M,C,ld", [(2 * 13 * 17, 8, 8), (4 * 64 * 64, 32, 64), (2 * 37 * 45, 256, 256), (4 * 512 * 512, 32, 32), (1000, 512, 512)])
def test_bn_relu_forward_backward_vs_float64(cuda, M, C, ld):
    """Train-mode BatchNorm2d + ReLU (unet_encoder.py:12-13,17-18): batch statistics, y, running stats, and the two backward
    reduction passes chan_reduce<1> / chan_reduce<3> (dgamma, dbeta, dz, conv-bias gradient)."""
    g = torch.Generator().manual_seed(C + 1)
    z = torch.randn((M, C), generator=g) * 1.7 + torch.linspace(-1, 1, C)
    gamma = torch.rand(C, generator=g) + 0.5
    gamma[::3] *= -1                                     # negative gammas flip the sign test of the recomputed ReLU mask
    beta = torch.rand(C, generator=g) - 0.5
    dy = torch.randn((M, C), generator=g)
    zd, gd, bd = z.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    yd = F.relu(F.batch_norm(zd, rm, rv, gd, bd, training=True, momentum=0.1, eps=1e-5))
    yd.backward(dy.double())
    zc, gc, bc = z.to(cuda), gamma.to(cuda), beta.to(cuda)
    y = torch.full((M, ld), float("nan"), device=cuda)
    mean, invstd = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    rmean, rvar = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    dyc = torch.zeros((M, ld), device=cuda)
    dyc[:, :C] = dy.to(cuda)
    dz = torch.full((M, C), float("nan"), device=cuda)
    dgam, dbet, dbias = (torch.full((C,), float("nan"), device=cuda) for _ in range(3))
    with context(cuda) as ctx:
        L, s = _lib.lib(), _lib.current_stream_ptr(cuda)
        _lib.check(L.mgu_bn_relu_train_nhwc(ctx.handle, zc.data_ptr(), gc.data_ptr(), bc.data_ptr(), M, C, y.data_ptr(), ld, mean.data_ptr(),
                                            invstd.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), s), ctx.handle)
        _lib.check(L.mgu_bn_relu_backward_nhwc(ctx.handle, dyc.data_ptr(), ld, zc.data_ptr(), gc.data_ptr(), bc.data_ptr(), mean.data_ptr(),
                                               invstd.data_ptr(), M, C, dz.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), dbias.data_ptr(), s),
                   ctx.handle)
        torch.cuda.synchronize()
    assert rel(y[:, :C], yd.detach()) <= TOL
    assert rel(mean, zd.detach().mean(0)) <= TOL and rel(rmean, rm) <= TOL and rel(rvar, rv) <= TOL
    e = (rel(dz, zd.grad), rel(dgam, gd.grad), rel(dbet, bd.grad))
    assert max(e) <= TOL, e
    # the conv bias in front of a BatchNorm has an analytically zero gradient: column sums of dz
    assert float(dbias.abs().max().cpu()) <= 1e-4 * float(zd.grad.abs().sum(0).max())


@pytest.mark.parametrize("B,H,W,C,ld", [(2, 8, 10, 8, 16), (1, 37, 45, 32, 64), (4, 256, 256, 64, 128)])
def test_maxpool_backward_vs_float64(cuda, B, H, W, C, ld):
    g = torch.Generator().manual_seed(H)
    y = torch.randn((B, C, H, W), generator=g)
    dpool = torch.randn((B, C, H // 2, W // 2), generator=g)
    base = torch.randn((B, C, H, W), generator=g)      # gradient already in dskip (decoder side): the kernel ACCUMULATES
    yd = y.double().requires_grad_()
    F.max_pool2d(yd, 2, 2).backward(dpool.double())
    ref = base.double() + yd.grad
    ybuf = torch.zeros((B, H, W, ld), device=cuda)
    ybuf[..., :C] = nhwc(y).to(cuda)
    dskip = torch.zeros((B, H, W, ld), device=cuda)
    dskip[..., :C] = nhwc(base).to(cuda)
    dp = nhwc(dpool).to(cuda)
    with context(cuda) as ctx:
        _lib.check(_lib.lib().mgu_maxpool2x2_backward_nhwc(ctx.handle, ybuf.data_ptr(), ld, dp.data_ptr(), dskip.data_ptr(), ld, B, H, W, C,
                                                           _lib.current_stream_ptr(cuda)), ctx.handle)
        got = dskip[..., :C].permute(0, 3, 1, 2)
        assert torch.equal(got.cpu().double(), ref.float().double()) or rel(got, ref) <= 1e-7
        assert float(dskip[..., C:].abs().max().cpu()) == 0.0


# ---- cross entropy: ignore_index and invalid labels ------------------------------------------------------------
def test_cross_entropy_ignore_index_and_label_range(cuda):
    npix, Cc = 5000, 3
    g = torch.Generator().manual_seed(3)
    logits = torch.randn((npix, Cc), generator=g)
    labels = torch.randint(0, Cc, (npix,), generator=g)
    labels[::7] = -100                                    # nn.CrossEntropyLoss's default ignore_index
    ld = logits.double().requires_grad_()
    loss_ref = F.cross_entropy(ld, labels)
    loss_ref.backward()
    lc, yc = logits.to(cuda), labels.to(cuda)
    dl = torch.full((npix, 4), float("nan"), device=cuda)
    loss = torch.zeros(1, device=cuda)
    with context(cuda) as ctx:
        L, s = _lib.lib(), _lib.current_stream_ptr(cuda)
        _lib.check(L.mgu_cross_entropy(ctx.handle, lc.data_ptr(), yc.data_ptr(), npix, Cc, 1.0 / npix, dl.data_ptr(), loss.data_ptr(), s), ctx.handle)
        _lib.check(L.mgu_sync_check(ctx.handle, s), ctx.handle)
        assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
        assert rel(dl[:, :Cc], ld.grad) <= TOL and float(dl[:, Cc:].abs().max()) == 0.0
        assert float(dl[::7].abs().max()) == 0.0          # ignored pixels: zero gradient
        # a raw 0/255 mask: 255 is neither a class nor the ignore_index -> the reference raises; here: NaN loss, zero gradient
        # for that pixel, no out-of-bounds read, and the error surfaces at the next check
        bad = labels.clone()
        bad[11] = 255
        bc = bad.to(cuda)
        _lib.check(L.mgu_cross_entropy(ctx.handle, lc.data_ptr(), bc.data_ptr(), npix, Cc, 1.0 / npix, dl.data_ptr(), loss.data_ptr(), s), ctx.handle)
        with pytest.raises(ValueError, match="label"):
            _lib.check(L.mgu_sync_check(ctx.handle, s), ctx.handle)
        assert bool(torch.isnan(loss).all()) and float(dl[11].abs().max()) == 0.0
        _lib.check(L.mgu_sync_check(ctx.handle, s), ctx.handle)   # reported once


# ---- the three-piece bf16 operand split of the fp32 Winograd convolutions (now the default) on adversarial operands ----------
def _conv_fwd(cuda, ctx, x, w):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xin, wd = nhwc(x).to(cuda), w.contiguous().to(cuda)
    out = torch.full((B, H, W, Cout), float("nan"), device=cuda)
    _lib.check(_lib.lib().mgu_conv2d_nhwc(ctx.handle, xin.data_ptr(), B, H, W, Cin, wd.data_ptr(), None, None, None, Cout, 3, 0,
                                          out.data_ptr(), Cout, 0, _lib.current_stream_ptr(cuda)), ctx.handle)
    return out.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("case", ["cancellation", "wide_exponents", "integers", "bf16_exact", "tiny_residuals"])
def test_three_piece_split_adversarial(cuda, case):
    """MGU_WINO_PREC=1 (default) must be an fp32 multiply in effect: measured against float64 its error may not exceed the
    exact-fp32 MFMA path's (MGU_WINO_PREC=0) by more than a rounding or two, on operands built to break a narrower product."""
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = 2, 64, 64, 24, 40
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * 0.1
    if case == "cancellation":
        # channel pairs (c, c + 32) carry +v and -v(1 + 2^-12) against equal weights: the k-sum cancels to ~2^-12 of its terms,
        # so an operand error of 2^-16 (two pieces only) would show up as a 6 % error of the result
        x[:, 32:] = -x[:, :32] * (1 + 2.0 ** -12)
        w[:, 32:] = w[:, :32]
    elif case == "wide_exponents":
        e = torch.linspace(-60, 60, Cin).round()
        x = x * (2.0 ** e).view(1, Cin, 1, 1)
        w = w * (2.0 ** -e).view(1, Cin, 1, 1)
    elif case == "integers":
        x = torch.randint(-3, 4, (B, Cin, H, W), generator=g).float()
        w = torch.randint(-2, 3, (Cout, Cin, 3, 3), generator=g).float()
    elif case == "bf16_exact":
        x = x.bfloat16().float()                            # second and third pieces are exactly zero
        w = w.bfloat16().float()
    elif case == "tiny_residuals":
        x = (x.bfloat16().float()) * (1 + 2.0 ** -23)       # mantissa 1...01: the third piece carries the last bit alone
    ref = F.conv2d(x.double(), w.double(), padding=1)
    scale = F.conv2d(x.double().abs(), w.double().abs(), padding=1)     # sum |a b|: the natural fp32 error scale
    with context(cuda, MGU_WINO_PREC=1) as c1:
        y1 = _conv_fwd(cuda, c1, x, w)
    with context(cuda, MGU_WINO_PREC=0) as c0:
        y0 = _conv_fwd(cuda, c0, x, w)
    e1 = float(((y1.double() - ref).abs() / scale).max())
    e0 = float(((y0.double() - ref).abs() / scale).max())
    if case == "integers":
        assert torch.equal(y1.double(), ref) and torch.equal(y0.double(), ref)
    # Winograd F(2x2,3x3) in fp32: a few 1e-7 of sum |a b| (the transforms add/subtract neighbours); the split may cost one
    # more rounding, never a different order of magnitude
    assert e0 <= 2e-6, (e0, e1)
    assert e1 <= max(2 * e0, 1e-6), (e0, e1)
