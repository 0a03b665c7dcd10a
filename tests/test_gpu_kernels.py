"""Kernel-level parity through the building-block C-ABI entry points (mgu_conv2d_nhwc,
mgu_conv_transpose2x2_nhwc, mgu_maxpool2x2_nhwc) against plain torch fp32 CPU ops.  These isolate the
implicit-GEMM tile configurations (N <= 32, <= 64, > 64), the K tail, the M tail, image borders and the
channel-slice (concat) stores."""
import ctypes as C

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mgunet_oracle as O
from mgunet import _lib
from mgunet.gat import _context

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def conv_gpu(cuda, x, w, b, k, relu, ld_out=None, c_off=0, scale=None, shift=None):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    ld_out = ld_out or Cout
    xin, wd, bd = nhwc(x).to(cuda), w.contiguous().to(cuda), b.to(cuda)
    out = torch.full((B, H, W, ld_out), -7.0, device=cuda)
    sc = scale.to(cuda) if scale is not None else None
    sh = shift.to(cuda) if shift is not None else None
    ctx = _context(cuda)
    rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, xin.data_ptr(), B, H, W, Cin, wd.data_ptr(), bd.data_ptr(),
                                    sc.data_ptr() if sc is not None else None, sh.data_ptr() if sh is not None else None,
                                    Cout, k, relu, out.data_ptr(), ld_out, c_off, _lib.current_stream_ptr(cuda))
    _lib.check(rc, ctx.handle)
    return out.cpu()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (1, 8, 8, 4, 8, 3),        # K=36: K tail inside a 32-step, tiny M
    (2, 13, 17, 8, 32, 3),     # odd sizes, N == 32 tile
    (1, 20, 24, 32, 64, 3),    # N == 64 tile
    (1, 9, 31, 64, 160, 3),    # N > 128: two N tiles + N tail, M tail
    (1, 16, 16, 128, 128, 3),  # tap-uniform K steps
    (3, 7, 5, 16, 2, 1),       # 1x1 head: N = 2
    (1, 33, 9, 36, 40, 1),     # 1x1, K = 36
    (1, 40, 40, 12, 20, 3),    # Cp = 12: taps straddle K steps
    (2, 13, 17, 32, 24, 3),    # halo kernel, N <= 32 tile, ragged 16x16 patches
    (1, 40, 33, 64, 32, 3),    # halo kernel, two channel chunks (halo prefetch), N == 32
    (2, 35, 18, 96, 48, 3),    # halo kernel, three chunks, N <= 64 tile
    (1, 17, 50, 256, 136, 3),  # halo kernel, 8x16 patches, eight chunks, N tail
])
def test_conv2d_vs_torch(cuda, B, H, W, Cin, Cout, k):
    x = torch.from_numpy(O.formula_normal("kc/x", (B, Cin, H, W), seed=Cin))
    w = torch.from_numpy(O.formula_uniform("kc/w", (Cout, Cin, k, k), -0.2, 0.2, seed=Cout))
    b = torch.from_numpy(O.formula_uniform("kc/b", (Cout,), -0.5, 0.5, seed=3))
    ref = F.conv2d(x, w, b, padding=k // 2)
    got = conv_gpu(cuda, x, w, b, k, 0)
    assert float((got.permute(0, 3, 1, 2) - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    got = conv_gpu(cuda, x, w, b, k, 1)
    assert float((got.permute(0, 3, 1, 2) - F.relu(ref)).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_conv2d_exact_integer_data_asymmetric(cuda):
    """Small-integer operands: every product and partial sum is exact in fp32, so the MFMA tile
    mapping (row/col, k order, tap order) must reproduce torch bit for bit.  Asymmetric weights catch
    transposed fragments; a delta input catches tap flips."""
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.integers(-3, 4, size=(2, 8, 11, 14)).astype(np.float32))
    w = torch.from_numpy(rng.integers(-2, 3, size=(40, 8, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.integers(-5, 6, size=(40,)).astype(np.float32))
    ref = F.conv2d(x, w, b, padding=1)
    got = conv_gpu(cuda, x, w, b, 3, 0).permute(0, 3, 1, 2)
    assert torch.equal(got, ref)
    x = torch.from_numpy(rng.integers(-3, 4, size=(2, 64, 19, 21)).astype(np.float32))   # halo kernel path
    w = torch.from_numpy(rng.integers(-2, 3, size=(40, 64, 3, 3)).astype(np.float32))
    assert torch.equal(conv_gpu(cuda, x, w, b, 3, 0).permute(0, 3, 1, 2), F.conv2d(x, w, b, padding=1))
    d = torch.zeros(1, 4, 9, 9)
    d[0, 1, 4, 4] = 1.0
    w2 = torch.arange(4 * 4 * 9, dtype=torch.float32).reshape(4, 4, 3, 3)
    assert torch.equal(conv_gpu(cuda, d, w2, torch.zeros(4), 3, 0).permute(0, 3, 1, 2), F.conv2d(d, w2, None, padding=1))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 5, 7, 16, 32),      # smaller than one 8x32 patch; N == 32 work split; one 16-channel chunk
    (2, 12, 10, 64, 64),    # two ragged patch rows, 64-channel work split, four chunks
    (1, 9, 70, 32, 96),     # three patch columns (last ragged), N tail inside a 64-channel n block
    (3, 16, 32, 48, 40),    # exact patches, Cin = 3 chunks, N tail of the second n tile
    (1, 31, 33, 128, 160),  # odd sizes both ways, three n blocks
    (4, 256, 256, 48, 32),  # four patches per workgroup with an ODD chunk count: the consumed raw buffer (= exchange region 0) alternates
                            # between the add-TID-reachable slot and the top slot of the LDS map (ordinary stores) from patch to patch
    (4, 256, 128, 80, 64),  # the same on the 64-channel work split (five chunks, two patches per workgroup)
])
def test_winograd_conv_vs_torch_and_direct(cuda, B, H, W, Cin, Cout, monkeypatch):
    """The fp32 3x3 layers with Cin % 16 == 0 route to wino3x3_f32_kernel (Winograd F(2x2,3x3), csrc/wino_f32.hip):
    compare against torch fp32 AND against the direct implicit-GEMM kernel (MGU_NO_WINOGRAD=1, read at mgu_create)
    on the same operands, including the scale/shift/ReLU epilogue into a channel slice of a wider buffer."""
    if os.environ.get("MGU_NO_WINOGRAD") or os.environ.get("MGU_WINO_PREC"):
        pytest.skip("an A/B of these switches: meaningless when one of them is forced for the whole run")
    x = torch.from_numpy(O.formula_normal("kw/x", (B, Cin, H, W), seed=H))
    w = torch.from_numpy(O.formula_uniform("kw/w", (Cout, Cin, 3, 3), -0.2, 0.2, seed=W))
    b = torch.zeros(Cout)
    sc = torch.from_numpy(O.formula_uniform("kw/sc", (Cout,), 0.5, 1.5, seed=1))
    sh = torch.from_numpy(O.formula_uniform("kw/sh", (Cout,), -0.5, 0.5, seed=2))
    ref = F.relu(F.conv2d(x, w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ld, off = Cout + 24, 8
    tol = 2e-5 * max(1.0, float(ref.abs().max()))

    def run():
        from mgunet import gat as G
        G._CTX.clear()   # new mgu_ctx: the environment switch is read by mgu_create
        got = conv_gpu(cuda, x, w, b, 3, 1, ld_out=ld, c_off=off, scale=sc, shift=sh)
        assert torch.all(got[..., :off] == -7.0) and torch.all(got[..., off + Cout:] == -7.0)
        return got[..., off:off + Cout].permute(0, 3, 1, 2)

    wino = run()
    monkeypatch.setenv("MGU_NO_WINOGRAD", "1")
    direct = run()
    monkeypatch.delenv("MGU_NO_WINOGRAD")
    from mgunet import gat as G
    G._CTX.clear()
    _context(cuda)   # mgu_create re-reads the environment: Winograd back on for the rest of the process
    assert float((wino - ref).abs().max()) <= tol
    assert float((direct - ref).abs().max()) <= tol
    assert float((wino - direct).abs().max()) <= tol
    assert not torch.equal(wino, direct)   # the two paths really are different kernels


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (2, 12, 10, 64, 64),    # 64-channel work split, four chunks
    (1, 9, 70, 32, 96),     # ragged patches, N tail
    (1, 31, 33, 128, 32),   # the 32-channel work split
    (1, 8, 32, 48, 40),     # odd chunk count
])
def test_winograd_three_piece_vs_fp32_mfma_operands(cuda, B, H, W, Cin, Cout, monkeypatch):
    """The default (MGU_WINO_PREC=1, read at mgu_create): every fp32 operand of the 16 Winograd GEMMs is split exactly into
    three bf16 pieces and multiplied on the bf16 MFMA with fp32 accumulation (csrc/wino_f32.hip PREC 1), against the same
    kernel on v_mfma_f32_32x32x2_f32 operands (MGU_WINO_PREC=0).  Same tolerance for both: the six kept piece products lose
    less than one fp32 rounding.  (Adversarial operands: test_gpu_backward_kernels.py::test_three_piece_split_adversarial.)"""
    if os.environ.get("MGU_NO_WINOGRAD") or os.environ.get("MGU_WINO_PREC"):
        pytest.skip("an A/B of these switches: meaningless when one of them is forced for the whole run")
    x = torch.from_numpy(O.formula_normal("k3/x", (B, Cin, H, W), seed=H))
    w = torch.from_numpy(O.formula_uniform("k3/w", (Cout, Cin, 3, 3), -0.2, 0.2, seed=W))
    b = torch.zeros(Cout)
    sc = torch.from_numpy(O.formula_uniform("k3/sc", (Cout,), 0.5, 1.5, seed=1))
    sh = torch.from_numpy(O.formula_uniform("k3/sh", (Cout,), -0.5, 0.5, seed=2))
    ref = F.relu(F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    from mgunet import gat as G

    def run():
        G._CTX.clear()   # new mgu_ctx: the switch is read by mgu_create and lives in the context
        got = conv_gpu(cuda, x, w, b, 3, 1, ld_out=Cout + 8, c_off=4, scale=sc, shift=sh)
        return got[..., 4:4 + Cout].permute(0, 3, 1, 2).double()

    split = run()
    monkeypatch.setenv("MGU_WINO_PREC", "0")
    base = run()
    monkeypatch.delenv("MGU_WINO_PREC")
    G._CTX.clear()
    _context(cuda)   # back to the default for the rest of the process
    e_base, e_split = float((base - ref).abs().max()), float((split - ref).abs().max())
    assert e_base <= tol and e_split <= tol, (e_base, e_split)
    assert e_split <= 2.0 * e_base + 1e-6, (e_base, e_split)   # as accurate as the fp32 MFMA path
    assert not torch.equal(base, split)                           # really a different kernel


def test_conv2d_scale_shift_relu_and_channel_slice_store(cuda):
    x = torch.from_numpy(O.formula_normal("ks/x", (1, 16, 12, 10), seed=1))
    w = torch.from_numpy(O.formula_uniform("ks/w", (24, 16, 3, 3), -0.2, 0.2, seed=1))
    b = torch.zeros(24)
    sc = torch.from_numpy(O.formula_uniform("ks/sc", (24,), 0.5, 1.5, seed=1))
    sh = torch.from_numpy(O.formula_uniform("ks/sh", (24,), -0.5, 0.5, seed=1))
    ref = F.relu(F.conv2d(x, w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got = conv_gpu(cuda, x, w, b, 3, 1, ld_out=64, c_off=32, scale=sc, shift=sh)
    assert float((got[..., 32:56].permute(0, 3, 1, 2) - ref).abs().max()) <= 2e-5
    assert torch.all(got[..., :32] == -7.0) and torch.all(got[..., 56:] == -7.0)  # neighbours untouched


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 4, 4, 8, 4), (2, 5, 7, 32, 16), (1, 8, 8, 64, 32), (1, 3, 9, 256, 128),
    # the three-piece bf16-MFMA kernel (convt_x3.hip: Cin % 16 == 0, Cout % 32 == 0): odd K-step count, pixel tiles that cross
    # image boundaries and end ragged, several n tiles, the deepest decoder shape
    (2, 16, 24, 48, 64), (3, 11, 13, 128, 96), (2, 32, 32, 512, 256), (1, 128, 128, 64, 32)])
def test_conv_transpose2x2_vs_torch(cuda, B, H, W, Cin, Cout):
    x = torch.from_numpy(O.formula_normal("kt/x", (B, Cin, H, W), seed=Cin))
    w = torch.from_numpy(O.formula_uniform("kt/w", (Cin, Cout, 2, 2), -0.2, 0.2, seed=Cout))
    b = torch.from_numpy(O.formula_uniform("kt/b", (Cout,), -0.5, 0.5, seed=3))
    ref = F.conv_transpose2d(x, w, b, stride=2)
    ld, off = 2 * Cout, Cout
    out = torch.full((B, 2 * H, 2 * W, ld), -7.0, device=cuda)
    xin, wd, bd = nhwc(x).to(cuda), w.contiguous().to(cuda), b.to(cuda)
    ctx = _context(cuda)
    rc = _lib.lib().mgu_conv_transpose2x2_nhwc(ctx.handle, xin.data_ptr(), B, H, W, Cin, wd.data_ptr(), bd.data_ptr(), Cout,
                                               out.data_ptr(), ld, off, _lib.current_stream_ptr(cuda))
    _lib.check(rc, ctx.handle)
    out = out.cpu()
    assert float((out[..., off:].permute(0, 3, 1, 2) - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.all(out[..., :off] == -7.0)


@pytest.mark.parametrize("B,H,W,Cc,ld", [(1, 8, 8, 4, 4), (2, 9, 13, 8, 16), (1, 6, 4, 32, 64)])
def test_maxpool_vs_torch(cuda, B, H, W, Cc, ld):
    x = torch.from_numpy(O.formula_normal("kp/x", (B, ld, H, W), seed=H))
    xin = nhwc(x).to(cuda)
    out = torch.empty((B, H // 2, W // 2, Cc), device=cuda)
    ctx = _context(cuda)
    rc = _lib.lib().mgu_maxpool2x2_nhwc(ctx.handle, xin.data_ptr(), ld, B, H, W, Cc, out.data_ptr(), _lib.current_stream_ptr(cuda))
    _lib.check(rc, ctx.handle)
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), F.max_pool2d(x[:, :Cc], 2, 2))


def test_invalid_arguments_return_errors_not_faults(cuda):
    ctx = _context(cuda)
    t = torch.zeros(64, device=cuda)
    L = _lib.lib()
    assert L.mgu_conv2d_nhwc(ctx.handle, t.data_ptr(), 1, 4, 4, 3, t.data_ptr(), None, None, None, 4, 3, 0, t.data_ptr(), 4, 0, None) == _lib.MGU_ERR_INVALID
    assert L.mgu_conv2d_nhwc(ctx.handle, t.data_ptr(), 1, 4, 4, 4, t.data_ptr(), None, None, None, 4, 5, 0, t.data_ptr(), 4, 0, None) == _lib.MGU_ERR_INVALID
    assert L.mgu_conv2d_nhwc(ctx.handle, t.data_ptr(), 1, 4, 4, 4, t.data_ptr(), None, None, None, 4, 3, 0, t.data_ptr(), 2, 0, None) == _lib.MGU_ERR_INVALID
    assert b"ld_out" in L.mgu_last_error(ctx.handle)
