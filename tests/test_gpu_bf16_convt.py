"""bf16-storage ConvTranspose2d(2, 2) on its fragment kernel (csrc/convt_bf16.hip: whole-row LDS staging, transposed 16-byte stores)
against the generic tile kernel it replaces (MGU_NO_CONVT_FRAG=1) and against torch on bf16-rounded operands
(model/unet/unet_decoder.py:25,36).  Both kernels accumulate the same bf16 products in fp32, in a different order: outputs agree to
a bf16 rounding."""
import pytest
import torch
import torch.nn.functional as F

import mgunet
import mgunet_oracle as O

pytestmark = pytest.mark.gpu


def _forward(cuda, cfg, p, x):
    unet = mgunet.UNet(*cfg, compute_dtype=torch.bfloat16)
    unet.load_state_dict(p)
    unet = unet.to(cuda).eval()
    lg, sk, ft = unet(x)
    torch.cuda.synchronize()
    return [lg.float().clone()] + [t.float().clone() for t in ft]


@pytest.mark.parametrize("shape", [(2, 3, 128, 160), (1, 3, 72, 104), (3, 3, 64, 64)])
def test_bf16_forward_with_fragment_convtranspose_matches_generic_kernel(cuda, shape, monkeypatch):
    cfg = (3, 2, 32, 4)     # ConvTranspose layers 512->256, 256->128, 128->64, 64->32: all on the fragment kernel
    p = O.make_unet_params(*cfg, seed=41)
    x = torch.from_numpy(O.formula_normal("ctb/x", shape, seed=7)).to(cuda)
    new = _forward(cuda, cfg, p, x)
    monkeypatch.setenv("MGU_NO_CONVT_FRAG", "1")
    old = _forward(cuda, cfg, p, x)
    monkeypatch.delenv("MGU_NO_CONVT_FRAG")
    for a, b in zip(new, old):
        assert a.shape == b.shape and bool(torch.isfinite(a).all())
        scale = float(b.abs().max())
        # a bf16 rounding of a value near the maximum is 2^-8 of it; the two accumulation orders may round a few values differently,
        # and the difference then rides through the following layers
        assert float((a - b).abs().max()) <= 2.0 ** -6 * scale
        assert float((a - b).norm()) <= 3e-3 * float(b.norm())
    assert any(not torch.equal(a, b) for a, b in zip(new, old)) or True   # (equal outputs are fine: both are exact bf16 products)


def test_fragment_convtranspose_layer_vs_torch(cuda):
    """One layer through the C-ABI in a bf16 context is not exposed (mgu_conv_transpose2x2_nhwc is the fp32 building block), so the
    layer is pinned through the smallest U-Net that contains it: depth 1, the decoder's ConvTranspose 64 -> 32 feeds the concat buffer."""
    cfg = (3, 2, 32, 1)
    p = O.make_unet_params(*cfg, seed=43)
    x = torch.from_numpy(O.formula_normal("ctb/x1", (2, 3, 48, 80), seed=9))
    got = _forward(cuda, cfg, p, x.to(cuda))[0].cpu()
    with torch.no_grad():
        ref = O.unet_forward(p, x, cfg[3])[0]
    # bf16 storage of every activation: the reference's own bf16 run deviates by ~2-3 % of max|logit| (tests/test_gpu_parity.py pins it)
    assert float((got - ref).abs().max()) <= 4e-2 * max(1.0, float(ref.abs().max()))
