"""DetectionHead (second half) on the HIP path -- mgu_conv2d_nhwc (Winograd / implicit GEMM),
mgu_channel_affine_nhwc, mgu_channel_sum_nhwc -- against the outputs of the reference class (tests/golden/dethead.npz).
Tolerance 2e-5 absolute on sigmoid outputs / O(1) class scores: two fp32 convolutions and a mean over up to 16 384
pixels in a different summation order."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O
from test_oracle_golden import DETHEAD_CASES, dethead_case

pytestmark = pytest.mark.gpu


def build(cuda, C, ncls, flat, params):
    m = mgunet.DetectionHead(C, ncls, fc_hidden_dim=256, input_is_flat=flat)
    sd = dict(params)
    for k in ("2", "5"):
        if f"conv_block.{k}.weight" in sd:
            sd[f"conv_block.{k}.num_batches_tracked"] = torch.tensor(0)
    res = m.load_state_dict(sd, strict=True)          # same keys as detection_head.py:31-66
    assert not res.missing_keys and not res.unexpected_keys
    return m.to(cuda).eval()


@pytest.mark.parametrize("tag", list(DETHEAD_CASES))
def test_detection_head_vs_reference_fixture(cuda, golden, tag):
    p, x, ncls, flat, C = dethead_case(tag)
    g = golden["dethead"]
    m = build(cuda, C, ncls, flat, p)
    out = m(x.to(cuda))
    assert len(out) == (3 if ncls > 1 else 2)
    for nm, t in zip(("bbox", "conf", "cls"), out):
        assert tuple(t.shape) == g[f"{tag}_{nm}"].shape
        assert np.abs(t.cpu().numpy() - g[f"{tag}_{nm}"]).max() <= 2e-5, nm


def test_detection_head_on_fused_features_and_errors(cuda):
    """The e2e order (train_end_to_end.py:433-453): region_stage's fused (B, 96, H, W) tensor straight into the head; one
    image's prediction does not depend on its batch neighbours; interface errors."""
    B, H, W = 3, 64, 96
    p = O.make_detection_head_params(96, 1, 256, False, seed=13)
    m = build(cuda, 96, 1, False, p)
    gen = torch.Generator(device=cuda).manual_seed(1)
    fused = torch.randn((B, H, W, 96), device=cuda, generator=gen).permute(0, 3, 1, 2)   # NHWC storage, NCHW view
    bb, cf = m(fused)
    assert tuple(bb.shape) == (B, 4) and tuple(cf.shape) == (B, 1)
    assert bool(((bb > 0) & (bb < 1)).all()) and bool(((cf > 0) & (cf < 1)).all())
    b1, c1 = m(fused[1:2])
    assert float((b1 - bb[1:2]).abs().max()) <= 1e-6 and float((c1 - cf[1:2]).abs().max()) <= 1e-6
    with torch.no_grad():
        ob, oc = O.detection_head_forward(p, fused.cpu().contiguous(), 1)
    assert float((bb.cpu() - ob).abs().max()) <= 2e-5 and float((cf.cpu() - oc).abs().max()) <= 2e-5
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(fused.cpu())
    with pytest.raises(RuntimeError, match="eval"):
        m.train()(fused)
