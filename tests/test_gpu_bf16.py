"""BASELINE configs[2] precision mode: bf16 storage + fp32 accumulate (inference).  The reference has no
bf16 path of its own for this comparison (its GAT is fp32-only), so the yardstick is the fp32
golden data with a bf16 tolerance: SURVEY 8d reports that the reference's OWN bf16 CPU run deviates 1.4 % of
max|logit| from its fp32 run with 99.57 % argmax agreement.  Stated tolerance here: max-abs <= 3 % of max|logit|,
mean-abs <= 0.5 % of max|logit|, argmax agreement >= 99 %."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O

pytestmark = pytest.mark.gpu


def build(cfg, seed, dev, dtype):
    m = mgunet.UNet(*cfg, compute_dtype=dtype)
    m.load_state_dict(O.make_unet_params(*cfg, seed=seed))
    return m.to(dev).eval()


def check_bf16(lg, ref, tag):
    lg, ref = lg.float().cpu().numpy(), np.asarray(ref)
    scale = np.abs(ref).max()
    d = np.abs(lg - ref)
    agree = (lg.argmax(1) == ref.argmax(1)).mean() if ref.ndim == 4 else None
    print(f"[bf16 {tag}] max-abs {d.max():.4f} ({d.max()/scale*100:.2f} % of max|logit| {scale:.2f}), mean-abs {d.mean():.5f}"
          + (f", argmax agreement {agree*100:.2f} %" if agree is not None else ""))
    assert d.max() <= 3e-2 * scale and d.mean() <= 5e-3 * scale
    if agree is not None:
        assert agree >= 0.99


@pytest.mark.parametrize("tag,cfg,shape", [("a", (1, 2, 8, 2), (1, 1, 32, 32)), ("b", (3, 3, 8, 2), (2, 3, 37, 45)),
                                           ("c", (3, 2, 8, 3), (2, 3, 64, 48))])
def test_bf16_tiny_vs_fp32_golden(cuda, golden, tag, cfg, shape):
    g = golden["unet_tiny"]
    m = build(cfg, 11, cuda, torch.bfloat16)
    x = torch.from_numpy(O.formula_normal(f"tiny/{tag}/x", shape, seed=11)).to(cuda)
    lg, sk, ft = m(x)
    assert lg.dtype == torch.float32 and sk[0].dtype == torch.bfloat16 and ft[0].dtype == torch.bfloat16
    check_bf16(lg, g[f"{tag}_logits"], tag)
    for i in range(cfg[3]):
        ref = g[f"{tag}_skip{i}"]
        assert float(np.abs(sk[i].float().cpu().numpy() - ref).max()) <= 3e-2 * np.abs(ref).max()
        ref = g[f"{tag}_feat{i}"]
        assert float(np.abs(ft[i].float().cpu().numpy() - ref).max()) <= 4e-2 * np.abs(ref).max()


def test_bf16_config3_shard_vs_fp32_path(cuda, golden):
    """8 images of 3x512x512 (one GPU's shard of configs[2]'s batch of 64) in bf16 vs the fp32 golden samples and vs
    the fp32 HIP path (full tensors: argmax agreement), plus the full forward with the fp32 GAT on bf16 features."""
    g = golden["c2"]
    cfg = (3, 2, 32, 4)
    mb = build(cfg, 0, cuda, torch.bfloat16)
    mf = build(cfg, 0, cuda, torch.float32)
    x = torch.cat([torch.from_numpy(O.formula_normal(f"c2/x/{b}", (1, 3, 512, 512), seed=1)) for b in range(8)]).to(cuda)
    lb = mb(x)[0]
    lf = mf(x)[0]
    check_bf16(lb, lf.cpu().numpy(), "c2 full tensors vs fp32 HIP")
    worst = 0.0
    for b in range(8):
        got = lb[b].contiguous().reshape(-1)[torch.from_numpy(g[f"idx_{b}"]).to(cuda)].cpu().numpy()
        worst = max(worst, float(np.abs(got - g[f"logits_{b}"]).max()))
    assert worst <= 3e-2 * 8.0
    gat = mgunet.GATNetwork(32, 128, 64, 4, 1)
    gat.load_state_dict(O.make_gat_params(32, 128, 64, 4, 1, seed=0))
    model = mgunet.MinGraphUNet(mb, gat.to(cuda).eval(), 16).eval()
    _, _, _, emb = model(x)
    ref = np.concatenate([g[f"gat_{b}"] for b in range(8)])
    got = np.concatenate([emb[b * 1024:(b + 1) * 1024].reshape(-1)[torch.from_numpy(g[f"gidx_{b}"]).to(cuda)].cpu().numpy()
                          for b in range(8)])
    print(f"[bf16 c2] GAT embedding max-abs {np.abs(got - ref).max():.4f} at max|emb| {np.abs(ref).max():.2f}")
    assert np.abs(got - ref).max() <= 5e-2 * np.abs(ref).max()
    assert torch.equal(mb(x[3:4])[0][0], lb[3])   # batch independence also holds in bf16


def test_bf16_is_inference_only_and_checked(cuda):
    m = build((3, 2, 8, 2), 5, cuda, torch.bfloat16)
    with pytest.raises(RuntimeError, match="inference"):
        m.train()(torch.zeros(1, 3, 32, 32, device=cuda))
    with pytest.raises(ValueError):
        mgunet.UNet(3, 2, 12, 2, compute_dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        mgunet.UNet(3, 2, 8, 2, compute_dtype=torch.float16)
