"""Region stage + feature fusion (first half) on the HIP path: mgu_region_mean_pool,
mgu_gat_layer_forward on the K-node region graphs, mgu_region_fuse_nhwc -- against the fixtures made with the
reference's GATNetwork / F.interpolate / FeatureFusion.  Pooling and gathers are exact data movement (the mean is one
fp32 sum in a different order: 1e-6); the region GAT carries the GAT tolerance (1e-5)."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O
from test_oracle_golden import REGION_CASES, region_case

pytestmark = pytest.mark.gpu


def make_region_gat(cuda, heads, p):
    g = mgunet.GATNetwork(64, 128, 64, heads, num_gat_layers=1)
    g.load_state_dict(p)
    return g.to(cuda).eval()


@pytest.mark.parametrize("tag", list(REGION_CASES))
def test_region_stage_vs_reference_fixture(cuda, golden, tag):
    g, H, W, K, heads, nph, npw, feats, hard, p, fu = region_case(golden, tag)
    gat = make_region_gat(cuda, heads, p)
    emb, fused = mgunet.region_stage(feats.to(cuda), hard.to(cuda), 1, K, gat, nph, npw, H, W, f_u=fu.to(cuda))
    assert tuple(emb.shape) == (K, 64) and tuple(fused.shape) == (1, 96, H, W)
    assert np.abs(emb.cpu().numpy() - g[f"{tag}_emb"]).max() <= 1e-5
    ff = fused.contiguous().cpu().reshape(-1).numpy()
    assert np.abs(ff[g[f"{tag}_fused_idx"]] - g[f"{tag}_fused"]).max() <= 1e-5
    assert torch.equal(fused[:, :32].cpu(), fu)                      # the U-Net half is a copy
    pix = fused[0, 32:].contiguous().cpu().reshape(-1).numpy()
    assert np.abs(pix[g[f"{tag}_pix_idx"]] - g[f"{tag}_pix"]).max() <= 1e-5
    # label-mean pooling alone against the oracle's loop
    reg = mgunet.region_mean_pool(feats.to(cuda), hard.to(cuda), 1, K).cpu()
    ref = torch.zeros(K, 64)
    for k in range(K):
        if int((hard == k).sum()):
            ref[k] = feats[hard == k].mean(0)
    assert float((reg - ref).abs().max()) <= 1e-6


def test_region_stage_batch_equals_per_image_and_feature_fusion_mirror(cuda):
    """A batch of three images through one block-diagonal region-GAT launch equals the three single-image results; and
    the FeatureFusion mirror (feature_fusion.py:43-162) gives the same tensor from the aligned / per-region inputs."""
    H, W, K, heads = 48, 80, 3, 4
    nph, npw = O.patch_grid(H, W, 16)
    Np = nph * npw
    p = O.make_gat_params(64, 128, 64, heads, 1, seed=9)
    gat = make_region_gat(cuda, heads, p)
    feats = torch.from_numpy(O.formula_normal("region/batch/x", (3 * Np, 64), seed=5)).to(cuda) * 0.5
    hard = torch.from_numpy(O.formula_labels("region/batch/y", (3 * Np,), K, seed=6)).to(cuda)
    fu = torch.from_numpy(O.formula_normal("region/batch/fu", (3, 32, H, W), seed=7)).to(cuda)
    emb, fused = mgunet.region_stage(feats, hard, 3, K, gat, nph, npw, H, W, f_u=fu)
    for b in range(3):
        e1, f1 = mgunet.region_stage(feats[b * Np:(b + 1) * Np], hard[b * Np:(b + 1) * Np], 1, K, gat, nph, npw, H, W, f_u=fu[b:b + 1])
        assert float((emb[b * K:(b + 1) * K] - e1).abs().max()) <= 1e-6 and torch.equal(fused[b], f1[0])
        # the oracle on the same image
        oe, op = O.region_stage(feats[b * Np:(b + 1) * Np].cpu(), hard[b * Np:(b + 1) * Np].cpu(), K, p, heads, nph, npw, H, W)
        assert float((e1.cpu() - oe).abs().max()) <= 1e-5 and float((f1[0, 32:].cpu() - op).abs().max()) <= 1e-5
    fuser = mgunet.FeatureFusion(unet_feature_dims=[32], gat_feature_dim=64)
    f_g = fused[:, 32:]
    assert torch.equal(fuser([fu], f_g, target_spatial_size=(H, W)), fused)       # aligned (B, D, H, W) F_g
    # per-region F_g + pixel map of global region indices (feature_fusion.py:83-138), one pixel marked invalid
    py = torch.clamp((torch.arange(H, device=cuda).float() * (nph / H)).floor().long(), max=nph - 1)
    px = torch.clamp((torch.arange(W, device=cuda).float() * (npw / W)).floor().long(), max=npw - 1)
    rmap = torch.stack([hard[b * Np:(b + 1) * Np].long().reshape(nph, npw)[py][:, px] + b * K for b in range(3)])
    rmap[1, 3, 4] = -1
    got = fuser([fu], emb, target_spatial_size=(H, W), region_to_pixel_map=rmap)
    want = fused.clone()
    want[1, 32:, 3, 4] = 0.0
    assert torch.equal(got, want)
    with pytest.raises(ValueError, match="unsupported shape"):
        fuser([fu], emb)                                                           # 2-D F_g without a map (:144-146)
    with pytest.raises(ValueError, match="Channel dimensions must match"):
        mgunet.FeatureFusion([32], 64, fusion_method="add")([fu], f_g)            # :153-154
    with pytest.raises(NotImplementedError):
        mgunet.FeatureFusion([32], 64, fusion_method="multiply")([fu], f_g)      # :157


def test_region_fuse_headline_size_properties(cuda):
    """BASELINE configs[1] size (8 x 512 x 512, 32 + 64 channels): every 16 x 16 pixel block of the F_g half is constant
    and equals the embedding row of its patch's label; the F_u half is untouched."""
    B, H, W, K = 8, 512, 512, 2
    nph, npw = 32, 32
    gen = torch.Generator(device=cuda).manual_seed(3)
    emb = torch.randn((B * K, 64), device=cuda, generator=gen)
    hard = torch.randint(0, K, (B * nph * npw,), device=cuda, generator=gen)
    fu = torch.randn((B, H, W, 32), device=cuda, generator=gen).permute(0, 3, 1, 2)   # NHWC storage like mgunet's features
    fused = mgunet.region_fuse(fu, emb, hard, B, H, W, nph, npw, K)
    assert torch.equal(fused[:, :32], fu)
    blocks = fused[:, 32:].reshape(B, 64, nph, 16, npw, 16)
    assert torch.equal(blocks, blocks[:, :, :, :1, :, :1].expand_as(blocks))
    want = emb.reshape(B, K, 64)[torch.arange(B, device=cuda)[:, None], hard.reshape(B, -1).long()]   # (B, Np, 64)
    assert torch.equal(blocks[:, :, :, 0, :, 0].permute(0, 2, 3, 1).reshape(B, -1, 64), want)
