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


def test_e2e_forward_stages_1_to_7_vs_oracle_composition(cuda):
    """mgunet.MinGraphUNetE2E (train_end_to_end.py:270-453 on deterministic node features) against the oracle's stage
    functions chained the same way, 2 x 3 x 64 x 64: every tensor the loop produces."""
    B, H, W, K = 2, 64, 64, 2
    up = O.make_unet_params(3, 2, 32, 4, seed=1)
    gp_ = O.make_gat_params(32, 128, 64, 4, 1, seed=2)
    # predictor seed 15 with its W scaled by 64: the hard labels are MIXED (6 of the 32 patches in segment 1) and the closest
    # soft assignment is 0.022 from a tie (found with the oracle), so the stages behind the arg-max are always compared
    pp = {k: (v * 64 if k.endswith("W.weight") else v) for k, v in O.make_segment_predictor_params(64, K, 32, True, 2, seed=15).items()}
    rp = O.make_gat_params(64, 128, 64, 4, 1, seed=4)
    dp = O.make_detection_head_params(96, 1, 256, False, seed=5)
    x = torch.from_numpy(O.formula_normal("e2e/x", (B, 3, H, W), seed=6))

    unet = mgunet.UNet(3, 2, 32, 4); unet.load_state_dict(up)
    pgat = mgunet.GATNetwork(32, 128, 64, 4, 1); pgat.load_state_dict(gp_)
    pred = mgunet.PatchSegmentPredictor(64, K, hidden_dim=32, use_gnn=True, num_heads=2); pred.load_state_dict(pp)
    rgat = mgunet.GATNetwork(64, 128, 64, 4, 1); rgat.load_state_dict(rp)
    det = mgunet.DetectionHead(96, 1)
    sd = dict(dp); sd["conv_block.2.num_batches_tracked"] = sd["conv_block.5.num_batches_tracked"] = torch.tensor(0)
    det.load_state_dict(sd)
    model = mgunet.MinGraphUNetE2E(unet, pgat, pred, mgunet.MinCutRefinement(), rgat, det, num_segments=K).to(cuda).eval()
    out = model(x.to(cuda))

    nph, npw = O.patch_grid(H, W, 16)
    ei = torch.from_numpy(O.patch_graph_edges(H, W, 16))
    with torch.no_grad():
        lg, _, ft = O.unet_forward(up, x, 4)
        losses, fused_ref, embs, softs = [], [], [], []
        for b in range(B):
            X = O.patch_mean_features(ft[0][b], 16)
            emb = O.gat_network_forward(gp_, X, ei, 4)
            sl = O.segment_predictor_forward(pp, emb, ei, True, 2)
            loss, soft, hard = O.mincut_forward(emb, ei, K, sl)
            _, pix = O.region_stage(emb, hard, K, rp, 4, nph, npw, H, W)
            fused_ref.append(O.feature_fusion([ft[0][b:b + 1]], pix.unsqueeze(0)))
            losses.append(loss), embs.append(emb), softs.append(soft)
        fused_ref = torch.cat(fused_ref, 0)
        bb, cf = O.detection_head_forward(dp, fused_ref, 1)
    assert float((out["logits"].cpu() - lg).abs().max()) <= 1e-3
    assert float((out["node_embeddings"].cpu() - torch.cat(embs)).abs().max()) <= 1e-4
    assert float((out["soft_assignments"].cpu() - torch.cat(softs)).abs().max()) <= 1e-4
    assert abs(float(out["loss_partition"]) - float(torch.stack(losses).mean())) <= 1e-4
    # the hard labels decide which region embedding a pixel gets: this case is chosen so that no assignment is a near-tie
    margin = (torch.cat(softs)[:, 0] - torch.cat(softs)[:, 1]).abs().min()
    assert float(margin) > 1e-2
    hard_ref = torch.cat(softs).argmax(1)
    assert 0 < int(hard_ref.sum()) < hard_ref.numel()          # both segments are populated
    assert torch.equal(out["hard_labels"].cpu().reshape(-1).long(), hard_ref)
    assert float((out["fused"].cpu() - fused_ref).abs().max()) <= 1e-3
    assert float((out["bboxes"].cpu() - bb).abs().max()) <= 1e-4 and float((out["confidence"].cpu() - cf).abs().max()) <= 1e-4
    assert tuple(out["fused"].shape) == (B, 96, H, W) and tuple(out["bboxes"].shape) == (B, 4)
