"""The oracle restatement (oracle/mgunet_oracle.py) against the fixtures generated FROM THE REFERENCE
(oracle/make_golden.py).  CPU only.  This is what pins the oracle on boxes without /root/reference."""
import numpy as np
import pytest
import torch

import mgunet_oracle as O


def sample_idx(name, numel, n):
    u = O.formula_uniform(name, (n,), 0.0, 1.0, 7).astype(np.float64)
    return np.minimum((u * numel).astype(np.int64), numel - 1)


@pytest.mark.parametrize("tag,cfg,shape", [("a", (1, 2, 8, 2), (1, 1, 32, 32)), ("b", (3, 3, 8, 2), (2, 3, 37, 45)),
                                           ("c", (3, 2, 8, 3), (2, 3, 64, 48))])
def test_unet_tiny_eval_and_train(golden, tag, cfg, shape):
    g = golden["unet_tiny"]
    p = O.make_unet_params(*cfg, seed=11)
    x = torch.from_numpy(O.formula_normal(f"tiny/{tag}/x", shape, seed=11))
    with torch.no_grad():
        lg, sk, ft = O.unet_forward(p, x, cfg[3])
        stats = {}
        lgt, _, _ = O.unet_forward(p, x, cfg[3], training=True, new_stats=stats)
    assert np.abs(lg.numpy() - g[f"{tag}_logits"]).max() <= 1e-5
    for i in range(cfg[3]):
        assert np.abs(sk[i].numpy() - g[f"{tag}_skip{i}"]).max() <= 1e-5
        assert np.abs(ft[i].numpy() - g[f"{tag}_feat{i}"]).max() <= 1e-5
    assert np.abs(lgt.numpy() - g[f"{tag}_train_logits"]).max() <= 2e-5
    assert np.abs(stats["encoder.encoder_blocks.0.bn1.running_mean"].numpy() - g[f"{tag}_bn_rm_first"]).max() <= 1e-6
    assert np.abs(stats["encoder.encoder_blocks.0.bn1.running_var"].numpy() - g[f"{tag}_bn_rv_first"]).max() <= 1e-6


def test_unet_param_count_and_keys():
    shapes = O.unet_param_shapes(3, 2, 32, 4)
    n = sum(int(np.prod(s)) for k, s in shapes.items() if "running_" not in k and "num_batches" not in k)
    assert n == 7766018  # SURVEY section 4, measured on the reference


def test_gat_small(golden):
    g = golden["gat_small"]
    e10 = torch.from_numpy(g["edge10"])
    X10 = torch.from_numpy(O.formula_normal("gat/x10", (10, 32), seed=3))
    cases = [("g10", (32, 64, 16, 4), X10, e10, 1.0, 1),
             ("iso", (32, 64, 16, 4), torch.from_numpy(O.formula_normal("gat/x12", (12, 32), seed=3)),
              torch.from_numpy(g["edge_iso"]), 1.0, 1),
             ("wide", (32, 64, 16, 2), torch.from_numpy(O.formula_normal("gat/xw", (10, 32), seed=4)) * 4.0, e10, 3.0, 1),
             ("mid", (32, 64, 16, 4), torch.from_numpy(O.formula_normal("gat/xm", (10, 32), seed=6)) * 2.0, e10, 1.5, 1),
             ("l2h1", (32, 24, 8, 1), X10, e10, 1.0, 2)]
    for tag, cfg, X, ei, scale, layers in cases:
        p = O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=3, scale=scale)
        with torch.no_grad():
            y = O.gat_network_forward(p, X, ei, cfg[3], layers)
        assert np.abs(y.numpy() - g[tag + "_out"]).max() <= 1e-5, tag
    assert np.all(g["iso_out"][10:] == 0.0)  # targets without in-edges are exactly zero


@pytest.mark.parametrize("tag,H,W,p,nodes,edges", [("g128", 128, 128, 32, 16, 48), ("g130", 130, 140, 32, 25, 80),
                                                    ("g512", 512, 512, 16, 1024, 3968),
                                                    ("g1024", 1024, 1024, 16, 4096, 16128), ("g1", 16, 16, 16, 1, 0),
                                                    ("grow", 16, 80, 16, 5, 8)])
def test_patch_graph_bit_exact(golden, tag, H, W, p, nodes, edges):
    ei = O.patch_graph_edges(H, W, p)
    ref = golden["patch_graph"][tag]
    assert ei.dtype == np.int64 and ei.shape == (2, edges) and ref.shape == (2, edges)
    assert np.array_equal(ei, ref)
    nph, npw = O.patch_grid(H, W, p)
    assert nph * npw == nodes
    if tag == "g512":  # SURVEY section 4 known answer
        assert ei[:, :8].tolist() == [[0, 1, 0, 32, 1, 2, 1, 33], [1, 0, 32, 0, 2, 1, 33, 1]]


def test_patch_mean_non_divisible(golden):
    img = torch.from_numpy(O.formula_normal("graph/img", (5, 37, 45), seed=2))
    got = O.patch_mean_features(img, 16).numpy()
    assert np.abs(got - golden["patch_graph"]["patches_37x45_mean"]).max() <= 1e-6


def test_c1_sampled_logits(golden):
    g = golden["c1"]
    p = O.make_unet_params(1, 2, 32, 4, seed=0)
    x = torch.from_numpy(O.formula_normal("c1/x", (1, 1, 256, 256), seed=0))
    with torch.no_grad():
        lg, sk, ft = O.unet_forward(p, x, 4)
    assert np.array_equal(g["idx"], sample_idx("c1/idx", lg.numel(), 4096))
    assert np.abs(lg.reshape(-1)[g["idx"]].numpy() - g["logits"]).max() <= 1e-5
    sums = np.array([[float(t.double().sum()), float(t.double().abs().sum())] for t in [lg] + sk + ft])
    assert np.allclose(sums, g["sums"], rtol=1e-6, atol=1e-3)


def test_c2_image0_and_gat(golden):
    g = golden["c2"]
    p = O.make_unet_params(3, 2, 32, 4, seed=0)
    gp = O.make_gat_params(32, 128, 64, 4, 1, seed=0)
    x = torch.from_numpy(O.formula_normal("c2/x/0", (1, 3, 512, 512), seed=1))
    ei = torch.from_numpy(O.patch_graph_edges(512, 512, 16))
    with torch.no_grad():
        lg, _, ft = O.unet_forward(p, x, 4)
        X = O.patch_mean_features(ft[0][0], 16)
        y = O.gat_network_forward(gp, X, ei, 4)
    assert np.abs(lg.reshape(-1)[g["idx_0"]].numpy() - g["logits_0"]).max() <= 1e-5
    assert np.abs(y.reshape(-1)[g["gidx_0"]].numpy() - g["gat_0"]).max() <= 1e-5


def test_c5_small_train_step(golden):
    g = golden["c5"]
    p = O.make_unet_params(3, 2, 32, 4, seed=0)
    x = torch.from_numpy(O.formula_normal("c5/s/x", (2, 3, 128, 128), seed=4))
    y = torch.from_numpy(O.formula_labels("c5/s/y", (2, 128, 128), 2, seed=5))
    loss, grads, newp, stats, _, _ = O.train_step(p, x, y, 4)
    assert abs(float(loss) - float(g["s_loss"])) <= 1e-5
    names = [str(n) for n in g["param_names"]]
    gn = np.array([float(grads[k].norm()) for k in names])
    assert np.allclose(gn, g["s_grad_norms"], rtol=2e-3, atol=1e-7)
    flat_p = torch.cat([newp[k].reshape(-1) for k in names])
    assert np.abs(flat_p[g["s_idx"]].numpy() - g["s_param_s"]).max() <= 2e-5
    assert np.abs(stats["encoder.encoder_blocks.0.bn1.running_mean"].numpy() - g["s_bn_rm"]).max() <= 1e-5


MINCUT_CASES = {  # tag -> (N, D, K, hidden, use_gnn, heads, seed, x name/seed/scale, logit shift)
    "a": (64, 64, 2, 32, True, 2, 5, ("mincut/a/x", 1, 0.15), None),
    "b": (50, 24, 3, None, False, 1, 6, ("mincut/b/x", 2, 0.2), None),
    "c": (50, 24, 3, None, False, 1, 6, ("mincut/b/x", 2, 0.2), [0.0, -60.0, 0.0]),
    "d": (1024, 64, 2, 32, True, 2, 7, ("mincut/d/x", 4, 0.15), None),
}


def mincut_case(golden, tag):
    N, D, K, hidden, use_gnn, heads, seed, (xn, xs, xscale), shift = MINCUT_CASES[tag]
    g = golden["mincut"]
    ei = {"a": O.patch_graph_edges(128, 128, 16), "d": O.patch_graph_edges(512, 512, 16)}.get(tag)
    if ei is None:
        ei = g["b_edges"]
    X = torch.from_numpy(O.formula_normal(xn, (N, D), seed=xs)) * xscale
    p = O.make_segment_predictor_params(D, K, hidden, use_gnn, heads, seed=seed)
    return g, X, torch.from_numpy(ei), K, p, use_gnn, heads, hidden, (torch.tensor(shift) if shift else None)


@pytest.mark.parametrize("tag", list(MINCUT_CASES))
def test_mincut_stage_vs_reference_fixture(golden, tag):
    """PatchSegmentPredictor + MinCutRefinement.forward: predictor logits, softmax, edge weights and the
    normalized-cut loss against what the reference's own classes produced (oracle/make_golden.py gen_mincut)."""
    g, X, ei, K, p, use_gnn, heads, hidden, shift = mincut_case(golden, tag)
    lg = O.segment_predictor_forward(p, X, ei, use_gnn, heads)
    if shift is not None:
        lg = lg + shift
    assert np.abs(lg.numpy() - g[f"{tag}_logits"]).max() <= 1e-5
    loss, soft, hard = O.mincut_forward(X, ei, K, lg)
    assert np.abs(soft.numpy() - g[f"{tag}_soft"]).max() <= 1e-6
    assert np.abs(O.ncut_edge_weights(X, ei).numpy() - g[f"{tag}_w"]).max() <= 1e-6
    assert abs(float(loss) - float(g[f"{tag}_loss"])) <= 1e-5 * max(1.0, float(g[f"{tag}_loss"]))
    assert np.array_equal(hard.numpy(), g[f"{tag}_soft"].argmax(1))
    if tag == "c":   # the empty segment is skipped, not divided by (mincut_refinement.py:152-153)
        assert float(soft[:, 1].max()) < 1e-20 and 0.9 < float(loss) < 1.1
    with pytest.raises(ValueError):
        O.normalized_cut_loss(X, ei, torch.zeros(X.shape[0], K + 1), K)


REGION_CASES = {"a": (64, 64, 2, 4), "b": (37, 45, 3, 2), "c": (32, 48, 1, 4), "d": (512, 512, 2, 4)}   # tag -> (H, W, K, heads)


def region_case(golden, tag):
    H, W, K, heads = REGION_CASES[tag]
    g = golden["region"]
    nph, npw = O.patch_grid(H, W, 16)
    feats = torch.from_numpy(O.formula_normal(f"region/{tag}/x", (nph * npw, 64), seed=1)) * 0.5
    hard = torch.from_numpy(g[f"{tag}_hard"])
    p = O.make_gat_params(64, 128, 64, heads, 1, seed=9)
    fu = torch.from_numpy(O.formula_normal(f"region/{tag}/fu", (1, 32, H, W), seed=3))
    return g, H, W, K, heads, nph, npw, feats, hard, p, fu


@pytest.mark.parametrize("tag", list(REGION_CASES))
def test_region_stage_and_fusion_vs_reference_fixture(golden, tag):
    """Label-mean pooling -> region GAT -> map back -> nearest upsample -> FeatureFusion concat against
    the reference's GATNetwork / F.interpolate / FeatureFusion outputs (oracle/make_golden.py gen_region)."""
    g, H, W, K, heads, nph, npw, feats, hard, p, fu = region_case(golden, tag)
    emb, pix = O.region_stage(feats, hard, K, p, heads, nph, npw, H, W)
    assert np.abs(emb.numpy() - g[f"{tag}_emb"]).max() <= 1e-6
    assert np.abs(pix.reshape(-1).numpy()[g[f"{tag}_pix_idx"]] - g[f"{tag}_pix"]).max() <= 1e-6
    fused = O.feature_fusion([fu], pix.unsqueeze(0))
    assert tuple(fused.shape) == (1, 96, H, W)
    assert np.abs(fused.reshape(-1).numpy()[g[f"{tag}_fused_idx"]] - g[f"{tag}_fused"]).max() <= 1e-6
    if tag == "b":   # the emptied segment's region feature is zero before the GAT (train_end_to_end.py:369-373)
        assert int((hard == 1).sum()) == 0


DETHEAD_CASES = {"a": (2, 96, 24, 40, 1, False), "b": (1, 96, 128, 128, 3, False), "c": (3, 64, 17, 9, 2, False), "f": (4, 24, 0, 0, 1, True)}


def dethead_case(tag):
    B, C, H, W, ncls, flat = DETHEAD_CASES[tag]
    p = O.make_detection_head_params(C, ncls, 256, flat, seed=13)
    x = torch.from_numpy(O.formula_normal(f"det/{tag}/x", (B, C) if flat else (B, C, H, W), seed=2))
    return p, x, ncls, flat, C


@pytest.mark.parametrize("tag", list(DETHEAD_CASES))
def test_detection_head_vs_reference_fixture(golden, tag):
    """DetectionHead.forward in eval mode (detection_head.py:69-114) against the reference class's outputs."""
    p, x, ncls, flat, _ = dethead_case(tag)
    g = golden["dethead"]
    with torch.no_grad():
        got = O.detection_head_forward(p, x, ncls, flat)
    assert len(got) == (3 if ncls > 1 else 2)
    for nm, t in zip(("bbox", "conf", "cls"), got):
        assert np.abs(t.numpy() - g[f"{tag}_{nm}"]).max() <= 1e-6


def test_aux_losses_and_feature_fusion_branches_vs_reference_fixtures(golden):
    """FeatureConsistencyLoss / EllipticalShapeLoss values and FeatureFusion's bilinear-resize and region-map branches as the
    reference's own classes produced them (oracle/make_golden.py gen_losses)."""
    g = golden["losses"]
    for tag, (B, N, D, margin, scale) in {"fc_a": (2, 64, 64, 1.0, 0.1), "fc_b": (3, 1024, 32, 2.5, 0.3), "fc_c": (1, 7, 20, 0.5, 1.0)}.items():
        fu = torch.from_numpy(O.formula_normal(f"loss/{tag}/u", (B, N, D), seed=1)) * scale
        fg = fu + torch.from_numpy(O.formula_normal(f"loss/{tag}/g", (B, N, D), seed=2)) * scale * 0.5
        y = torch.from_numpy(O.formula_labels(f"loss/{tag}/y", (B, N), 2, seed=3))
        fg[0, 0] = fu[0, 0]
        assert abs(float(O.feature_consistency_loss(fu, fg, y, margin)) - float(g[tag])) <= 1e-6 * max(1.0, float(g[tag]))
    masks = [[torch.from_numpy(m.astype(bool)) for m in img] for img in g["shape_masks_in"]]
    assert abs(float(O.elliptical_shape_loss(None, masks, 1e-6)) - float(g["shape_masks"])) <= 1e-5
    probs = torch.from_numpy(g["shape_probs_in"])
    assert abs(float(O.elliptical_shape_loss(probs, None, 1e-6)) - float(g["shape_probs"])) <= 1e-5
    assert float(O.elliptical_shape_loss(probs[:, :1])) == 0.0
    fu0 = torch.from_numpy(O.formula_normal("loss/ff/u0", (2, 8, 24, 40), seed=5))
    fu1 = torch.from_numpy(O.formula_normal("loss/ff/u1", (2, 16, 12, 20), seed=6))
    fu2 = torch.from_numpy(O.formula_normal("loss/ff/u2", (2, 4, 7, 9), seed=7))
    fg4 = torch.from_numpy(O.formula_normal("loss/ff/g4", (2, 12, 5, 11), seed=8))
    assert float((O.feature_fusion_full([fu0, fu1, fu2], fg4, 12) - torch.from_numpy(g["ff_multi"])).abs().max()) <= 1e-6
    assert float((O.feature_fusion_full([fu0, fu1, fu2], fg4, 12, target_spatial_size=(33, 17)) - torch.from_numpy(g["ff_target"])).abs().max()) <= 1e-6
    fg2 = torch.from_numpy(O.formula_normal("loss/ff/g2", (7, 12), seed=9))
    rmap = torch.from_numpy(g["ff_regions_map"])
    assert torch.equal(O.feature_fusion_full([fu0, fu1], fg2, 12, region_to_pixel_map=rmap), torch.from_numpy(g["ff_regions"]))


GATGRAD_CASES = {   # tag: (cfg (in, hidden, out, heads), layers, nodes, graph, x scale, weight scale) -- as oracle/make_golden.py
    "g10": ((32, 64, 16, 4), 1, 10, "edge10", 1.0, 1.0), "iso": ((32, 64, 16, 4), 1, 12, "edge_iso", 1.0, 1.0),
    "wide": ((32, 64, 16, 2), 1, 10, "edge10", 4.0, 3.0), "mid": ((32, 64, 16, 4), 1, 10, "edge10", 2.0, 1.5),
    "l2h1": ((32, 24, 8, 1), 2, 10, "edge10", 1.0, 1.0), "grid": ((32, 128, 64, 4), 1, 256, "grid16", 1.0, 1.0),
    "pred": ((64, 64, 2, 2), 1, 64, "grid8", 1.0, 1.0)}


def gatgrad_inputs(golden, tag):
    cfg, layers, N, graph, xs, ws = GATGRAD_CASES[tag]
    if graph in ("edge10", "edge_iso"):
        ei = golden["gat_small"][graph]
    else:
        side = int(graph[4:])
        ei = O.patch_graph_edges(side * 16, side * 16, 16)
    X = torch.from_numpy(O.formula_normal(f"gatgrad/{tag}/x", (N, cfg[0]), seed=3)) * xs
    R = torch.from_numpy(O.formula_normal(f"gatgrad/{tag}/r", (N, cfg[2]), seed=4))
    p = O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=3, scale=ws)
    return cfg, layers, torch.from_numpy(np.ascontiguousarray(ei)), X, R, p


@pytest.mark.parametrize("tag", list(GATGRAD_CASES))
def test_gat_gradients_vs_reference_fixture(golden, tag):
    """d/dX, d/dW, d/da of sum(GATNetwork(X) * R) as the REFERENCE class produced them under torch autograd (eval-mode dropout;
    tests/golden/gat_grad.npz): the oracle restatement reproduces them, including the gradient through the graph-wide max."""
    g = golden["gat_grad"]
    cfg, layers, ei, X, R, p = gatgrad_inputs(golden, tag)
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Xo = X.clone().requires_grad_(True)
    (O.gat_network_forward(q, Xo, ei, cfg[3], layers) * R).sum().backward()
    assert np.abs(Xo.grad.numpy() - g[tag + "_dX"]).max() <= 2e-6 * max(1.0, np.abs(g[tag + "_dX"]).max())
    for k in q:
        ref = g[f"{tag}_d_{k}"]
        assert np.abs(q[k].grad.numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), k


SCRIPT_TV = {"tv_a": ((2, 1, 17, 23), 1.0, 1), "tv_b": ((3, 2, 64, 48), 0.37, 2), "tv_c": ((1, 1, 2, 2), 1.0, 3), "tv_d": ((8, 1, 128, 128), 0.1, 4)}
SCRIPT_DICE = {"dice_a": (2, 2, 16, 16, 1.0, 2.0, 5), "dice_b": (3, 4, 33, 20, 0.5, 1.0, 6), "dice_c": (1, 2, 128, 128, 1.0, 3.0, 7),
               "dice_d": (2, 3, 8, 8, 1e-3, 8.0, 8)}
SCRIPT_FC = {"fc_a": (2, 64, 64, 1.0, 0.1), "fc_b": (3, 1024, 32, 2.5, 0.3), "fc_c": (1, 7, 20, 0.5, 1.0)}


def script_dice_inputs(g, tag):
    B, C, H, W, smooth, scale, seed = SCRIPT_DICE[tag]
    lg = torch.from_numpy(O.formula_normal(f"sloss/{tag}/x", (B, C, H, W), seed=seed)) * scale
    y = torch.from_numpy(g[tag + "_y"]) if tag + "_y" in g.files else torch.from_numpy(O.formula_labels(f"sloss/{tag}/y", (B, H, W), C, seed=seed + 10))
    return lg, y, smooth


def script_fc_inputs(tag):
    B, N, D, margin, scale = SCRIPT_FC[tag]
    fu = torch.from_numpy(O.formula_normal(f"loss/{tag}/u", (B, N, D), seed=1)) * scale
    fg = fu + torch.from_numpy(O.formula_normal(f"loss/{tag}/g", (B, N, D), seed=2)) * scale * 0.5
    y = torch.from_numpy(O.formula_labels(f"loss/{tag}/y", (B, N), 2, seed=3))
    fg[0, 0] = fu[0, 0] + 1e-3
    return fu, fg, y, margin


def test_tv_dice_featcons_values_and_gradients_vs_reference_fixture(golden):
    """tests/golden/script_losses.npz holds what the reference's OWN TVLoss (scripts/train_end_to_end.py:73-89) and dice_loss
    (scripts/train_segmentation.py:29-40) definitions returned -- executed from the reference files' syntax trees by
    oracle/make_golden.py, their modules need cv2 -- with the gradients torch autograd gave them, and the gradients of
    FeatureConsistencyLoss: the restatements are pinned by them, values and gradients."""
    g = golden["script_losses"]
    for tag, (shape, weight, seed) in SCRIPT_TV.items():
        x = torch.from_numpy(O.formula_normal(f"sloss/{tag}/x", shape, seed=seed)).requires_grad_(True)
        v = O.tv_loss(x, weight)
        v.backward()
        assert abs(float(v) - float(g[tag])) <= 1e-6 * max(1.0, float(g[tag])), tag
        assert np.abs(x.grad.numpy() - g[tag + "_grad"]).max() <= 1e-7, tag
    for tag in SCRIPT_DICE:
        lg, y, smooth = script_dice_inputs(g, tag)
        lg.requires_grad_(True)
        v = O.dice_loss(lg, y, smooth)
        v.backward()
        assert abs(float(v) - float(g[tag])) <= 1e-6, tag
        assert np.abs(lg.grad.numpy() - g[tag + "_grad"]).max() <= 1e-8, tag
        l2 = lg.detach().clone().requires_grad_(True)
        tot = torch.nn.functional.cross_entropy(l2, y) + O.dice_loss(l2, y, smooth)
        tot.backward()
        assert abs(float(tot) - float(g[tag + "_cedice"])) <= 1e-6 * float(g[tag + "_cedice"]), tag
        assert np.abs(l2.grad.numpy() - g[tag + "_cedice_grad"]).max() <= 1e-7, tag
    for tag in SCRIPT_FC:
        fu, fg, y, margin = script_fc_inputs(tag)
        fu.requires_grad_(True), fg.requires_grad_(True)
        v = O.feature_consistency_loss(fu, fg, y, margin)
        v.backward()
        idx = torch.from_numpy(g[tag + "_idx"])
        assert abs(float(v) - float(g[tag + "_val"])) <= 1e-6 * max(1.0, float(g[tag + "_val"])), tag
        assert np.abs(fu.grad.reshape(-1)[idx].numpy() - g[tag + "_grad_u"]).max() <= 1e-7, tag
        assert np.abs(fg.grad.reshape(-1)[idx].numpy() - g[tag + "_grad_g"]).max() <= 1e-7, tag


def test_train_step_ce_plus_dice_vs_reference_fixture(golden):
    """`loss = loss_ce + loss_dice; loss.backward()` (scripts/train_segmentation.py:126-133) on the reference UNet with the
    reference's dice_loss definition: the oracle's train_step(loss_kind="ce+dice") reproduces loss and gradients."""
    g = golden["script_losses"]
    cfg = (3, 2, 8, 2)
    p = O.make_unet_params(*cfg, seed=21)
    x = torch.from_numpy(O.formula_normal("sloss/tr/x", (2, 3, 32, 32), seed=22))
    y = torch.from_numpy(O.formula_labels("sloss/tr/y", (2, 32, 32), 2, seed=23))
    loss, grads, _, _, _, _ = O.train_step(p, x, y, cfg[3], loss_kind="ce+dice")
    assert abs(float(loss) - float(g["tr_loss"])) <= 1e-6 * float(g["tr_loss"])
    names = [str(n) for n in g["tr_names"]]
    assert list(grads.keys()) == names
    gn = np.array([float(grads[k].norm()) for k in names])
    big = g["tr_grad_norms"] > 1e-4 * g["tr_grad_norms"].max()
    assert np.all(np.abs(gn[big] / g["tr_grad_norms"][big] - 1) <= 2e-3)


def test_tv_and_dice_known_answers():
    """Hand-computed answers from scripts/train_end_to_end.py:84-88 and scripts/train_segmentation.py:30-40 (kept beside the
    reference-generated fixtures of the test above)."""
    x = torch.arange(24.0).reshape(1, 1, 4, 6)              # vertical steps 6, horizontal steps 1
    assert float(O.tv_loss(x)) == 37.0                      # 18*36/18 + 20*1/20
    assert float(O.tv_loss(torch.cat([x, x]), weight=0.5)) == 18.5     # sums double, / batch 2, * weight
    assert float(O.tv_loss(torch.ones(2, 3, 5, 5))) == 0.0
    z = torch.zeros(1, 2, 2, 2)                             # uniform probabilities 0.5
    t = torch.tensor([[[0, 1], [1, 1]]])
    # class 0: I = .5, P = 2, T = 1 -> 2/4;  class 1: I = 1.5, P = 2, T = 3 -> 4/6;  1 - mean = 5/12
    assert abs(float(O.dice_loss(z, t)) - 5.0 / 12.0) <= 1e-6
    big = torch.full((1, 2, 2, 2), -30.0)
    big[0, 0, 0, 0] = big[0, 1, 0, 1] = big[0, 1, 1, 0] = big[0, 1, 1, 1] = 30.0     # a perfect prediction: dice 1 per class
    assert abs(float(O.dice_loss(big, t))) <= 1e-6
    assert abs(float(O.dice_loss(z, t, smooth=0.0)) - (1 - (1 / 3 + 3 / 5) / 2)) <= 1e-6


def test_input_pipeline_restatements_self_consistency():
    """The pipeline restatements (cv2 / torchvision absent: written against the libraries' published definitions) on cases with
    known answers: identity-size resize is the identity, nearest mask resize picks floor(dst * src / dst), a flat image has no
    edges, equalising a two-level image stretches it to 0 / 255, patch means zero-pad."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    t = O.preprocess_image(img, (20, 30), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), bgr=True)
    assert torch.equal(t, torch.from_numpy(img[:, :, ::-1].copy()).permute(2, 0, 1).float().div(255))
    m = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert torch.equal(O.preprocess_mask(m, (6, 8), 100), torch.from_numpy(np.repeat(np.repeat(m, 2, 0), 2, 1).astype(np.int64)))
    assert int(O.preprocess_mask(m, (2, 2), 5).max()) == 4                       # np.clip to num_classes - 1
    assert int(O.sobel_edges(np.full((9, 9, 3), 77, np.uint8)).max()) == 0
    step = np.zeros((8, 8, 3), np.uint8)
    step[:, 4:] = 200
    e = O.sobel_edges(step)
    assert e[:, 3:5].min() == 255 and e[:, :2].max() == 0 and e[:, 6:].max() == 0
    two = np.full((4, 4, 3), 60, np.uint8)
    two[2:] = 180
    h = O.equalize_histogram_rgb(two)
    assert h[:2].max() <= 1 and h[2:].min() >= 254                               # luminance stretched to the ends
    pm = O.patch_mean_u8(np.full((5, 5), 100, np.uint8), 4)
    assert pm.shape == (4, 1) and abs(float(pm[0]) - 100) < 1e-4 and abs(float(pm[3]) - 100 / 16) < 1e-4


@pytest.mark.parametrize("tag", list(O.MINCUTGRAD_CASES))
def test_mincut_gradients_vs_reference_fixture(golden, tag):
    """L_partition.backward() (train_end_to_end.py:348-356, 472-479) as the REFERENCE MinCutRefinement + predictor produced it
    under torch autograd (tests/golden/mincut_grad.npz): (i) the analytic gradient the HIP kernel implements
    (O.normalized_cut_loss_grad) equals the reference's autograd of normalized_cut_loss called directly, (ii) the oracle
    restatement under autograd reproduces d/dX and the predictor's parameter gradients of the whole stage."""
    g = golden["mincut_grad"]
    ei, X, R, p, K, use_gnn, heads, shift = O.mincutgrad_inputs(tag)
    lg = O.segment_predictor_forward(p, X, ei, use_gnn, heads)
    soft = torch.softmax(lg if shift is None else lg + shift, dim=1)
    aP, aF = O.normalized_cut_loss_grad(X, ei, soft, K, gloss=2.5)
    for got, key in ((aP, "_direct_dP"), (aF, "_direct_dX")):
        ref = g[tag + key]
        assert np.abs(got.numpy() - ref).max() <= 5e-6 * max(1.0, np.abs(ref).max()), key
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Xo = X.clone().requires_grad_(True)
    lg = O.segment_predictor_forward(q, Xo, ei, use_gnn, heads)
    loss, so, _ = O.mincut_forward(Xo, ei, K, lg if shift is None else lg + shift)
    (loss + 0.05 * (so * R).sum()).backward()
    assert abs(float(loss.detach()) - float(g[tag + "_loss"])) <= 1e-5 * max(1.0, float(g[tag + "_loss"]))
    assert np.abs(Xo.grad.numpy() - g[tag + "_dX"]).max() <= 5e-6 * max(1.0, np.abs(g[tag + "_dX"]).max())
    for k in q:
        ref = g[f"{tag}_d_{k}"]
        assert np.abs(q[k].grad.numpy() - ref).max() <= 5e-6 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize("tag,cfg,shape,xname,xseed,pseed", [("b", (3, 3, 8, 2), (2, 3, 37, 45), "tiny/b/x", 11, 11),
                                                             ("c", (3, 2, 8, 3), (2, 3, 64, 48), "tiny/c/x", 11, 11)])
def test_bf16_storage_emulation_within_the_references_own_bf16_deviation(golden, tag, cfg, shape, xname, xseed, pseed):
    """The oracle's bf16-storage restatement (the kernel-level yardstick of tests/test_gpu_bf16.py) against the fixture of the
    REFERENCE U-Net run in bfloat16 on the CPU (tests/golden/bf16_reference.npz, oracle/make_golden.py gen_bf16ref): its deviation
    from the fp32 forward stays within 1.25 x what the reference's own bf16 run shows, and the fixture's fp32 samples are the
    oracle's fp32 logits."""
    g = golden["bf16_reference"]
    ref_max, ref_mean, _, ref_agree, scale = [float(v) for v in g[f"{tag}_stats"]]
    p = O.make_unet_params(*cfg, seed=pseed)
    x = torch.from_numpy(O.formula_normal(xname, shape, seed=xseed))
    with torch.no_grad():
        lf = O.unet_forward(p, x, cfg[3])[0]
        lb = O.unet_forward_bf16_storage(p, x, depth=cfg[3], first_fp32=False)[0]
    idx = torch.from_numpy(g[f"{tag}_idx"])
    assert float((lf.reshape(-1)[idx] - torch.from_numpy(g[f"{tag}_fp32"])).abs().max()) <= 1e-5 * scale
    d = (lb - lf).abs()
    assert float(d.max()) <= 1.25 * ref_max and float(d.mean()) <= 1.25 * ref_mean
    assert float((lb.argmax(1) == lf.argmax(1)).float().mean()) >= ref_agree - 0.01
