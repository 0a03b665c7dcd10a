"""Auxiliary losses, FeatureFusion's resize / region-map branches and the input / output pipeline (row 4) on
the HIP path, through the mirrors of the reference's classes:
  * FeatureConsistencyLoss, EllipticalShapeLoss, FeatureFusion: against fixtures the reference's own classes produced
    (tests/golden/losses.npz) and against the oracle;
  * TVLoss, dice_loss: values AND gradients against what the reference's own definitions produced (tests/golden/script_losses.npz:
    oracle/make_golden.py executes the class / function nodes of the reference scripts), hand-computed answers, the oracle;
  * the gradients of TVLoss / dice_loss / FeatureConsistencyLoss (HIP backward kernels behind torch.autograd.Function nodes);
  * ImagePreprocessor / EdgeDetector / HistogramEqualizer / patch means / colour map: BIT-EXACT against the oracle's PIL / numpy
    restatement (byte and integer work)."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O

pytestmark = pytest.mark.gpu


def close(a, b, rel):
    return abs(float(a) - float(b)) <= rel * max(1.0, abs(float(b)))


def test_tv_loss(cuda):
    x = torch.arange(24.0).reshape(1, 1, 4, 6)
    assert float(mgunet.TVLoss()(x.to(cuda))) == 37.0
    assert float(mgunet.TVLoss(weight=0.5)(torch.cat([x, x]).to(cuda))) == 18.5
    big = torch.from_numpy(O.formula_normal("tv/x", (3, 2, 257, 190), seed=1))
    assert close(mgunet.TVLoss(0.7)(big.to(cuda)), O.tv_loss(big.double(), 0.7), 1e-6)
    # NHWC storage is read in place through its strides (the U-Net's logits are stored that way)
    nhwc = big.to(cuda).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert float(mgunet.TVLoss(0.7)(nhwc)) == float(mgunet.TVLoss(0.7)(big.to(cuda)))


def test_dice_loss(cuda):
    z = torch.zeros(1, 2, 2, 2)
    t = torch.tensor([[[0, 1], [1, 1]]])
    assert close(mgunet.dice_loss(z.to(cuda), t.to(cuda)), 5.0 / 12.0, 1e-6)
    lg = torch.from_numpy(O.formula_normal("dice/l", (3, 4, 65, 50), seed=2)) * 2
    y = torch.from_numpy(O.formula_labels("dice/y", (3, 65, 50), 4, seed=3))
    ref = O.dice_loss(lg.double(), y, 1.0)
    assert close(mgunet.dice_loss(lg.to(cuda), y.to(cuda)), ref, 1e-6)
    nhwc = lg.to(cuda).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)      # the layout mgunet.UNet returns
    assert close(mgunet.dice_loss(nhwc, y.to(cuda), smooth=0.5), O.dice_loss(lg.double(), y, 0.5), 1e-6)
    lg7 = torch.from_numpy(O.formula_normal("dice/l7", (2, 7, 20, 20), seed=4))
    y7 = torch.from_numpy(O.formula_labels("dice/y7", (2, 20, 20), 7, seed=5))
    assert close(mgunet.dice_loss(lg7.to(cuda), y7.to(cuda)), O.dice_loss(lg7.double(), y7), 1e-6)


def test_tv_dice_featcons_values_and_gradients_vs_reference_fixture(cuda, golden):
    """Values and autograd gradients of the HIP losses against the reference definitions' own outputs (script_losses.npz)."""
    from test_oracle_golden import SCRIPT_DICE, SCRIPT_FC, SCRIPT_TV, script_dice_inputs, script_fc_inputs
    g = golden["script_losses"]
    for tag, (shape, weight, seed) in SCRIPT_TV.items():
        x = torch.from_numpy(O.formula_normal(f"sloss/{tag}/x", shape, seed=seed)).to(cuda).requires_grad_(True)
        v = mgunet.TVLoss(weight)(x)
        assert close(v, g[tag], 2e-6), (tag, float(v), float(g[tag]))
        (3.0 * v).backward()                                      # an upstream factor: the grad_output path
        ref = 3.0 * g[tag + "_grad"]
        assert np.abs(x.grad.cpu().numpy() - ref).max() <= 2e-6 * max(1e-3, np.abs(ref).max()) + 1e-9, tag
    for tag in SCRIPT_DICE:
        lg, y, smooth = script_dice_inputs(g, tag)
        for nhwc in (False, True):                                # NCHW storage and the NHWC storage mgunet.UNet returns
            l = lg.to(cuda)
            if nhwc:
                l = l.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            l.requires_grad_(True)
            v = mgunet.dice_loss(l, y.to(cuda), smooth)
            assert close(v, g[tag], 2e-6), (tag, float(v), float(g[tag]))
            v.backward()
            ref = g[tag + "_grad"]
            assert np.abs(l.grad.cpu().numpy() - ref).max() <= 5e-6 * np.abs(ref).max() + 1e-10, (tag, nhwc)
    for tag in SCRIPT_FC:
        fu, fg, y, margin = script_fc_inputs(tag)
        a, b = fu.to(cuda).requires_grad_(True), fg.to(cuda).requires_grad_(True)
        v = mgunet.FeatureConsistencyLoss(margin)(a, b, y.to(cuda))
        assert close(v, g[tag + "_val"], 2e-6), tag
        v.backward()
        idx = torch.from_numpy(g[tag + "_idx"])
        for got, key in ((a.grad, "_grad_u"), (b.grad, "_grad_g")):
            ref = g[tag + key]
            assert np.abs(got.cpu().reshape(-1)[idx].numpy() - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-9, (tag, key)
    # an out-of-range label: F.one_hot raises in the reference; here the next check_labels() does
    lg, y, _ = script_dice_inputs(g, "dice_a")
    y = y.clone()
    y[0, 0, 0] = 5
    mgunet.dice_loss(lg.to(cuda), y.to(cuda))
    with pytest.raises(ValueError, match="label outside"):
        mgunet.losses.check_labels(cuda)
    mgunet.losses.check_labels(cuda)                              # reported once


def test_trainer_ce_plus_dice_vs_reference_fixture(cuda, golden):
    """Trainer(loss="ce+dice") = `loss = loss_ce + loss_dice; loss.backward()` of scripts/train_segmentation.py:126-133 against
    the step the reference UNet + the reference's dice_loss definition took under torch autograd (script_losses.npz, float64
    gradients as the yardstick; contract: per-parameter gradient norm within 2 * cond + 1e-3 of float64, cond = the deviation
    of the reference's own fp32 gradients from float64, and the flat gradient within 2e-3 relative L2)."""
    g = golden["script_losses"]
    cfg = (3, 2, 8, 2)
    model = mgunet.UNet(*cfg)
    model.load_state_dict(O.make_unet_params(*cfg, seed=21))
    model = model.to(cuda)
    x = torch.from_numpy(O.formula_normal("sloss/tr/x", (2, 3, 32, 32), seed=22)).to(cuda)
    y = torch.from_numpy(O.formula_labels("sloss/tr/y", (2, 32, 32), 2, seed=23)).to(cuda)
    tr = mgunet.Trainer(model, loss="ce+dice", comm=None)
    loss = tr.forward_backward(x, y)
    assert abs(float(loss) - float(g["tr_loss"])) <= 1e-5 * float(g["tr_loss"]), (float(loss), float(g["tr_loss"]))
    names = [str(n) for n in g["tr_names"]]
    params = dict(model.named_parameters())
    assert list(params.keys()) == names
    gn = np.array([float(params[k].grad.norm()) for k in names])
    ref, cond = g["tr_grad_norms64"], g["tr_cond"]
    big = ref > 1e-4 * ref.max()
    assert np.all(np.abs(gn[big] / ref[big] - 1) <= 2 * cond[big] + 1e-3)
    flat = tr.grad.cpu().numpy()
    r64 = g["tr_grad_flat64"]
    assert np.linalg.norm(flat - r64) <= 2e-3 * np.linalg.norm(r64)
    with pytest.raises(ValueError):
        mgunet.Trainer(model, loss="dice")


def test_feature_consistency_loss_vs_reference_fixture(cuda, golden):
    g = golden["losses"]
    for tag, (B, N, D, margin, scale) in {"fc_a": (2, 64, 64, 1.0, 0.1), "fc_b": (3, 1024, 32, 2.5, 0.3), "fc_c": (1, 7, 20, 0.5, 1.0)}.items():
        fu = torch.from_numpy(O.formula_normal(f"loss/{tag}/u", (B, N, D), seed=1)) * scale
        fg = fu + torch.from_numpy(O.formula_normal(f"loss/{tag}/g", (B, N, D), seed=2)) * scale * 0.5
        y = torch.from_numpy(O.formula_labels(f"loss/{tag}/y", (B, N), 2, seed=3))
        fg[0, 0] = fu[0, 0]
        got = mgunet.FeatureConsistencyLoss(margin=margin)(fu.to(cuda), fg.to(cuda), y.to(cuda))
        assert close(got, g[tag], 2e-6), (tag, float(got), float(g[tag]))
    with pytest.raises(ValueError):   # the 2-D call of train_end_to_end.py:344 fails in the reference too (B, N, D = f_unet.shape)
        mgunet.FeatureConsistencyLoss()(torch.zeros(8, 4, device=cuda), torch.zeros(8, 4, device=cuda), torch.zeros(8, device=cuda))
    with pytest.raises(ValueError, match="correspondence_map_y"):
        mgunet.FeatureConsistencyLoss()(torch.zeros(1, 8, 4, device=cuda), torch.zeros(1, 8, 4, device=cuda), torch.zeros(8, device=cuda))


def test_elliptical_shape_loss_vs_reference_fixture(cuda, golden):
    g = golden["losses"]
    masks = [[torch.from_numpy(m.astype(bool)).to(cuda) for m in img] for img in g["shape_masks_in"]]
    got = mgunet.EllipticalShapeLoss(epsilon=1e-6)(None, object_masks_list=masks)
    assert close(got, g["shape_masks"], 2e-5), (float(got), float(g["shape_masks"]))      # the reference accumulates in fp32
    probs = torch.from_numpy(g["shape_probs_in"]).to(cuda)
    got = mgunet.EllipticalShapeLoss(epsilon=1e-6)(probs)
    assert close(got, g["shape_probs"], 2e-5), (float(got), float(g["shape_probs"]))
    assert float(mgunet.EllipticalShapeLoss()(probs[:, :1])) == 0.0                        # one class: nothing to analyse (:63-64)
    assert float(mgunet.EllipticalShapeLoss()(None, object_masks_list=[[torch.zeros(8, 8, dtype=torch.bool, device=cuda)]])) == 0.0
    # a full-size case: four 512 x 512 objects, against the float64 oracle
    yy, xx = np.mgrid[0:512, 0:512]
    big = [[torch.from_numpy(((yy - 250) / (60 + 30 * k)) ** 2 + ((xx - 260) / (150 - 20 * k)) ** 2 <= 1) for k in range(4)]]
    ref = O.elliptical_shape_loss(None, [[m for m in big[0]]], 1e-6)
    got = mgunet.EllipticalShapeLoss()(None, object_masks_list=[[m.to(cuda) for m in big[0]]])
    assert close(got, ref, 1e-3), (float(got), float(ref))      # fp32 reference: its N x N Mahalanobis product loses digits at 5e4 pixels


def test_feature_fusion_resize_and_region_map_vs_reference_fixture(cuda, golden):
    g = golden["losses"]
    fu0 = torch.from_numpy(O.formula_normal("loss/ff/u0", (2, 8, 24, 40), seed=5)).to(cuda)
    fu1 = torch.from_numpy(O.formula_normal("loss/ff/u1", (2, 16, 12, 20), seed=6)).to(cuda)
    fu2 = torch.from_numpy(O.formula_normal("loss/ff/u2", (2, 4, 7, 9), seed=7)).to(cuda)
    fg4 = torch.from_numpy(O.formula_normal("loss/ff/g4", (2, 12, 5, 11), seed=8)).to(cuda)
    got = mgunet.FeatureFusion([8, 16, 4], 12)([fu0, fu1, fu2], fg4)
    assert float((got.cpu() - torch.from_numpy(g["ff_multi"])).abs().max()) <= 2e-6
    got = mgunet.FeatureFusion([8, 16, 4], 12)([fu0, fu1, fu2], fg4, target_spatial_size=(33, 17))
    assert float((got.cpu() - torch.from_numpy(g["ff_target"])).abs().max()) <= 2e-6
    fg2 = torch.from_numpy(O.formula_normal("loss/ff/g2", (7, 12), seed=9)).to(cuda)
    rmap = torch.from_numpy(g["ff_regions_map"]).to(cuda)
    got = mgunet.FeatureFusion([8, 16], 12)([fu0, fu1], fg2, region_to_pixel_map=rmap)
    ref = torch.from_numpy(g["ff_regions"])
    assert torch.equal(got.cpu()[:, :8], ref[:, :8]) and torch.equal(got.cpu()[:, 24:], ref[:, 24:])   # copies and gathers: exact
    assert float((got.cpu() - ref).abs().max()) <= 2e-6
    fadd = torch.from_numpy(O.formula_normal("loss/ff/ga", (2, 24, 6, 10), seed=11)).to(cuda)
    got = mgunet.FeatureFusion([8, 16], 24, fusion_method="add")([fu0, fu1], fadd)
    assert float((got.cpu() - torch.from_numpy(g["ff_add"])).abs().max()) <= 4e-6


@pytest.mark.parametrize("src,dst", [((70, 93), (32, 48)), ((480, 640), (512, 512)), ((1080, 1920), (512, 512)), ((100, 100), (100, 100)),
                                      ((33, 500), (128, 128)), ((512, 512), (1024, 1024))])
def test_image_preprocessor_bit_exact(cuda, src, dst):
    rng = np.random.default_rng(src[0] + dst[0])
    img = rng.integers(0, 256, src + (3,), dtype=np.uint8)
    pre = mgunet.ImagePreprocessor(resize_dim=dst)
    got = pre.preprocess(img)
    ref = O.preprocess_image(img, dst, pre.mean, pre.std, bgr=True)
    assert tuple(got.shape) == (3,) + dst and torch.equal(got.cpu(), ref)
    grey = img[:, :, 0].copy()
    assert torch.equal(pre.preprocess(grey).cpu(), O.preprocess_image(grey, dst, pre.mean, pre.std))
    # straight into an image slot of an NHWC batch (the layout the first convolution reads)
    batch = torch.zeros((2, dst[0], dst[1], 3), device=cuda)
    pre.preprocess(img, out=batch[1].permute(2, 0, 1))
    assert torch.equal(batch[1].permute(2, 0, 1).cpu(), ref) and float(batch[0].abs().max()) == 0.0
    mask = rng.integers(0, 7, src, dtype=np.uint8)
    assert torch.equal(pre.preprocess_mask(mask, 3).cpu(), O.preprocess_mask(mask, dst, 3))


def test_sobel_histeq_patch_means_and_colour_map_bit_exact(cuda):
    rng = np.random.default_rng(11)
    for H, W in ((37, 45), (512, 512), (1, 9)):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        img[: H // 2, : W // 2] //= 4                                    # some structure for the histogram
        s = mgunet.EdgeDetector(kernel_size=3).sobel_edges(img)
        assert isinstance(s, np.ndarray) and np.array_equal(s, O.sobel_edges(img))
        h = mgunet.HistogramEqualizer().equalize_histogram_rgb(img)
        assert np.array_equal(h, O.equalize_histogram_rgb(img))
        assert torch.equal(mgunet.patch_features_u8(s, 16).cpu(), O.patch_mean_u8(s, 16))
        assert float((mgunet.patch_features_u8(h, 16, per_channel=True).cpu() - O.patch_mean_u8(h, 16, True)).abs().max()) <= 1e-5
    flat = np.full((9, 9, 3), 77, np.uint8)
    assert int(mgunet.EdgeDetector().sobel_edges(flat).max()) == 0
    assert np.array_equal(mgunet.HistogramEqualizer().equalize_histogram_rgb(flat), O.equalize_histogram_rgb(flat))
    with pytest.raises(ValueError, match="RGB image"):
        mgunet.EdgeDetector().sobel_edges(np.zeros((4, 4), np.uint8))
    logits = torch.from_numpy(O.formula_normal("pp/l", (1, 3, 40, 56), seed=2)).to(cuda)
    labels, vis = mgunet.postprocess_segmentation(logits, 3)
    ref_l = logits[0].argmax(0).cpu().numpy()
    assert np.array_equal(labels, ref_l) and np.array_equal(vis, O.colorize_labels(ref_l, 3, mgunet.preprocess.DEFAULT_COLORS_BGR))
