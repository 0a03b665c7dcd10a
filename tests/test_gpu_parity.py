"""Parity of the HIP path (through the C-ABI, via the mgunet mirror) against the oracle and the golden
fixtures generated from the reference.  Tolerance: fp32 logits / features within 1e-3 ABSOLUTE of the
reference's CPU fp32 result at |logit| <= ~8 (north star); index maps bit-exact."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3


def to_dev(params, dev):
    return {k: v.to(dev) for k, v in params.items()}


def make_unet(cfg, seed, dev):
    m = mgunet.UNet(*cfg)
    m.load_state_dict(O.make_unet_params(*cfg, seed=seed))
    return m.to(dev).eval()


def make_gat(cfg, dev, seed=0, scale=1.0, layers=1):
    g = mgunet.GATNetwork(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers)
    g.load_state_dict(O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=seed, scale=scale))
    return g.to(dev).eval()


def maxdiff(a, b):
    return float(np.abs(a.detach().cpu().numpy() - np.asarray(b)).max()) if a.numel() else 0.0


def test_native_library_is_the_path(cuda):
    mgunet.lib()
    assert "libmgunet.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("tag,cfg,shape", [("a", (1, 2, 8, 2), (1, 1, 32, 32)), ("b", (3, 3, 8, 2), (2, 3, 37, 45)),
                                           ("c", (3, 2, 8, 3), (2, 3, 64, 48))])
def test_unet_tiny_full_tensors(cuda, golden, tag, cfg, shape):
    g = golden["unet_tiny"]
    m = make_unet(cfg, 11, cuda)
    x = torch.from_numpy(O.formula_normal(f"tiny/{tag}/x", shape, seed=11)).to(cuda)
    lg, sk, ft = m(x)
    assert tuple(lg.shape) == g[f"{tag}_logits"].shape
    assert maxdiff(lg, g[f"{tag}_logits"]) <= TOL
    for i in range(cfg[3]):
        assert tuple(sk[i].shape) == g[f"{tag}_skip{i}"].shape
        assert maxdiff(sk[i], g[f"{tag}_skip{i}"]) <= TOL, f"skip{i}"
        assert maxdiff(ft[i], g[f"{tag}_feat{i}"]) <= TOL, f"feat{i}"
    # expected accuracy of exact-fp32 MFMA is far better than the bar: report it
    print(f"[{tag}] max|logit diff| = {maxdiff(lg, g[f'{tag}_logits']):.2e}")


@pytest.mark.parametrize("cfg,shape", [((3, 2, 8, 2), (2, 3, 50, 70)), ((2, 5, 12, 3), (1, 2, 41, 33)),
                                       ((3, 2, 16, 1), (3, 3, 16, 130)), ((4, 2, 32, 4), (1, 4, 96, 80)),
                                       # Winograd layers + fused max-pool + F.pad on odd sizes at every level
                                       ((3, 2, 16, 2), (2, 3, 37, 45)), ((3, 3, 32, 3), (1, 3, 70, 93))])
def test_unet_vs_oracle_ragged_shapes(cuda, cfg, shape):
    p = O.make_unet_params(*cfg, seed=5)
    x = torch.from_numpy(O.formula_normal("ragged/x", shape, seed=5))
    with torch.no_grad():
        olg, osk, oft = O.unet_forward(p, x, cfg[3])
    m = make_unet(cfg, 5, cuda)
    lg, sk, ft = m(x.to(cuda))
    assert maxdiff(lg, olg.numpy()) <= TOL
    for i in range(cfg[3]):
        assert maxdiff(sk[i], osk[i].numpy()) <= TOL and maxdiff(ft[i], oft[i].numpy()) <= TOL


def test_unet_accepts_channels_last_and_strided_input(cuda):
    cfg = (3, 2, 8, 2)
    m = make_unet(cfg, 5, cuda)
    x = torch.from_numpy(O.formula_normal("strided/x", (2, 3, 40, 48), seed=5)).to(cuda)
    ref = m(x)[0]
    a = m(x.contiguous(memory_format=torch.channels_last))[0]
    big = torch.zeros(2, 3, 44, 50, device=cuda)
    big[:, :, 2:42, 1:49] = x
    b = m(big[:, :, 2:42, 1:49])[0]
    assert torch.equal(ref, a) and torch.equal(ref, b)


@pytest.mark.parametrize("cin", [1, 2, 3])
def test_first_convolution_on_the_matrix_cores_reads_any_view_of_the_image(cuda, cin, monkeypatch):
    """The 32-feature U-Net's first convolution runs on conv3x3_first_mfma_kernel, which reads the caller's image through its strides
    (no packed copy): NCHW, channels_last and a window of a larger tensor give the same bytes, ragged sizes (edge patches) included, and
    the result equals the VALU kernel behind pack_input_kernel (MGU_NO_FIRST_MFMA=1) and the oracle to the fp32 tolerance."""
    cfg = (cin, 2, 32, 2)
    x = torch.from_numpy(O.formula_normal("firstmfma/x", (2, cin, 37, 45), seed=cin)).to(cuda)
    m = make_unet(cfg, 7, cuda)
    ref = m(x)[0]
    a = m(x.contiguous(memory_format=torch.channels_last))[0]
    big = torch.zeros(2, cin, 41, 50, device=cuda)
    big[:, :, 3:40, 2:47] = x
    b = m(big[:, :, 3:40, 2:47])[0]
    assert torch.equal(ref, a) and torch.equal(ref, b)
    with torch.no_grad():
        olg = O.unet_forward(O.make_unet_params(*cfg, seed=7), x.cpu(), cfg[3])[0]
    assert maxdiff(ref, olg.numpy()) <= 2e-5
    monkeypatch.setenv("MGU_NO_FIRST_MFMA", "1")   # read by mgu_create: a new model = a new context
    v = make_unet(cfg, 7, cuda)(x)[0]
    assert maxdiff(ref, v.cpu().numpy()) <= 2e-5


def test_c1_sampled_logits_and_checksums(cuda, golden):
    g = golden["c1"]
    m = make_unet((1, 2, 32, 4), 0, cuda)
    x = torch.from_numpy(O.formula_normal("c1/x", (1, 1, 256, 256), seed=0)).to(cuda)
    lg, sk, ft = m(x)
    got = lg.contiguous().reshape(-1)[torch.from_numpy(g["idx"]).to(cuda)]
    assert maxdiff(got, g["logits"]) <= TOL
    sums = np.array([[float(t.double().sum()), float(t.double().abs().sum())] for t in [lg] + sk + ft])
    assert np.allclose(sums, g["sums"], rtol=1e-5, atol=1e-2)


def test_c2_full_forward_batch8(cuda, golden):
    """BASELINE config 2: b=8 3x512x512 fp32, U-Net + patch-mean + patch-graph GAT, one launch sequence."""
    g = golden["c2"]
    unet = make_unet((3, 2, 32, 4), 0, cuda)
    gat = make_gat((32, 128, 64, 4), cuda)
    model = mgunet.MinGraphUNet(unet, gat, 16).eval()
    x = torch.cat([torch.from_numpy(O.formula_normal(f"c2/x/{b}", (1, 3, 512, 512), seed=1)) for b in range(8)]).to(cuda)
    lg, sk, ft, emb = model(x)
    assert tuple(lg.shape) == (8, 2, 512, 512) and tuple(emb.shape) == (8 * 1024, 64)
    worst = 0.0
    for b in range(8):
        got = lg[b].contiguous().reshape(-1)[torch.from_numpy(g[f"idx_{b}"]).to(cuda)]
        worst = max(worst, maxdiff(got, g[f"logits_{b}"]))
        gg = emb[b * 1024:(b + 1) * 1024].reshape(-1)[torch.from_numpy(g[f"gidx_{b}"]).to(cuda)]
        assert maxdiff(gg, g[f"gat_{b}"]) <= TOL, f"gat image {b}"
        tens = [lg[b]] + [s[b] for s in sk] + [f[b] for f in ft] + [None, emb[b * 1024:(b + 1) * 1024]]
        for j, t in enumerate(tens):
            if t is None:
                continue
            s = np.array([float(t.double().sum()), float(t.double().abs().sum())])
            assert np.allclose(s, g["sums"][b, j], rtol=2e-5, atol=5e-2), (b, j, s, g["sums"][b, j])
    assert worst <= TOL
    print(f"[c2] worst sampled |logit diff| over 8 images = {worst:.2e}")
    # size-independent property: an image's result does not depend on its batch neighbours (bit-exact)
    lg1, _, _, emb1 = model(x[3:4])
    assert torch.equal(lg1[0], lg[3]) and torch.equal(emb1, emb[3 * 1024:4 * 1024])
    # eval loop (segmentation_performance.py:141): argmax on the GPU == torch.argmax
    _, pred = mgunet.segment_batch(unet, x[:2])
    assert torch.equal(pred, torch.argmax(lg[:2], dim=1))


def test_c4_unet_1024_and_stress_graph(cuda, golden):
    g = golden["c4"]
    unet = make_unet((3, 2, 32, 4), 0, cuda)
    for b in (0, 31):
        x = torch.from_numpy(O.formula_normal(f"c4/x/{b}", (1, 3, 1024, 1024), seed=2)).to(cuda)
        lg, _, _ = unet(x)
        got = lg.contiguous().reshape(-1)[torch.from_numpy(g[f"idx_{b}"]).to(cuda)]
        assert maxdiff(got, g[f"logits_{b}"]) <= TOL
    # synthetic superpixel-like graph: 2048 nodes, in-degree exactly 8
    N, deg = 2048, 8
    u = O.formula_uniform("c4/src", (N * deg,), 0.0, 1.0, 3).astype(np.float64)
    src = np.minimum((u * N).astype(np.int64), N - 1)
    ei = torch.from_numpy(np.stack([src, np.repeat(np.arange(N, dtype=np.int64), deg)])).to(cuda)
    X = torch.from_numpy(O.formula_normal("c4/X", (N, 64), seed=3)).to(cuda)
    gat = make_gat((64, 128, 64, 4), cuda)
    y = gat(X, ei)
    assert maxdiff(y, g["gat_out"]) <= TOL
    # the same graph replicated 4x as a block-diagonal batch must reproduce the single-graph result
    eib = torch.cat([ei + k * N for k in range(4)], 1)
    yb = gat(X.repeat(4, 1), eib, graph_ptr=torch.arange(5) * N)
    assert torch.equal(yb[:N], y) and torch.equal(yb[3 * N:], y)


def test_gat_small_cases(cuda, golden):
    g = golden["gat_small"]
    e10 = torch.from_numpy(g["edge10"]).to(cuda)
    X10 = torch.from_numpy(O.formula_normal("gat/x10", (10, 32), seed=3)).to(cuda)
    cases = [("g10", (32, 64, 16, 4), X10, e10, 1.0, 1, 1.0),
             ("iso", (32, 64, 16, 4), torch.from_numpy(O.formula_normal("gat/x12", (12, 32), seed=3)).to(cuda),
              torch.from_numpy(g["edge_iso"]).to(cuda), 1.0, 1, 1.0),
             ("wide", (32, 64, 16, 2), (torch.from_numpy(O.formula_normal("gat/xw", (10, 32), seed=4)) * 4.0).to(cuda), e10, 3.0, 1, 1.0),
             ("mid", (32, 64, 16, 4), (torch.from_numpy(O.formula_normal("gat/xm", (10, 32), seed=6)) * 2.0).to(cuda), e10, 1.5, 1, 1.0),
             ("l2h1", (32, 24, 8, 1), X10, e10, 1.0, 2, 1.0)]
    for tag, cfg, X, ei, scale, layers, _ in cases:
        net = make_gat(cfg, cuda, seed=3, scale=scale, layers=layers)
        y = net(X, ei)
        ref = g[tag + "_out"]
        tol = TOL * max(1.0, float(np.abs(ref).max()))
        assert maxdiff(y, ref) <= tol, (tag, maxdiff(y, ref))
    iso = make_gat((32, 64, 16, 4), cuda, seed=3)(torch.from_numpy(O.formula_normal("gat/x12", (12, 32), seed=3)).to(cuda),
                                                   torch.from_numpy(g["edge_iso"]).to(cuda))
    assert torch.all(iso[10:] == 0)  # no in-edges -> exactly elu(0) = 0
    # concat=True layer (graph_attention.py:153-155)
    mh = mgunet.MultiHeadGATLayer(32, 64, 4, 0.1, 0.2, concat=True)
    sd = {}
    for h in range(4):
        for nm, shp in ((f"heads.{h}.W.weight", (16, 32)), (f"heads.{h}.a.weight", (1, 32))):
            a = 1.414 * float(np.sqrt(6.0 / (shp[0] + shp[1])))
            sd[nm] = torch.from_numpy(O.formula_uniform("mhc/" + nm, shp, -a, a, 5))
    mh.load_state_dict(sd)
    yc = mh.to(cuda).eval()(X10, e10)
    assert maxdiff(yc, g["mhc_out"]) <= TOL
    # empty graph: every row is elu(0) = 0
    ye = make_gat((32, 64, 16, 4), cuda, seed=3)(X10, torch.zeros(2, 0, dtype=torch.long, device=cuda))
    assert tuple(ye.shape) == (10, 16) and torch.all(ye == 0)


def test_gat_non_multiple_of_4_input_width_and_errors(cuda):
    X = torch.from_numpy(O.formula_normal("gat/x22", (30, 22), seed=9))
    ei = torch.from_numpy(O.patch_graph_edges(80, 96, 16))
    p = O.make_gat_params(22, 16, 8, 2, 1, seed=9)
    with torch.no_grad():
        ref = O.gat_network_forward(p, X, ei, 2)
    net = mgunet.GATNetwork(22, 16, 8, 2)
    net.load_state_dict(p)
    y = net.to(cuda).eval()(X.to(cuda), ei.to(cuda))
    assert maxdiff(y, ref.numpy()) <= TOL
    with pytest.raises(IndexError):
        net(X.to(cuda), torch.tensor([[0], [30]], device=cuda))
    # train mode (dropout 0.1 by default, graph_attention.py:97, :160) runs since round 4: its own masks, a different result, and
    # eval mode is unchanged afterwards (parity of the train mode itself: tests/test_gpu_gat_train.py)
    yt = net.train()(X.to(cuda), ei.to(cuda)).detach()
    assert bool(torch.isfinite(yt).all()) and not torch.equal(yt, y)
    assert torch.equal(net.eval()(X.to(cuda), ei.to(cuda)), y)


def test_patch_mean_kernel(cuda, golden):
    img = torch.from_numpy(O.formula_normal("graph/img", (5, 37, 45), seed=2))
    x = torch.zeros(1, 8, 37, 45)
    x[0, :5] = img
    got = mgunet.PatchGraphConstructor(16).patch_mean_features(x.to(cuda))
    assert maxdiff(got[:, :5], golden["patch_graph"]["patches_37x45_mean"]) <= 1e-5
    assert torch.all(got[:, 5:] == 0)


def test_errors_surface_as_python_exceptions(cuda):
    m = make_unet((3, 2, 8, 2), 5, cuda)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 2, 2, device=cuda))           # too small for depth 2
    with pytest.raises(TypeError):
        m(torch.zeros(1, 3, 32, 32, device=cuda, dtype=torch.float16))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 32, 32, device=cuda))          # wrong channel count


def test_weights_are_repacked_after_update(cuda):
    cfg = (3, 2, 8, 2)
    m = make_unet(cfg, 5, cuda)
    x = torch.from_numpy(O.formula_normal("upd/x", (1, 3, 32, 32), seed=5)).to(cuda)
    a = m(x)[0].clone()
    p2 = O.make_unet_params(*cfg, seed=6)
    m.load_state_dict(p2)
    with torch.no_grad():
        ref = O.unet_forward(p2, x.cpu(), 2)[0]
    b = m(x)[0]
    assert not torch.equal(a, b) and maxdiff(b, ref.numpy()) <= TOL


def test_full_size_properties_c3_c4_shapes(cuda):
    """BASELINE configs[2] per-GPU shard and configs[3] at FULL size (no oracle at this size): size-independent
    properties -- an image's logits do not depend on its batch neighbours (bit-exact), the batch-of-32 1024^2
    forward equals the golden samples of images 0 and 31, outputs are finite, argmax loop == torch.argmax."""
    unet = make_unet((3, 2, 32, 4), 0, cuda)
    # configs[3]: batch 32 x 3 x 1024 x 1024 in ONE call (34 M pixels; M = B*H*W close to the int32 row limit / 64)
    x = torch.empty((32, 3, 1024, 1024), device=cuda)
    for b in (0, 31):
        x[b] = torch.from_numpy(O.formula_normal(f"c4/x/{b}", (1, 3, 1024, 1024), seed=2))[0].to(cuda)
    gen = torch.Generator(device=cuda)
    gen.manual_seed(7)
    x[1:31] = torch.randn((30, 3, 1024, 1024), device=cuda, generator=gen)
    lg, sk, ft = unet(x)
    assert tuple(lg.shape) == (32, 2, 1024, 1024) and bool(torch.isfinite(lg).all())
    import numpy as np
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "c4.npz"))
    for b in (0, 31):
        got = lg[b].contiguous().reshape(-1)[torch.from_numpy(g[f"idx_{b}"]).to(cuda)]
        assert maxdiff(got, g[f"logits_{b}"]) <= TOL
    alone = unet(x[17:18])[0]
    assert torch.equal(alone[0], lg[17])
    del lg, sk, ft, alone, x
    torch.cuda.empty_cache()
    # configs[2] shard: 8 images of 512^2 inside a batch of 64 (one GPU's share is 8; here all 64 on one GPU)
    xb = torch.randn((64, 3, 512, 512), device=cuda, generator=gen)
    lgb = unet(xb)[0]
    shard = unet(xb[24:32])[0]
    assert torch.equal(shard, lgb[24:32]) and bool(torch.isfinite(lgb).all())
    _, pred = mgunet.segment_batch(unet, xb[:4])
    assert torch.equal(pred, torch.argmax(lgb[:4], dim=1))


def test_requested_patch_means_equal_standalone_kernel(cuda):
    """mgu_unet_request_patch_mean: the fused head + patch-mean pass must give the same logits and node features as the
    stand-alone kernels (final 1x1 conv, mgu_patch_mean), on a size that is not a multiple of the patch."""
    import mgunet
    from mgunet import _lib
    from mgunet.patch_graph import PatchGraphConstructor
    cfg = (3, 2, 16, 2)
    p = O.make_unet_params(*cfg, seed=5)
    x = torch.from_numpy(O.formula_normal("pm/x", (2, 3, 40, 56), seed=5)).to(cuda)
    model = mgunet.UNet(*cfg)
    model.load_state_dict(p)
    model = model.to(cuda).eval()
    pg = PatchGraphConstructor(16)
    with torch.no_grad():
        lg0, _, f0 = model(x)
        X0 = pg.patch_mean_features(f0[0])
        X1 = torch.full_like(X0, -7.0)
        ctx = model._context(cuda)
        _lib.check(_lib.lib().mgu_unet_request_patch_mean(ctx.handle, 16, X1.data_ptr()), ctx.handle)
        lg1, _, f1 = model(x)
        lg2, _, _ = model(x)          # the request is one-shot: this forward must not touch X1 again
    assert torch.equal(f0[0], f1[0])
    assert float((X1 - X0).abs().max()) <= 1e-6
    assert float((lg1 - lg0).abs().max()) <= 1e-5 and torch.equal(lg2, lg0)
    olg = O.unet_forward({k: v.clone() for k, v in p.items()}, x.cpu(), cfg[3])[0]
    assert float((lg1.cpu() - olg).abs().max()) <= 1e-3

