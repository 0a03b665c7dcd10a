"""BASELINE configs[2] precision mode: bf16 storage + fp32 accumulate (inference).  The yardstick is PINNED by the reference
since round 4: tests/golden/bf16_reference.npz holds the reference U-Net's own bfloat16 CPU run at the formula weights (2.6 - 2.9 %
of max|logit| max-abs / 0.2 - 0.32 % mean-abs away from its fp32 run, 99.35 % argmax agreement on the 512^2 image), and
test_bf16_deviation_pinned_by_the_references_own_bf16_run holds the HIP mode to 1.25 x that.  (The reference's GAT is fp32-only,
SURVEY App. A: the graph branch stays fp32 on the bf16 features.)  The older, tighter budget below is kept as well.

Two yardsticks, two tolerances:
  * against the fp32 results (what storing 23 activations per pixel in bf16 costs): every stored activation carries a
    relative rounding error up to 2^-9 = 0.195 %, the 18 conv layers + 4 transposed convs in series add theirs
    roughly in quadrature (sqrt(22) * 0.2 % ~ 0.9 % rms of the activation scale at the head), and the maximum over the
    4.2 M logits of a batch sits 4-5 sigma out.  Measured on MI355X (r02): max-abs 1.86 % of max|logit|, mean-abs
    0.21 %, 99.9th percentile below 1 %, argmax agreement 99.53 %.  Tolerance: max-abs <= 2.5 %, 99.9th percentile
    <= 1 %, mean-abs <= 0.3 % of max|logit|, argmax agreement >= 99 %.  A max-abs bound of 1 % is not reachable by ANY
    bf16-storage implementation of this network (the reference's own bf16 run is at 1.4 %).
  * against the oracle's bf16-storage restatement, segment by segment from the HIP path's own stored tensors (what the
    HIP bf16 KERNELS may add on top of the storage format: only the fp32 accumulation order, which now and then flips
    the bf16 rounding of a stored activation by one ulp): every exposed tensor <= 1 % of its max (the 1e-2 bar),
    >= 90 % of the values bit-equal, logits == fp32 head of the stored feature to 1e-4."""
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
    p999 = float(np.quantile(d.reshape(-1)[:: max(1, d.size // 2_000_000)], 0.999))
    print(f"    99.9th percentile {p999:.4f} ({p999/scale*100:.2f} %)")
    assert d.max() <= 2.5e-2 * scale and d.mean() <= 3e-3 * scale
    if d.size >= 100_000:
        assert p999 <= 1e-2 * scale
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


@pytest.mark.parametrize("tag,cfg,shape,first_fp32", [
    ("b", (3, 3, 8, 2), (2, 3, 37, 45), False), ("c", (3, 2, 8, 3), (2, 3, 64, 48), False),
    ("f16", (3, 2, 16, 3), (2, 3, 96, 80), True), ("c2", (3, 2, 32, 4), (2, 3, 512, 512), True)])
def test_bf16_kernels_vs_storage_emulation(cuda, tag, cfg, shape, first_fp32):
    """Kernel-level bf16 parity, the <= 1e-2 * max bar: every tensor the bf16 forward exposes (skips, decoder features,
    logits) is recomputed by the oracle FROM THE HIP PATH'S OWN exposed predecessors with the same storage roundings
    (oracle.conv_block_bf16_storage / decoder_block_bf16_storage: two to five layers per segment) and must agree to
    1e-2 of the tensor's max with >= 90 % of the bf16 values bit-equal; the fp32 logits must equal the oracle head on the
    HIP feature to 1e-4.  (The whole network against the whole emulation is printed and held to the storage budget only:
    see the note in oracle.unet_forward_bf16_storage.)"""
    depth = cfg[3]
    m = build(cfg, 21, cuda, torch.bfloat16)
    p = O.make_unet_params(*cfg, seed=21)
    x = torch.from_numpy(O.formula_normal(f"bf16emu/{tag}/x", shape, seed=21))
    lg, sk, ft = m(x.to(cuda))
    sk = [t.float().cpu() for t in sk]
    ft = [t.float().cpu() for t in ft]
    F = torch.nn.functional
    worst_all = 0.0

    def seg(name, got, ref):
        nonlocal worst_all
        same = float((got == ref).float().mean())
        worst = float((got - ref).abs().max() / ref.abs().max())
        worst_all = max(worst_all, worst)
        print(f"    [{tag}] {name}: bit-equal {same*100:.2f} %, max-abs {worst*100:.3f} % of max")
        assert same >= 0.90 and worst <= 1e-2, name

    with torch.no_grad():
        seg("skip0 <- input", sk[0], O.conv_block_bf16_storage(p, "encoder.encoder_blocks.0.", O._bf16(x), first_fp32))
        for i in range(1, depth):
            seg(f"skip{i} <- skip{i-1}", sk[i], O.conv_block_bf16_storage(p, f"encoder.encoder_blocks.{i}.", F.max_pool2d(sk[i - 1], 2, 2)))
        bott = O.conv_block_bf16_storage(p, "encoder.bottleneck.", F.max_pool2d(sk[-1], 2, 2))
        # the bottleneck is not exposed: the deepest decoder feature is recomputed from the deepest skip (5 layers)
        seg(f"feat{depth-1} <- skip{depth-1}", ft[depth - 1], O.decoder_block_bf16_storage(p, 0, bott, sk[depth - 1]))
        for i in range(depth - 2, -1, -1):
            seg(f"feat{i} <- feat{i+1}, skip{i}", ft[i], O.decoder_block_bf16_storage(p, depth - 1 - i, ft[i + 1], sk[i]))
        head = F.conv2d(ft[0], p["decoder.final_conv.weight"], p["decoder.final_conv.bias"])
        d = float((lg.cpu() - head).abs().max())
        print(f"    [{tag}] logits <- feat0: max-abs {d:.2e} (fp32 head on the bf16 feature)")
        assert d <= 1e-4 * max(1.0, float(head.abs().max()))
        elg = O.unet_forward_bf16_storage(p, x, depth=depth, first_fp32=first_fp32)[0]
    e = (lg.cpu() - elg).abs()
    print(f"[bf16 {tag}] worst segment {worst_all*100:.3f} % of its max; whole network vs whole emulation: max-abs "
          f"{float(e.max()/elg.abs().max())*100:.2f} %, mean-abs {float(e.mean()/elg.abs().max())*100:.3f} % of max|logit|")
    assert float(e.max()) <= 2.5e-2 * float(elg.abs().max())


@pytest.mark.parametrize("tag,cfg,shape,xname,xseed,pseed", [
    ("b", (3, 3, 8, 2), (2, 3, 37, 45), "tiny/b/x", 11, 11), ("c", (3, 2, 8, 3), (2, 3, 64, 48), "tiny/c/x", 11, 11),
    ("c2_0", (3, 2, 32, 4), (1, 3, 512, 512), "c2/x/0", 1, 0)])
def test_bf16_deviation_pinned_by_the_references_own_bf16_run(cuda, golden, tag, cfg, shape, xname, xseed, pseed):
    """The bf16 tolerance pinned by the REFERENCE: tests/golden/bf16_reference.npz (oracle/make_golden.py gen_bf16ref) holds what the
    reference's own U-Net gives in bfloat16 on the CPU at these weights and inputs -- its deviation from its own fp32 run (2.6 - 2.9 %
    of max|logit| max-abs, 0.2 - 0.32 % mean-abs, 99.35 - 100 % argmax agreement) and sampled logits of both runs.  The HIP bf16-storage
    mode (bf16 weights and activations, fp32 accumulation) must deviate from the reference's fp32 logits by at most 1.25 x what the
    reference's bf16 run does: full-tensor statistics against the fp32 HIP path (itself within 1e-5 of the reference) and the sampled
    logits against the reference's own fp32 samples."""
    g = golden["bf16_reference"]
    ref_max, ref_mean, ref_p999, ref_agree, scale = [float(v) for v in g[f"{tag}_stats"]]
    x = torch.from_numpy(O.formula_normal(xname, shape, seed=xseed)).to(cuda)
    lb = build(cfg, pseed, cuda, torch.bfloat16)(x)[0]
    lf = build(cfg, pseed, cuda, torch.float32)(x)[0]
    idx = torch.from_numpy(g[f"{tag}_idx"]).to(cuda)
    fp32_s = torch.from_numpy(g[f"{tag}_fp32"]).to(cuda)
    assert float((lf.contiguous().reshape(-1)[idx] - fp32_s).abs().max()) <= 1e-4 * scale      # same network, same inputs as the fixture
    d = (lb - lf).abs()
    agree = float((lb.argmax(1) == lf.argmax(1)).float().mean())
    flat = d.reshape(-1)
    p999 = float(torch.quantile(flat[:: max(1, flat.numel() // 2_000_000)].float(), 0.999))
    print(f"[bf16 vs reference-bf16 {tag}] HIP max-abs {float(d.max()):.4e} / mean {float(d.mean()):.3e} / p99.9 {p999:.3e} / agree {agree*100:.2f} %   "
          f"reference bf16: {ref_max:.4e} / {ref_mean:.3e} / {ref_p999:.3e} / {ref_agree*100:.2f} %")
    assert float(d.max()) <= 1.25 * ref_max and float(d.mean()) <= 1.25 * ref_mean and p999 <= 1.25 * ref_p999
    assert agree >= ref_agree - 0.005
    # the sampled logits: HIP bf16 against the reference's fp32 samples, next to the reference's bf16 samples against the same
    hip_s = (lb.contiguous().reshape(-1)[idx] - fp32_s).abs()
    ref_s = (torch.from_numpy(g[f"{tag}_bf16"]).to(cuda) - fp32_s).abs()
    assert float(hip_s.max()) <= 1.25 * max(float(ref_s.max()), ref_p999) and float(hip_s.mean()) <= 1.25 * float(ref_s.mean())


def test_bf16_full_batch64_properties(cuda):
    """BASELINE configs[2] at its FULL global batch (64 x 3 x 512 x 512) in the bf16 mode it is quoted in, all 64 images
    on one GPU (no oracle finishes at this size): an image's logits do not depend on its batch neighbours (bit-exact
    against the 8-image shard and the single image), two runs are bit-identical, outputs are finite, and the batch
    statistics of the deviation from the fp32 HIP path stay inside the bf16-storage budget above."""
    cfg = (3, 2, 32, 4)
    mb = build(cfg, 0, cuda, torch.bfloat16)
    gen = torch.Generator(device=cuda)
    gen.manual_seed(11)
    xb = torch.randn((64, 3, 512, 512), device=cuda, generator=gen)
    lgb = mb(xb)[0]
    assert tuple(lgb.shape) == (64, 2, 512, 512) and bool(torch.isfinite(lgb).all())
    assert torch.equal(mb(xb)[0], lgb)
    assert torch.equal(mb(xb[40:48])[0], lgb[40:48])
    assert torch.equal(mb(xb[63:64])[0][0], lgb[63])
    mf = build(cfg, 0, cuda, torch.float32)
    lf = mf(xb[40:48])[0]
    d = (lgb[40:48] - lf).abs()
    scale = float(lf.abs().max())
    agree = float((lgb[40:48].argmax(1) == lf.argmax(1)).float().mean())
    print(f"[bf16 b64] images 40-47 vs fp32 HIP: max-abs {float(d.max())/scale*100:.2f} %, mean-abs {float(d.mean())/scale*100:.3f} % of "
          f"max|logit| {scale:.2f}, argmax agreement {agree*100:.2f} %")
    assert float(d.max()) <= 2.5e-2 * scale and float(d.mean()) <= 3e-3 * scale and agree >= 0.99


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
    assert worst <= 2.5e-2 * 8.0
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


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_forward_is_bitwise_repeatable_with_hand_counted_waits(cuda, dtype, monkeypatch):
    """The halo kernel's weight tiles arrive by LDS-DMA and are waited for with hand-counted vmcnt values behind raw barriers
    (csrc/igemm.hip, GLDS): a wait that under-counts shows up as rare wrong tiles that come and go with timing.  60 forwards of the
    configs[2] shard (8 x 3 x 512^2; fp32: the same kernel with MGU_NO_WINOGRAD=1) beside a second stream streaming through HBM must give
    the same bytes every time."""
    if dtype == torch.float32:
        monkeypatch.setenv("MGU_NO_WINOGRAD", "1")   # fp32 layers on conv3x3_halo_kernel<float>
    m = build((3, 2, 32, 4), 0, cuda, dtype)
    x = torch.from_numpy(O.formula_normal("repeat/x", (8, 3, 512, 512), seed=3)).to(cuda)
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device=cuda)
    first = None
    with torch.no_grad():
        for i in range(60 if dtype == torch.bfloat16 else 12):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    big.mul_(1.0001)
            lg = m(x)[0]
            if first is None:
                first = lg.clone()
            else:
                assert torch.equal(lg, first), f"forward {i} differs from the first"
    torch.cuda.synchronize()
