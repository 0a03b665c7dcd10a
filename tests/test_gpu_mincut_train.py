"""Training through the MinCut stage on the HIP path: `loss.backward()` of `mgunet.MinCutRefinement` runs mgu_ncut_backward, the MLP
predictor's Linear layers run their backward on the library's 1x1-convolution dgrad / wgrad kernels, the GNN predictor and the patch
GAT run mgu_gat_layer_backward.  Reference: the gradients the reference's own MinCutRefinement + predictor produced under torch
autograd (tests/golden/mincut_grad.npz, oracle/make_golden.py gen_mincutgrad).  Tolerance: 2e-5 of max|gradient| (fp32 sums in a
different order; the float64 analytic gradient is the second yardstick for the loss kernel alone)."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O

pytestmark = pytest.mark.gpu

TOL = 2e-5


def build_predictor(cuda, D, K, hidden, use_gnn, heads, params):
    m = mgunet.PatchSegmentPredictor(D, K, hidden_dim=hidden, use_gnn=use_gnn, num_gnn_layers=1, num_heads=heads)
    m.load_state_dict(params, strict=True)
    return m.to(cuda).eval()   # eval: the GNN predictor's dropout(0.1) is torch-RNG noise in train mode


def close(got, ref, what, tol=TOL):
    got = got.detach().cpu().numpy().astype(np.float64)
    d = np.abs(got - ref).max()
    assert d <= tol * max(1.0, np.abs(ref).max()), (what, d, np.abs(ref).max())


@pytest.mark.parametrize("tag", list(O.MINCUTGRAD_CASES))
def test_mincut_backward_vs_reference_fixture(cuda, golden, tag):
    g = golden["mincut_grad"]
    ei, X, R, p, K, use_gnn, heads, shift = O.mincutgrad_inputs(tag)
    hidden = O.MINCUTGRAD_CASES[tag][4]
    pred = build_predictor(cuda, X.shape[1], K, hidden, use_gnn, heads, p)
    net = pred if shift is None else (lambda x, e: pred(x, e) + shift.to(cuda))
    mc = mgunet.MinCutRefinement()
    Xd = X.to(cuda).requires_grad_(True)
    eid = ei.to(cuda)
    loss, soft = mc(Xd, eid, K, net)
    assert loss.requires_grad and soft.requires_grad
    assert abs(float(loss.detach()) - float(g[tag + "_loss"])) <= 2e-5 * max(1.0, float(g[tag + "_loss"]))
    (loss + 0.05 * (soft * R.to(cuda)).sum()).backward()
    close(Xd.grad, g[tag + "_dX"], "dX")
    for k, v in pred.named_parameters():
        assert v.grad is not None, k
        close(v.grad, g[f"{tag}_d_{k}"], k)
    # normalized_cut_loss called directly, the soft assignments as the leaf (:55-160); float64 analytic gradient as yardstick
    lg = O.segment_predictor_forward(p, X, ei, use_gnn, heads)
    P = torch.softmax(lg if shift is None else lg + shift, dim=1).to(cuda).requires_grad_(True)
    Xq = X.to(cuda).requires_grad_(True)
    (2.5 * mc.normalized_cut_loss(Xq, eid, P, K)).backward()
    close(P.grad, g[tag + "_direct_dP"], "direct dP")
    close(Xq.grad, g[tag + "_direct_dX"], "direct dX")
    close(P.grad, g[tag + "_direct_dP64"], "direct dP vs float64", tol=1e-5)
    close(Xq.grad, g[tag + "_direct_dX64"], "direct dX vs float64", tol=1e-5)
    # a second backward gives the same bytes: both gradients are gathers, no atomics
    P2 = P.detach().clone().requires_grad_(True)
    X2 = X.to(cuda).requires_grad_(True)
    (2.5 * mc.normalized_cut_loss(X2, eid, P2, K)).backward()
    assert torch.equal(P2.grad, P.grad) and torch.equal(X2.grad, Xq.grad)


def test_patch_gat_to_mincut_chain_trains_like_the_reference(cuda, golden):
    """patch GAT -> MinCut with the GNN predictor (train_end_to_end.py:332-356), parameters of both in one SGD optimizer
    (:219-226): first-step gradients, three steps of losses and the final parameters against the reference's own run."""
    g = golden["mincut_grad"]
    ei = torch.from_numpy(O.patch_graph_edges(128, 128, 16)).to(cuda)
    X0 = (torch.from_numpy(O.formula_normal("mincutgrad/chain/x", (64, 16), seed=21)) * 0.5).to(cuda)
    gat = mgunet.GATNetwork(16, 8, 16, 2, num_gat_layers=1, dropout_rate=0.0)
    gat.load_state_dict(O.make_gat_params(16, 8, 16, 2, 1, seed=22))
    gat = gat.to(cuda).train()
    pred = build_predictor(cuda, 16, 2, 8, True, 2, O.make_segment_predictor_params(16, 2, 8, True, 2, seed=23))
    mc = mgunet.MinCutRefinement()
    opt = torch.optim.SGD(list(gat.parameters()) + list(pred.parameters()), lr=0.2)
    losses = []
    for step in range(4):
        opt.zero_grad()
        loss, soft = mc(gat(X0, ei), ei, 2, pred)
        losses.append(float(loss.detach()))
        if step < 3:
            loss.backward()
            if step == 0:
                for k, v in gat.named_parameters():
                    close(v.grad, g[f"chain_d_gat.{k}"], "gat." + k)
                for k, v in pred.named_parameters():
                    close(v.grad, g[f"chain_d_{k}"], k)
            opt.step()
    assert np.abs(np.array(losses) - g["chain_losses"]).max() <= 2e-5
    assert losses[3] < losses[0]
    for k, v in gat.state_dict().items():
        close(v, g[f"chain_final_gat.{k}"], "final gat." + k)
    for k, v in pred.state_dict().items():
        close(v, g[f"chain_final_{k}"], "final " + k)


def test_mlp_predictor_linear_backward_vs_torch(cuda):
    """The Linear -> ReLU -> Linear predictor alone against torch's own autograd on the same weights (float64), widths that are
    not multiples of the MFMA tile and K = 3 outputs (a padded 4-column buffer)."""
    torch.manual_seed(3)
    pred = mgunet.PatchSegmentPredictor(24, 3, hidden_dim=40, use_gnn=False).to(cuda)
    x = torch.randn(301, 24, device=cuda)
    r = torch.randn(301, 3, device=cuda)
    xr = x.clone().requires_grad_(True)
    (pred(xr) * r).sum().backward()
    ref = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.ReLU(), torch.nn.Linear(40, 3)).double()
    ref.load_state_dict({k[len("mlp_predictor."):]: v.double().cpu() for k, v in pred.state_dict().items()})
    x64 = x.double().cpu().requires_grad_(True)
    (ref(x64) * r.double().cpu()).sum().backward()
    close(xr.grad, x64.grad.numpy(), "dx")
    for (k, v), (_, w) in zip(pred.named_parameters(), ref.named_parameters()):
        close(v.grad, w.grad.numpy(), k)


def test_ncut_backward_interface(cuda):
    mc = mgunet.MinCutRefinement()
    ei = torch.tensor([[0, 1, 2], [1, 0, 0]], device=cuda)
    # no gradient is recorded under no_grad, and a graph without edges gives zero gradients (every segment is skipped)
    X = torch.randn(4, 8, device=cuda, requires_grad=True)
    P = torch.full((4, 2), 0.5, device=cuda, requires_grad=True)
    with torch.no_grad():
        assert not mc.normalized_cut_loss(X, ei, P, 2).requires_grad
    l = mc.normalized_cut_loss(X, torch.zeros((2, 0), dtype=torch.int64, device=cuda), P, 2)
    l.backward()
    assert float(X.grad.abs().max()) == 0.0 and float(P.grad.abs().max()) == 0.0
    # feature rows wider than the kernel's register budget are refused, not truncated
    Xw = torch.randn(4, 1028, device=cuda, requires_grad=True)
    with pytest.raises(ValueError, match="D <= 1024"):
        mc.normalized_cut_loss(Xw, ei, P, 2)
    with torch.no_grad():
        assert torch.isfinite(mc.normalized_cut_loss(Xw, ei, P, 2))   # the forward alone takes any width


def test_graph_branch_losses_of_the_e2e_step_vs_oracle_composition(cuda):
    """The graph-branch half of one iteration of scripts/train_end_to_end.py:300-356, 438-479 for a batch of two images: per image
    patch GAT -> L_feature (FeatureConsistencyLoss, called with the batch dimension the script forgets: SURVEY appendix A) and
    MinCut -> L_partition; total = 0.1 mean(L_feature) + 0.5 mean(L_partition) (the script's default weights, :462-465);
    total.backward().  The patch GAT output feeds three consumers (feature loss, predictor, edge weights of the cut): its
    gradient is their sum.  Compared with the oracle's restatement of the same composition under torch autograd (CPU, fp32)."""
    B, Dp, K = 2, 32, 2
    ei = torch.from_numpy(O.patch_graph_edges(128, 128, 16))
    Np = 64
    gp = O.make_gat_params(Dp, 16, Dp, 2, 1, seed=31)
    pp = O.make_segment_predictor_params(Dp, K, 16, True, 2, seed=32)
    feats = [torch.from_numpy(O.formula_normal(f"e2e/{b}/patch", (Np, Dp), seed=33 + b)) * 0.4 for b in range(B)]   # :326 placeholder
    funet = [torch.from_numpy(O.formula_normal(f"e2e/{b}/funet", (Np, Dp), seed=43 + b)) * 0.4 for b in range(B)]   # :338 placeholder
    ylab = [torch.from_numpy((O.formula_uniform(f"e2e/{b}/y", (Np,), 0.0, 1.0, 53 + b) > 0.5).astype(np.int64)) for b in range(B)]  # :342

    # oracle composition
    q = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    r = {k: v.clone().requires_grad_(True) for k, v in pp.items()}
    lf, lp = 0.0, 0.0
    for b in range(B):
        h = O.gat_network_forward(q, feats[b], ei, 2, 1)
        lf = lf + O.feature_consistency_loss(funet[b][None], h[None], ylab[b][None])
        l, _, _ = O.mincut_forward(h, ei, K, O.segment_predictor_forward(r, h, ei, True, 2))
        lp = lp + l
    tot = 0.1 * lf / B + 0.5 * lp / B
    tot.backward()

    # HIP path
    gat = mgunet.GATNetwork(Dp, 16, Dp, 2, num_gat_layers=1, dropout_rate=0.0)
    gat.load_state_dict(gp)
    gat = gat.to(cuda).train()
    pred = build_predictor(cuda, Dp, K, 16, True, 2, pp)
    mc, fl = mgunet.MinCutRefinement(), mgunet.FeatureConsistencyLoss(margin=1.0)
    eid = ei.to(cuda)
    hlf, hlp = 0.0, 0.0
    for b in range(B):
        h = gat(feats[b].to(cuda), eid)
        hlf = hlf + fl(funet[b].to(cuda)[None], h[None], ylab[b].to(cuda)[None])
        l, soft = mc(h, eid, K, pred)
        hlp = hlp + l
    htot = 0.1 * hlf / B + 0.5 * hlp / B
    htot.backward()
    assert abs(float(hlf.detach()) - float(lf.detach())) <= 2e-5 * max(1.0, float(lf.detach()))
    assert abs(float(hlp.detach()) - float(lp.detach())) <= 2e-5 * max(1.0, float(lp.detach()))
    for k, v in gat.named_parameters():
        close(v.grad, q[k].grad.numpy().astype(np.float64), "gat." + k)
    for k, v in pred.named_parameters():
        close(v.grad, r[k].grad.numpy().astype(np.float64), k)


def test_e2e_trainer_iteration_vs_oracle_and_torch_adam(cuda):
    """mgunet.E2ETrainer.step = one iteration of scripts/train_end_to_end.py:262-480: the U-Net half is Trainer's CE step, the graph
    half the composition checked above, and BOTH parameter sets take the script's single Adam(lr, weight_decay) step (:228).  Graph
    branch against the oracle composition + torch.optim.Adam on the CPU; U-Net half against a plain Trainer on the same inputs."""
    torch.manual_seed(0)
    B, Dp, K, H = 2, 32, 2, 64
    ei = torch.from_numpy(O.patch_graph_edges(H, H, 16))
    Np = (H // 16) ** 2
    gp = O.make_gat_params(Dp, 16, Dp, 2, 1, seed=61)
    pp = O.make_segment_predictor_params(Dp, K, 16, True, 2, seed=62)
    feats = [torch.from_numpy(O.formula_normal(f"e2et/{b}/patch", (Np, Dp), seed=63 + b)) * 0.4 for b in range(B)]
    funet = [torch.from_numpy(O.formula_normal(f"e2et/{b}/funet", (Np, Dp), seed=73 + b)) * 0.4 for b in range(B)]
    ylab = [torch.from_numpy((O.formula_uniform(f"e2et/{b}/y", (Np,), 0.0, 1.0, 83 + b) > 0.5).astype(np.int64)) for b in range(B)]
    lr, wd = 1e-3, 1e-4
    # oracle: graph branch under torch autograd + torch's Adam
    q = {k: torch.nn.Parameter(v.clone()) for k, v in gp.items()}
    r = {k: torch.nn.Parameter(v.clone()) for k, v in pp.items()}
    opt = torch.optim.Adam(list(q.values()) + list(r.values()), lr=lr, weight_decay=wd)
    lf = lp = 0.0
    for b in range(B):
        h = O.gat_network_forward(q, feats[b], ei, 2, 1)
        lf = lf + O.feature_consistency_loss(funet[b][None], h[None], ylab[b][None])
        l, _, _ = O.mincut_forward(h, ei, K, O.segment_predictor_forward(r, h, ei, True, 2))
        lp = lp + l
    lf, lp = lf / B, lp / B
    (0.1 * lf + 0.5 * lp).backward()
    g_ref = {**{"gat." + k: v.grad.clone() for k, v in q.items()}, **{k: v.grad.clone() for k, v in r.items()}}
    opt.step()
    # HIP path
    up = O.make_unet_params(3, 2, 8, 2, seed=9)
    images = torch.from_numpy(O.formula_normal("e2et/img", (B, 3, H, H), seed=5)).to(cuda)
    masks = torch.from_numpy((O.formula_uniform("e2et/mask", (B, H, H), 0.0, 1.0, 6) > 0.5).astype(np.int64)).to(cuda)

    def unet():
        m = mgunet.UNet(3, 2, 8, 2)
        m.load_state_dict(up)
        return m.to(cuda)
    plain = mgunet.Trainer(unet(), lr=lr, weight_decay=wd)
    loss_plain = plain.train_step(images, masks)
    gat = mgunet.GATNetwork(Dp, 16, Dp, 2, num_gat_layers=1, dropout_rate=0.0)
    gat.load_state_dict(gp)
    gat = gat.to(cuda).train()
    pred = build_predictor(cuda, Dp, K, 16, True, 2, pp)
    # the sub-models the loss does not reach but the script's optimizer holds (:223-226): zero gradient, weight decay only
    region_gat = mgunet.GATNetwork(Dp, 16, Dp, 2, num_gat_layers=1, dropout_rate=0.0)
    region_gat.load_state_dict(O.make_gat_params(Dp, 16, Dp, 2, 1, seed=64))
    region_gat = region_gat.to(cuda).train()
    region_before = {k: v.detach().clone() for k, v in region_gat.state_dict().items()}
    tr = mgunet.E2ETrainer(mgunet.Trainer(unet(), lr=lr, weight_decay=wd), gat, pred, mgunet.MinCutRefinement(),
                           mgunet.FeatureConsistencyLoss(margin=1.0), num_segments=K, extra_modules=[region_gat])
    out = tr.step(images, masks, [f.to(cuda) for f in feats], [f.to(cuda) for f in funet], [y.to(cuda) for y in ylab], ei.to(cuda))
    # torch.optim.Adam on a zero gradient with weight decay: g = wd * p, first step = -lr * sign(p) (where |wd p| >> eps)
    for k, v in region_gat.state_dict().items():
        p0 = region_before[k]
        want = p0 - lr * (wd * p0) / ((wd * p0).abs() + 1e-8)
        assert float((v - want).abs().max()) <= 1e-7, k
    assert set(out) == {"total", "l_unet_seg", "l_shape", "l_feature", "l_partition", "l_smooth"}
    assert abs(float(out["l_unet_seg"]) - float(loss_plain)) <= 1e-6                       # same step as the U-Net-only trainer ...
    assert torch.equal(tr.unet.flat, plain.flat)                                            # ... bit for bit (deterministic kernels)
    assert abs(float(out["l_feature"]) - float(lf.detach())) <= 2e-5 * max(1.0, float(lf.detach()))
    assert abs(float(out["l_partition"]) - float(lp.detach())) <= 2e-5 * max(1.0, float(lp.detach()))
    assert abs(float(out["total"]) - (float(loss_plain) + 0.1 * float(lf.detach()) + 0.5 * float(lp.detach()))) <= 1e-4
    # Adam's first step moves every parameter by ~lr * sign(g): compare where the gradient is not at rounding level
    new = {**{"gat." + k: v for k, v in gat.state_dict().items()}, **dict(pred.state_dict())}
    ref_new = {**{"gat." + k: v.detach() for k, v in q.items()}, **{k: v.detach() for k, v in r.items()}}
    for k, v in new.items():
        solid = g_ref[k].abs() > 1e-6
        d = (v.cpu() - ref_new[k]).abs()
        assert float(d[solid].max() if solid.any() else 0.0) <= 2e-6, (k, float(d[solid].max()))
        assert float(d.max()) <= 2.1 * lr, k
    # and a second iteration runs on the updated parameters (packed-weight caches follow the in-place update)
    # (stacked (B, Np, .) tensors are accepted like lists of per-image tensors)
    out2 = tr.step(images, masks, torch.stack(feats).to(cuda), torch.stack(funet).to(cuda), torch.stack(ylab).to(cuda), ei.to(cuda))
    assert float(out2["l_partition"]) != float(out["l_partition"]) and torch.isfinite(out2["total"])
