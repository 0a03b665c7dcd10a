"""GAT training on the HIP path: mgu_gat_layer_backward behind the autograd nodes of mgunet.GATNetwork, against the gradients
the REFERENCE GATNetwork produced under torch autograd (tests/golden/gat_grad.npz; eval-mode dropout: the reference's train-mode
dropout draws from torch's RNG stream).  Tolerance: relative to max|gradient|, 2e-5 + 20 x the deviation of
the reference's own fp32 gradients from float64 (the wide-logit case is conditioned like 5e-6)."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O
from test_oracle_golden import GATGRAD_CASES, gatgrad_inputs

pytestmark = pytest.mark.gpu


def build(cfg, layers, p, dev, train=False):
    g = mgunet.GATNetwork(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers, dropout_rate=0.0 if train else 0.1)
    g.load_state_dict(p)
    g = g.to(dev)
    return g.train() if train else g.eval()


@pytest.mark.parametrize("tag", list(GATGRAD_CASES))
def test_gat_backward_vs_reference_fixture(cuda, golden, tag):
    g = golden["gat_grad"]
    cfg, layers, ei, X, R, p = gatgrad_inputs(golden, tag)
    net = build(cfg, layers, p, cuda)
    Xd = X.to(cuda).requires_grad_(True)
    out = net(Xd, ei.to(cuda))
    assert out.requires_grad
    (out * R.to(cuda)).sum().backward()
    cond = float(g[tag + "_cond"])
    tol = 2e-5 + 20 * cond

    def close(got, key):
        ref64 = g[key]
        d = np.abs(got.detach().cpu().numpy().astype(np.float64) - ref64).max()
        assert d <= tol * max(1.0, np.abs(ref64).max()), (tag, key, d, np.abs(ref64).max())
    close(Xd.grad, tag + "_dX64")
    for k, v in net.named_parameters():
        assert v.grad is not None, k
        close(v.grad, f"{tag}_d64_{k}")
        ref32 = g[f"{tag}_d_{k}"]            # and the reference's fp32 gradients themselves
        assert np.abs(v.grad.cpu().numpy() - ref32).max() <= (tol + 2 * cond) * max(1.0, np.abs(ref32).max()), k


def test_gat_backward_batched_graphs_and_determinism(cuda, golden):
    """Two graphs in one block-diagonal call (graph_ptr: a max per graph, graph_attention.py:86 applied per image as the
    reference's per-image loop does) give the per-graph gradients: dX rows per graph, parameter gradients summed; a second
    run gives the same bytes."""
    cfg, layers, ei, X, R, p = gatgrad_inputs(golden, "grid")
    _, _, ei2, X2, R2, _ = gatgrad_inputs(golden, "g10")
    X2 = torch.cat([X2, X2[:6]], 0)                             # 16 nodes, features 32
    R2 = torch.from_numpy(O.formula_normal("gatgrad/b/r2", (16, cfg[2]), seed=9))
    N1 = X.shape[0]
    net = build(cfg, layers, p, cuda, train=True)              # train mode with dropout 0 is accepted
    grads = []
    for Xi, ei_i, Ri in ((X, ei, R), (X2, ei2, R2)):
        net.zero_grad()
        Xi = Xi.to(cuda).requires_grad_(True)
        (net(Xi, ei_i.to(cuda)) * Ri.to(cuda)).sum().backward()
        grads.append((Xi.grad.clone(), {k: v.grad.clone() for k, v in net.named_parameters()}))
    eib = torch.cat([ei, ei2 + N1], 1).to(cuda)
    Xb = torch.cat([X, X2], 0).to(cuda)
    Rb = torch.cat([R, R2], 0).to(cuda)
    gp = torch.tensor([0, N1, N1 + 16])
    runs = []
    for _ in range(2):
        net.zero_grad()
        Xv = Xb.clone().requires_grad_(True)
        (net(Xv, eib, graph_ptr=gp) * Rb).sum().backward()
        runs.append((Xv.grad.clone(), {k: v.grad.clone() for k, v in net.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0]) and all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    dXb, dPb = runs[0]
    sx = max(1.0, float(dXb.abs().max()))
    assert float((dXb[:N1] - grads[0][0]).abs().max()) <= 2e-5 * sx and float((dXb[N1:] - grads[1][0]).abs().max()) <= 2e-5 * sx
    for k in dPb:
        want = grads[0][1][k] + grads[1][1][k]
        assert float((dPb[k] - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), k


def test_gat_trains_with_torch_optimizer(cuda, golden):
    """A few Adam steps on the patch-GAT shape lower a regression loss: parameters are ordinary nn.Parameters, gradients come
    from the HIP backward, the prepared weights are rebuilt when the optimizer changes them."""
    cfg, layers, ei, X, R, p = gatgrad_inputs(golden, "grid")
    net = build(cfg, layers, p, cuda, train=True)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    Xd, eid, tgt = X.to(cuda), ei.to(cuda), (R * 0.1).to(cuda)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = ((net(Xd, eid) - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.8 * losses[0], losses
    # the oracle agrees with the trained weights (eval forward after training)
    with torch.no_grad():
        y = net.eval()(Xd, eid)
        oy = O.gat_network_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, X, ei, cfg[3], layers)
    assert float((y.cpu() - oy).abs().max()) <= 1e-3


GATDROP_CASES = {   # tag: (cfg, layers, N, graph, x scale, p) -- oracle/make_golden.py GATDROP_CASES
    "edge10": ((8, 16, 16, 4), 1, 10, "edge10", 1.0, 0.1),
    "patch": ((32, 128, 64, 4), 1, 36, "grid6", 1.0, 0.1),
    "seg": ((64, 64, 2, 2), 1, 36, "grid6", 0.5, 0.1),
    "l2h1": ((8, 16, 8, 1), 2, 11, "edge_iso", 1.0, 0.3),
}
def gatdrop_inputs(golden, tag):
    cfg, layers, N, graph, xs, p = GATDROP_CASES[tag]
    ei = torch.from_numpy(golden["gat_dropout"][tag + "_ei"])          # the edge list the reference ran on (stored with the fixture)
    X = torch.from_numpy(O.formula_normal(f"gatdrop/{tag}/x", (N, cfg[0]), seed=5)) * xs
    R = torch.from_numpy(O.formula_normal(f"gatdrop/{tag}/r", (N, cfg[2]), seed=6))
    params = O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=5, scale=1.0)
    return cfg, layers, ei, X, R, params, p


@pytest.mark.parametrize("tag", list(GATDROP_CASES))
def test_gat_train_mode_dropout_vs_reference_fixture(cuda, golden, tag):
    """GATNetwork.train() with dropout_rate > 0 (graph_attention.py:97, :160; default 0.1, configs/model.yaml:18): the masks of
    tests/golden/gat_dropout.npz -- the draw the REFERENCE ran with, its nn.Dropout modules replaced by explicit masks in
    oracle/make_golden.py gen_gatdrop -- are injected through `layer.dropout_masks`; output and every gradient must match what the
    reference's forward and torch autograd produced (2e-5 of the tensor's max, the kernel bar)."""
    g = golden["gat_dropout"]
    cfg, layers, ei, X, R, params, p = gatdrop_inputs(golden, tag)
    net = mgunet.GATNetwork(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers, dropout_rate=p)
    net.load_state_dict(params)
    net = net.to(cuda).train()
    for l, layer in enumerate(net.gat_layers):
        layer.dropout_masks = (torch.from_numpy(g[f"{tag}_emask{l}"]), torch.from_numpy(g[f"{tag}_omask{l}"]))
    Xd = X.to(cuda).requires_grad_(True)
    out = net(Xd, ei.to(cuda))
    (out * R.to(cuda)).sum().backward()

    def close(got, key):
        ref = g[key]
        d = np.abs(got.detach().cpu().numpy() - ref).max()
        assert d <= 2e-5 * max(1.0, np.abs(ref).max()), (tag, key, d, np.abs(ref).max())
    close(out, tag + "_out")
    close(Xd.grad, tag + "_dX")
    for k, v in net.named_parameters():
        assert v.grad is not None, k
        close(v.grad, f"{tag}_d_{k}")
    # eval mode is untouched by the hook
    net.eval()
    with torch.no_grad():
        ev = net(X.to(cuda), ei.to(cuda))
        ref = O.gat_network_forward(params, X, ei, cfg[3], layers)
    assert float((ev.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_gat_train_mode_draws_its_own_masks(cuda, golden):
    """Without injected masks the layer draws them on the device (mgu_dropout_mask, Philox-4x32-10): reproducible under
    mgunet.seed_dropout, different from call to call, p of the coefficients dropped, kept ones scaled by 1 / (1 - p); the output
    equals the oracle's train-mode forward on the masks the generator produced."""
    from mgunet import _lib
    from mgunet.gat import _context
    cfg, layers, ei, X, R, params, _ = gatdrop_inputs(golden, "patch")
    p = 0.25
    net = mgunet.GATNetwork(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers, dropout_rate=p)
    net.load_state_dict(params)
    net = net.to(cuda).train()
    Xd, eid = X.to(cuda), ei.to(cuda)
    mgunet.seed_dropout(123)
    a1, a2 = net(Xd, eid).detach().clone(), net(Xd, eid).detach().clone()
    mgunet.seed_dropout(123)
    b1 = net(Xd, eid).detach().clone()
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
    # the generator itself: statistics and the stream / seed / index contract
    L, c = _lib.lib(), _context(cuda)
    n = 1 << 20
    m = torch.empty(n, device=cuda)
    st = _lib.current_stream_ptr(cuda)
    _lib.check(L.mgu_dropout_mask(c.handle, 7, 3, n, p, m.data_ptr(), st), c.handle)
    vals = torch.unique(m).cpu().tolist()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6
    assert abs(float((m == 0).float().mean()) - p) < 3e-3
    m2 = torch.empty(1000, device=cuda)
    _lib.check(L.mgu_dropout_mask(c.handle, 7, 3, 1000, p, m2.data_ptr(), st), c.handle)
    assert torch.equal(m2, m[:1000])                                  # element i depends on (seed, stream, i) only
    _lib.check(L.mgu_dropout_mask(c.handle, 7, 4, 1000, p, m2.data_ptr(), st), c.handle)
    assert not torch.equal(m2, m[:1000])
    # the layer's output under the masks it drew == the oracle on the same masks (drawn again with the same stream ids)
    E, H, N = ei.shape[1], cfg[3], X.shape[0]
    mgunet.seed_dropout(99)
    got = net(Xd, eid).detach().cpu()
    em = torch.empty((E, H), device=cuda)
    om = torch.empty((N, cfg[2]), device=cuda)
    _lib.check(L.mgu_dropout_mask(c.handle, 99, 1, E * H, p, em.data_ptr(), st), c.handle)   # stream 1: the edge mask, CSR order
    _lib.check(L.mgu_dropout_mask(c.handle, 99, 2, N * cfg[2], p, om.data_ptr(), st), c.handle)
    perm = torch.sort(ei[1], stable=True).indices
    em_coo = torch.empty((H, E))
    em_coo[:, perm] = em.cpu().t()
    ref = O.gat_network_forward(params, X, ei, H, layers, masks=[(em_coo, om.cpu())])
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
