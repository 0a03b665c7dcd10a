#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE implementation (build container only).

Imports the reference's own `model/` and `preprocessing/graph_construction` packages from
/root/reference/MinGraph-UNet (they never travel to the GPU box), feeds them the formula
weights / inputs of oracle/mgunet_oracle.py, checks that the oracle restatement agrees, and
stores the reference's outputs as small fixtures.  Fixtures hold data only (inputs that are
not formula-derivable, expected outputs, sample indices) -- never reference source text.

Usage:  python oracle/make_golden.py [--only tiny,gat,gatgrad,graph,mincut,region,dethead,losses,scriptlosses,c1,c2,c4,c5,bf16ref,gatdrop]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/MinGraph-UNet"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import mgunet_oracle as O  # noqa: E402
from model.unet.unet_model import UNet as RefUNet  # noqa: E402
from model.gat.graph_attention import GATNetwork as RefGAT, MultiHeadGATLayer as RefMH  # noqa: E402
from preprocessing.graph_construction.patch_graph_construction import PatchGraphConstructor as RefPGC  # noqa: E402
from model.graph_partition.mincut_refinement import MinCutRefinement as RefMinCut  # noqa: E402
from model.fusion_detection.feature_fusion import FeatureFusion as RefFusion  # noqa: E402
from model.fusion_detection.detection_head import DetectionHead as RefDet  # noqa: E402
from model.unet.feature_loss import FeatureConsistencyLoss as RefFeatLoss  # noqa: E402
from model.unet.shape_loss import EllipticalShapeLoss as RefShapeLoss  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 1e-5


def sample_idx(name, numel, n=4096):
    u = O.formula_uniform(name, (n,), 0.0, 1.0, 7).astype(np.float64)
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def ref_unet(cfg, params, train=False):
    m = RefUNet(*cfg)
    m.load_state_dict(params)
    return m.train() if train else m.eval()


def check(name, a, b, tol=TOL):
    d = float((a - b).abs().max()) if a.numel() else 0.0
    print(f"   oracle-vs-reference {name}: max|d|={d:.3e} (max|ref|={float(b.abs().max()) if b.numel() else 0:.3f})")
    assert d <= tol, (name, d)


def save(fname, **arrs):
    path = os.path.join(GOLD, fname)
    np.savez_compressed(path, **arrs)
    print(f"   wrote {fname}: {os.path.getsize(path)/1024:.1f} KiB")


def gen_tiny():
    print("[tiny] UNet(1,2,8,2) @1x1x32x32 eval + train; UNet(3,3,8,2) @2x3x37x45 (odd: pad + floor pool)")
    out = {}
    for tag, cfg, shape in (("a", (1, 2, 8, 2), (1, 1, 32, 32)), ("b", (3, 3, 8, 2), (2, 3, 37, 45)),
                            ("c", (3, 2, 8, 3), (2, 3, 64, 48))):
        p = O.make_unet_params(*cfg, seed=11)
        x = torch.from_numpy(O.formula_normal(f"tiny/{tag}/x", shape, seed=11))
        m = ref_unet(cfg, p)
        with torch.no_grad():
            lg, sk, ft = m(x)
            olg, osk, oft = O.unet_forward(p, x, cfg[3])
        check(f"{tag}.logits", olg, lg)
        for i in range(cfg[3]):
            check(f"{tag}.skip{i}", osk[i], sk[i])
            check(f"{tag}.feat{i}", oft[i], ft[i])
        out[f"{tag}_logits"] = lg.numpy()
        for i in range(cfg[3]):
            out[f"{tag}_skip{i}"] = sk[i].numpy()
            out[f"{tag}_feat{i}"] = ft[i].numpy()
        # train-mode forward: batch statistics + running-stat update (unet_encoder.py:12-13)
        mt = ref_unet(cfg, p, train=True)
        with torch.no_grad():
            lgt, _, _ = mt(x)
            stats = {}
            olgt, _, _ = O.unet_forward(p, x, cfg[3], training=True, new_stats=stats)
        check(f"{tag}.train_logits", olgt, lgt, 2e-5)
        sd = mt.state_dict()
        for k, v in stats.items():
            check(f"{tag}.{k}", v, sd[k])
        out[f"{tag}_train_logits"] = lgt.numpy()
        out[f"{tag}_bn_rm_last"] = sd["decoder.decoder_blocks.%d.conv_block.bn2.running_mean" % (cfg[3] - 1)].numpy()
        out[f"{tag}_bn_rv_last"] = sd["decoder.decoder_blocks.%d.conv_block.bn2.running_var" % (cfg[3] - 1)].numpy()
        out[f"{tag}_bn_rm_first"] = sd["encoder.encoder_blocks.0.bn1.running_mean"].numpy()
        out[f"{tag}_bn_rv_first"] = sd["encoder.encoder_blocks.0.bn1.running_var"].numpy()
    save("unet_tiny.npz", **out)


# graph_attention.py:209-210 -- the module's own 10-node / 18-edge smoke fixture (data)
EDGE10 = np.array([[0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 7, 8, 8, 9, 9, 4],
                   [1, 0, 2, 1, 3, 2, 0, 3, 5, 4, 6, 5, 8, 7, 9, 8, 4, 9]], dtype=np.int64)


def gen_gat():
    print("[gat] 10-node fixture, isolated targets, wide-logit, concat layer, 2-layer/1-head")
    out = {"edge10": EDGE10}

    def run(tag, cfg, X, ei, scale=1.0, layers=1):
        p = O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=3, scale=scale)
        g = RefGAT(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers).eval()
        g.load_state_dict(p)
        with torch.no_grad():
            y = g(X, torch.from_numpy(ei))
            oy = O.gat_network_forward(p, X, torch.from_numpy(ei), cfg[3], layers)
        check(tag, oy, y)
        out[tag + "_out"] = y.numpy()

    X10 = torch.from_numpy(O.formula_normal("gat/x10", (10, 32), seed=3))
    run("g10", (32, 64, 16, 4), X10, EDGE10)
    # isolated targets: nodes 10,11 have no in-edges (rows must be exactly elu(0)=0); node 11 has no edges at all
    ei_iso = np.concatenate([EDGE10, np.array([[10, 10], [0, 3]], dtype=np.int64)], axis=1)
    X12 = torch.from_numpy(O.formula_normal("gat/x12", (12, 32), seed=3))
    out["edge_iso"] = ei_iso
    run("iso", (32, 64, 16, 4), X12, ei_iso)
    # wide-logit case: |e| range > 23 so the +1e-10 and the GLOBAL max bite (graph_attention.py:86,96)
    Xw = torch.from_numpy(O.formula_normal("gat/xw", (10, 32), seed=4)) * 4.0
    run("wide", (32, 64, 16, 2), Xw, EDGE10, scale=3.0)
    p = O.make_gat_params(32, 64, 16, 2, 1, seed=3, scale=3.0)
    with torch.no_grad():
        h = Xw @ p["gat_layers.0.heads.0.W.weight"].t()
        e = torch.nn.functional.leaky_relu(torch.cat([h[EDGE10[0]], h[EDGE10[1]]], 1) @ p["gat_layers.0.heads.0.a.weight"].t(), 0.2)
    print(f"   wide case: e range [{float(e.min()):.1f}, {float(e.max()):.1f}]")
    assert float(e.max() - e.min()) > 23
    # moderate range (~25-60): every edge still contributes, the global max shifts per-target sums by many e-folds
    Xm = torch.from_numpy(O.formula_normal("gat/xm", (10, 32), seed=6)) * 2.0
    run("mid", (32, 64, 16, 4), Xm, EDGE10, scale=1.5)
    # 2 layers, 1 head (the only multi-layer shape the reference can run: SURVEY G3)
    run("l2h1", (32, 24, 8, 1), X10, EDGE10, layers=2)
    # MultiHeadGATLayer(concat=True) directly: graph_attention.py:153-155
    mh = RefMH(32, 64, 4, 0.1, 0.2, concat=True).eval()
    pc = {}
    for h_ in range(4):
        for nm, shp in ((f"heads.{h_}.W.weight", (16, 32)), (f"heads.{h_}.a.weight", (1, 32))):
            a = 1.414 * float(np.sqrt(6.0 / (shp[0] + shp[1])))
            pc[nm] = torch.from_numpy(O.formula_uniform("mhc/" + nm, shp, -a, a, 5))
    mh.load_state_dict(pc)
    with torch.no_grad():
        yc = mh(X10, torch.from_numpy(EDGE10))
        oc = torch.cat([O.gat_head_forward(X10, torch.from_numpy(EDGE10), pc[f"heads.{k}.W.weight"],
                                           pc[f"heads.{k}.a.weight"]) for k in range(4)], 1)
    check("mh_concat", oc, yc)
    out["mhc_out"] = yc.numpy()
    save("gat_small.npz", **out)


GATGRAD_CASES = {   # tag: (cfg (in, hidden, out, heads), layers, nodes, graph, x scale, weight scale)
    "g10": ((32, 64, 16, 4), 1, 10, "edge10", 1.0, 1.0),
    "iso": ((32, 64, 16, 4), 1, 12, "edge_iso", 1.0, 1.0),
    "wide": ((32, 64, 16, 2), 1, 10, "edge10", 4.0, 3.0),          # |e| range > 23: the 1e-10 and the graph-wide max bite
    "mid": ((32, 64, 16, 4), 1, 10, "edge10", 2.0, 1.5),
    "l2h1": ((32, 24, 8, 1), 2, 10, "edge10", 1.0, 1.0),           # two layers chained (one head: SURVEY G3)
    "grid": ((32, 128, 64, 4), 1, 256, "grid16", 1.0, 1.0),        # the patch GAT's shape on a 16 x 16 patch grid
    "pred": ((64, 64, 2, 2), 1, 64, "grid8", 1.0, 1.0),            # the segment predictor's shape (train_end_to_end.py:155-163)
}


def gatgrad_inputs(tag):
    cfg, layers, N, graph, xs, ws = GATGRAD_CASES[tag]
    if graph == "edge10":
        ei = EDGE10
    elif graph == "edge_iso":
        ei = np.concatenate([EDGE10, np.array([[10, 10], [0, 3]], dtype=np.int64)], axis=1)
    else:
        side = int(graph[4:])
        ei = O.patch_graph_edges(side * 16, side * 16, 16)
    X = torch.from_numpy(O.formula_normal(f"gatgrad/{tag}/x", (N, cfg[0]), seed=3)) * xs
    R = torch.from_numpy(O.formula_normal(f"gatgrad/{tag}/r", (N, cfg[2]), seed=4))     # loss = sum(out * R)
    p = O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=3, scale=ws)
    return cfg, layers, torch.from_numpy(np.ascontiguousarray(ei)), X, R, p


def gen_gatgrad():
    print("[gatgrad] gradients of the reference GATNetwork (eval-mode dropout) under torch autograd: d/dX, d/dW, d/da")
    out = {}
    for tag in GATGRAD_CASES:
        cfg, layers, ei, X, R, p = gatgrad_inputs(tag)
        g = RefGAT(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers).eval()
        g.load_state_dict(p)
        Xr = X.clone().requires_grad_(True)
        (g(Xr, ei) * R).sum().backward()
        # the oracle restatement under autograd
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        Xo = X.clone().requires_grad_(True)
        (O.gat_network_forward(q, Xo, ei, cfg[3], layers) * R).sum().backward()
        check(tag + ".dX", Xo.grad, Xr.grad, tol=2e-6 * max(1.0, float(Xr.grad.abs().max())))
        out[tag + "_dX"] = Xr.grad.numpy()
        for k, v in g.named_parameters():
            check(f"{tag}.d{k}", q[k].grad, v.grad, tol=2e-6 * max(1.0, float(v.grad.abs().max())))
            out[f"{tag}_d_{k}"] = v.grad.numpy()
        # float64 yardstick from the restatement (the reference class itself is fp32-only: its torch.zeros() calls carry no dtype,
        # SURVEY appendix A); it says how far the reference's OWN fp32 gradients sit from exact arithmetic (wide-logit case!)
        q64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
        X64 = X.double().requires_grad_(True)
        (O.gat_network_forward(q64, X64, ei, cfg[3], layers) * R.double()).sum().backward()
        out[tag + "_dX64"] = X64.grad.numpy()
        for k in q64:
            out[f"{tag}_d64_{k}"] = q64[k].grad.numpy()
        dev = float((Xr.grad.double() - X64.grad).abs().max() / (X64.grad.abs().max() + 1e-300))
        out[tag + "_cond"] = np.float64(dev)
        print(f"   {tag}: max|dX| {float(Xr.grad.abs().max()):.3e}, reference fp32 vs float64 {dev:.1e}")
    save("gat_grad.npz", **out)


def gen_mincut():
    print("[mincut] segment predictor + normalized-cut loss (SURVEY 8f row 1)")
    out = {}
    ref = RefMinCut()

    def ref_predictor(in_dim, K, hidden, use_gnn, heads, params):
        # PatchSegmentPredictor is defined inside scripts/train_end_to_end.py (:40-70), a script whose import runs the
        # whole training set-up; it is exactly this wrapper: GATNetwork(in, hidden, K, heads, 1 layer) or Linear-ReLU-Linear
        if use_gnn:
            g = RefGAT(in_dim, hidden if hidden else in_dim, K, heads, num_gat_layers=1, dropout_rate=0.1, alpha=0.2).eval()
            g.load_state_dict({k[len("gnn_predictor."):]: v for k, v in params.items()})
            return lambda x, ei: g(x, ei)
        hd = hidden if hidden is not None else in_dim * 2
        m = torch.nn.Sequential(torch.nn.Linear(in_dim, hd), torch.nn.ReLU(), torch.nn.Linear(hd, K)).eval()
        m.load_state_dict({k[len("mlp_predictor."):]: v for k, v in params.items()})
        return lambda x, ei: m(x)

    def run(tag, X, ei, K, hidden, use_gnn, heads, seed, logit_shift=None):
        eit = torch.from_numpy(ei)
        p = O.make_segment_predictor_params(X.shape[1], K, hidden, use_gnn, heads, seed=seed)
        pred = ref_predictor(X.shape[1], K, hidden, use_gnn, heads, p)
        net = pred if logit_shift is None else (lambda x, e: pred(x, e) + logit_shift)
        with torch.no_grad():
            loss, soft = ref(X, eit, K, net)                       # MinCutRefinement.forward, :163-205
            w = ref.compute_edge_weights_for_ncut(X, eit)          # :30-52
            lg = O.segment_predictor_forward(p, X, eit, use_gnn, heads)
            if logit_shift is not None:
                lg = lg + logit_shift
            oloss, osoft, ohard = O.mincut_forward(X, eit, K, lg)
        check(tag + ".soft", osoft, soft)
        check(tag + ".loss", oloss.reshape(1), torch.as_tensor(loss, dtype=torch.float32).reshape(1), tol=1e-5 * max(1.0, float(loss)))
        check(tag + ".w", O.ncut_edge_weights(X, eit), w)
        assert torch.equal(ohard, torch.argmax(soft, dim=1))        # train_end_to_end.py:356
        out[tag + "_loss"] = np.float32(float(loss))
        out[tag + "_soft"] = soft.numpy()
        out[tag + "_w"] = w.numpy()
        out[tag + "_logits"] = lg.numpy()
        print(f"   {tag}: N={X.shape[0]} E={ei.shape[1]} K={K} loss={float(loss):.6f} w range [{float(w.min()):.3e}, {float(w.max()):.3e}]")

    # (a) the configuration of train_end_to_end.py:155-163 on an 8x8 patch graph: GAT predictor 64 -> 2, 2 heads
    ei8 = O.patch_graph_edges(128, 128, 16)
    Xa = torch.from_numpy(O.formula_normal("mincut/a/x", (64, 64), seed=1)) * 0.15
    run("a", Xa, ei8, 2, 32, True, 2, seed=5)
    # (b) MLP predictor, K = 3, a DIRECTED random graph (degree is summed over the SOURCE index only, :96), node 49 has
    #     no outgoing edge, node 48 no edge at all
    u = O.formula_uniform("mincut/b/e", (2, 200), 0.0, 1.0, 3)
    eib = np.stack([np.minimum((u[0] * 48).astype(np.int64), 47), np.minimum((u[1] * 50).astype(np.int64), 49)])
    Xb = torch.from_numpy(O.formula_normal("mincut/b/x", (50, 24), seed=2)) * 0.2
    out["b_edges"] = eib
    run("b", Xb, eib, 3, None, False, 1, seed=6)
    # (c) a segment nobody belongs to: its association is below 1e-8 and the term is skipped (:152-153)
    run("c", Xb, eib, 3, None, False, 1, seed=6, logit_shift=torch.tensor([0.0, -60.0, 0.0]))
    # (d) the headline graph: 1024 patches of a 512x512 image, 64 features
    ei32 = O.patch_graph_edges(512, 512, 16)
    Xd = torch.from_numpy(O.formula_normal("mincut/d/x", (1024, 64), seed=4)) * 0.15
    run("d", Xd, ei32, 2, 32, True, 2, seed=7)
    # the reference's own shape check (:73-74)
    try:
        ref.normalized_cut_loss(Xa, torch.from_numpy(ei8), torch.zeros(64, 3), 2)
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    save("mincut.npz", **out)


def _ref_predictor_module(in_dim, K, hidden, use_gnn, heads, params):
    """PatchSegmentPredictor (scripts/train_end_to_end.py:40-70) is this wrapper around the reference GATNetwork / an MLP; the script
    itself cannot be imported (its import runs the whole training set-up).  Returns (module, call(x, ei))."""
    if use_gnn:
        g = RefGAT(in_dim, hidden if hidden else in_dim, K, heads, num_gat_layers=1, dropout_rate=0.1, alpha=0.2).eval()
        g.load_state_dict({k[len("gnn_predictor."):]: v for k, v in params.items()})
        return g, "gnn_predictor.", (lambda x, ei: g(x, ei))
    hd = hidden if hidden is not None else in_dim * 2
    m = torch.nn.Sequential(torch.nn.Linear(in_dim, hd), torch.nn.ReLU(), torch.nn.Linear(hd, K)).eval()
    m.load_state_dict({k[len("mlp_predictor."):]: v for k, v in params.items()})
    return m, "mlp_predictor.", (lambda x, ei: m(x))


def gen_mincutgrad():
    print("[mincutgrad] L_partition.backward() of the reference MinCutRefinement + segment predictor (+ patch GAT) under torch autograd")
    out = {}
    ref = RefMinCut()
    for tag in O.MINCUTGRAD_CASES:
        ei, X, R, p, K, use_gnn, heads, shift = O.mincutgrad_inputs(tag)
        hidden = O.MINCUTGRAD_CASES[tag][4]
        mod, prefix, call = _ref_predictor_module(X.shape[1], K, hidden, use_gnn, heads, p)
        net = call if shift is None else (lambda x, e: call(x, e) + shift)
        Xr = X.clone().requires_grad_(True)
        loss, soft = ref(Xr, ei, K, net)                         # MinCutRefinement.forward, :163-205
        total = loss + 0.05 * (soft * R).sum()                   # the second term sends a gradient through the returned assignments
        total.backward()
        out[tag + "_loss"] = np.float32(float(loss))
        out[tag + "_dX"] = Xr.grad.numpy()
        for k, v in mod.named_parameters():
            out[f"{tag}_d_{prefix}{k}"] = v.grad.numpy().copy()
        # the loss function called directly with soft assignments as the leaf (normalized_cut_loss, :55-160)
        P = soft.detach().clone().requires_grad_(True)
        Xq = X.clone().requires_grad_(True)
        l2 = ref.normalized_cut_loss(Xq, ei, P, K)
        (2.5 * l2).backward()
        out[tag + "_direct_dP"] = P.grad.numpy()
        out[tag + "_direct_dX"] = Xq.grad.numpy()
        # the analytic restatement against the reference's autograd, fp32 and (as a yardstick for the tests) float64
        aP, aF = O.normalized_cut_loss_grad(X, ei, soft.detach(), K, gloss=2.5)
        check(tag + ".direct_dP", aP, P.grad, tol=5e-6 * max(1.0, float(P.grad.abs().max())))
        check(tag + ".direct_dX", aF, Xq.grad, tol=5e-6 * max(1.0, float(Xq.grad.abs().max())))
        aP64, aF64 = O.normalized_cut_loss_grad(X.double(), ei, soft.detach().double(), K, gloss=2.5)
        out[tag + "_direct_dP64"] = aP64.numpy()
        out[tag + "_direct_dX64"] = aF64.numpy()
        print(f"   {tag}: loss {float(loss):.6f} max|dX| {float(Xr.grad.abs().max()):.3e} max|dP| {float(P.grad.abs().max()):.3e}")

    # patch GAT -> MinCut (GNN predictor): the chain the e2e loop trains (train_end_to_end.py:332-356, 219-226), three SGD steps
    ei = torch.from_numpy(O.patch_graph_edges(128, 128, 16))
    X0 = torch.from_numpy(O.formula_normal("mincutgrad/chain/x", (64, 16), seed=21)) * 0.5
    gp = O.make_gat_params(16, 8, 16, 2, 1, seed=22)
    pp = O.make_segment_predictor_params(16, 2, 8, True, 2, seed=23)
    gat = RefGAT(16, 8, 16, 2, num_gat_layers=1).eval()
    gat.load_state_dict(gp)
    pred, prefix, call = _ref_predictor_module(16, 2, 8, True, 2, pp)
    params = list(gat.parameters()) + list(pred.parameters())
    opt = torch.optim.SGD(params, lr=0.2)
    losses = []
    for step in range(4):
        opt.zero_grad()
        feats = gat(X0, ei)
        loss, soft = ref(feats, ei, 2, call)
        losses.append(float(loss))
        if step == 0:
            loss.backward()
            for k, v in gat.named_parameters():
                out[f"chain_d_gat.{k}"] = v.grad.numpy().copy()
            for k, v in pred.named_parameters():
                out[f"chain_d_{prefix}{k}"] = v.grad.numpy().copy()
        elif step < 3:
            loss.backward()
        if step < 3:
            opt.step()
    out["chain_losses"] = np.array(losses, np.float32)
    for k, v in gat.state_dict().items():
        out[f"chain_final_gat.{k}"] = v.numpy().copy()
    for k, v in pred.state_dict().items():
        out[f"chain_final_{prefix}{k}"] = v.numpy().copy()
    print(f"   chain: losses {losses}")
    save("mincut_grad.npz", **out)


def gen_region():
    print("[region] label-mean pooling -> region GAT -> map back -> nearest upsample -> FeatureFusion (SURVEY 8f row 2)")
    import torch.nn.functional as TF
    out = {}

    def ref_region_stage(feats, hard, K, gat, nph, npw, H, W):
        # scripts/train_end_to_end.py:366-421 is loop-body code, not a function: the same torch calls in the same order,
        # with the reference's GATNetwork as the region model
        reg = torch.zeros(K, feats.shape[1])
        for k in range(K):
            mk = hard == k
            if mk.sum() > 0:
                reg[k] = feats[mk].mean(dim=0)
        if K > 1:
            s_, t_ = torch.triu_indices(K, K, offset=1)
            ei = torch.stack([torch.cat([s_, t_]), torch.cat([t_, s_])], dim=0)
        else:
            ei = torch.empty((2, 0), dtype=torch.long)
        emb = gat(reg, ei) if ei.numel() > 0 else reg
        mapped = emb[hard]
        pix = TF.interpolate(mapped.T.reshape(emb.shape[1], nph, npw).unsqueeze(0), size=(H, W), mode="nearest").squeeze(0)
        return emb, pix

    cases = (("a", 64, 64, 2, 4, 16), ("b", 37, 45, 3, 2, 16), ("c", 32, 48, 1, 4, 16), ("d", 512, 512, 2, 4, 16))
    for tag, H, W, K, heads, patch in cases:
        nph, npw = O.patch_grid(H, W, patch)
        Np, D = nph * npw, 64
        feats = torch.from_numpy(O.formula_normal(f"region/{tag}/x", (Np, D), seed=1)) * 0.5
        hard = torch.from_numpy(O.formula_labels(f"region/{tag}/y", (Np,), K, seed=2))
        if tag == "b":
            hard[hard == 1] = 0          # an empty segment: its region feature stays zero (:369-373)
        p = O.make_gat_params(D, 128, D, heads, 1, seed=9)
        gat = RefGAT(D, 128, D, heads, num_gat_layers=1).eval()
        gat.load_state_dict(p)
        with torch.no_grad():
            emb, pix = ref_region_stage(feats, hard, K, gat, nph, npw, H, W)
            oemb, opix = O.region_stage(feats, hard, K, p, heads, nph, npw, H, W)
        check(f"{tag}.emb", oemb, emb)
        check(f"{tag}.pix", opix, pix)
        out[f"{tag}_emb"] = emb.numpy()
        out[f"{tag}_hard"] = hard.numpy()
        idx = sample_idx(f"region/{tag}/idx", pix.numel(), 2048)
        out[f"{tag}_pix_idx"] = idx
        out[f"{tag}_pix"] = pix.reshape(-1).numpy()[idx]
        # FeatureFusion (train_end_to_end.py:433-437): one F_u scale + the pixel-mapped F_g
        fu = torch.from_numpy(O.formula_normal(f"region/{tag}/fu", (1, 32, H, W), seed=3))
        fuser = RefFusion(unet_feature_dims=[32], gat_feature_dim=D)
        with torch.no_grad():
            ff = fuser(f_u_list=[fu], f_g=pix.unsqueeze(0), target_spatial_size=(H, W))
        check(f"{tag}.fused", O.feature_fusion([fu], opix.unsqueeze(0)), ff)
        assert tuple(ff.shape) == (1, 32 + D, H, W)
        fidx = sample_idx(f"region/{tag}/fidx", ff.numel(), 2048)
        out[f"{tag}_fused_idx"] = fidx
        out[f"{tag}_fused"] = ff.reshape(-1).numpy()[fidx]
        print(f"   {tag}: {H}x{W} -> {nph}x{npw} patches, K={K}, emb max {float(emb.abs().max()):.3f}")
    fa = RefFusion([64], 64, fusion_method="add")
    a_, b_ = torch.ones(1, 64, 4, 4), torch.full((1, 64, 4, 4), 2.0)
    assert torch.equal(fa([a_], b_), O.feature_fusion([a_], b_, "add"))
    save("region.npz", **out)


def gen_dethead():
    print("[dethead] DetectionHead: conv-ReLU-BN x2, global average pool, MLP, sigmoid heads (SURVEY 8f row 2)")
    out = {}
    for tag, B, C, H, W, ncls, flat in (("a", 2, 96, 24, 40, 1, False), ("b", 1, 96, 128, 128, 3, False), ("c", 3, 64, 17, 9, 2, False),
                                        ("f", 4, 24, 0, 0, 1, True)):
        p = O.make_detection_head_params(C, ncls, 256, flat, seed=13)
        m = RefDet(C, ncls, fc_hidden_dim=256, input_is_flat=flat).eval()
        sd = dict(p)
        for k in ("2", "5"):
            if f"conv_block.{k}.weight" in sd:
                sd[f"conv_block.{k}.num_batches_tracked"] = torch.tensor(0)
        m.load_state_dict(sd)
        x = torch.from_numpy(O.formula_normal(f"det/{tag}/x", (B, C) if flat else (B, C, H, W), seed=2))
        with torch.no_grad():
            ref = m(x)
            got = O.detection_head_forward(p, x, ncls, flat)
        assert len(ref) == len(got) == (3 if ncls > 1 else 2)
        for nm, r, g_ in zip(("bbox", "conf", "cls"), ref, got):
            check(f"{tag}.{nm}", g_, r)
            out[f"{tag}_{nm}"] = r.numpy()
        print(f"   {tag}: input {tuple(x.shape)} classes {ncls} -> bbox {tuple(ref[0].shape)} conf {tuple(ref[1].shape)}")
    save("dethead.npz", **out)


def ellipse_mask(H, W, cy, cx, a, b, theta):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    c, s_ = np.cos(theta), np.sin(theta)
    u, v = (yy - cy) * c + (xx - cx) * s_, -(yy - cy) * s_ + (xx - cx) * c
    return torch.from_numpy((u / a) ** 2 + (v / b) ** 2 <= 1.0)


def gen_losses():
    print("[losses] FeatureConsistencyLoss, EllipticalShapeLoss (SURVEY 8f row 3) + FeatureFusion's resize / region-map branches")
    out = {}
    # ---- FeatureConsistencyLoss (model/unet/feature_loss.py:88-125): (B, N, D) features, (B, N) labels ----
    for tag, (B, N, D, margin, scale) in {"fc_a": (2, 64, 64, 1.0, 0.1), "fc_b": (3, 1024, 32, 2.5, 0.3), "fc_c": (1, 7, 20, 0.5, 1.0)}.items():
        fu = torch.from_numpy(O.formula_normal(f"loss/{tag}/u", (B, N, D), seed=1)) * scale
        fg = fu + torch.from_numpy(O.formula_normal(f"loss/{tag}/g", (B, N, D), seed=2)) * scale * 0.5
        y = torch.from_numpy(O.formula_labels(f"loss/{tag}/y", (B, N), 2, seed=3))
        fg[0, 0] = fu[0, 0]                                   # an identical pair: dist = sqrt(1e-8)
        ref = RefFeatLoss(margin=margin)(fu, fg, y)
        check(tag, O.feature_consistency_loss(fu, fg, y, margin).reshape(1), ref.reshape(1), tol=1e-6 * max(1.0, float(ref)))
        out[tag] = np.float32(float(ref))
        print(f"   {tag}: loss {float(ref):.6f}")
    # ---- EllipticalShapeLoss (model/unet/shape_loss.py:17-180) ----
    H, W = 96, 128
    masks = [[ellipse_mask(H, W, 40, 50, 25, 12, 0.4), ellipse_mask(H, W, 70, 100, 8, 15, -0.9), torch.zeros(H, W, dtype=torch.bool)],
             [ellipse_mask(H, W, 30, 30, 3, 1, 0.0),                      # 9 pixels: skipped (< 10)
              torch.from_numpy(np.pad(np.ones((20, 30), bool), ((10, 66), (40, 58)))),          # a rectangle
              ellipse_mask(H, W, 60, 64, 30, 30, 0.0)]]
    ref = RefShapeLoss(epsilon=1e-6)(None, object_masks_list=masks)
    check("shape_masks", O.elliptical_shape_loss(None, masks, 1e-6).reshape(1), torch.as_tensor(ref).reshape(1), tol=1e-5)
    out["shape_masks"] = np.float32(float(ref))
    out["shape_masks_in"] = np.stack([np.stack([m.numpy() for m in img]) for img in masks]).astype(np.uint8)
    print(f"   shape (mask list): loss {float(ref):.6f}")
    # from probabilities: class-1 arg-max region of each image as one object (:61-98); image 2 has no foreground, image 3 a tiny one
    probs = torch.zeros(4, 3, H, W)
    fg = [ellipse_mask(H, W, 48, 64, 35, 18, 0.7) | ellipse_mask(H, W, 20, 20, 6, 6, 0.0), ellipse_mask(H, W, 50, 60, 20, 20, 0.0),
          torch.zeros(H, W, dtype=torch.bool), ellipse_mask(H, W, 10, 10, 1.5, 1.5, 0.0)]
    noise = torch.from_numpy(O.formula_uniform("loss/shape/p", (4, 3, H, W), 0.0, 0.2, seed=4))
    for b in range(4):
        probs[b, 0] = 0.5 + noise[b, 0]
        probs[b, 1] = torch.where(fg[b], torch.tensor(0.9), torch.tensor(0.1)) + noise[b, 1]
        probs[b, 2] = 0.3 + noise[b, 2]
    probs = probs / probs.sum(1, keepdim=True)
    ref = RefShapeLoss(epsilon=1e-6)(probs)
    check("shape_probs", O.elliptical_shape_loss(probs, None, 1e-6).reshape(1), torch.as_tensor(ref).reshape(1), tol=1e-5)
    out["shape_probs"] = np.float32(float(ref))
    out["shape_probs_in"] = probs.numpy()
    ref1 = RefShapeLoss()(probs[:, :1])                       # a single class: 0 (:63-64)
    assert float(ref1) == 0.0 and float(O.elliptical_shape_loss(probs[:, :1])) == 0.0
    print(f"   shape (probabilities): loss {float(ref):.6f}")
    # ---- FeatureFusion: bilinear resize of multi-scale F_u (:69-76) and of a per-pixel F_g (:140-144), region map with -1 (:84-138)
    fu0 = torch.from_numpy(O.formula_normal("loss/ff/u0", (2, 8, 24, 40), seed=5))
    fu1 = torch.from_numpy(O.formula_normal("loss/ff/u1", (2, 16, 12, 20), seed=6))
    fu2 = torch.from_numpy(O.formula_normal("loss/ff/u2", (2, 4, 7, 9), seed=7))
    fg4 = torch.from_numpy(O.formula_normal("loss/ff/g4", (2, 12, 5, 11), seed=8))
    ref = RefFusion([8, 16, 4], 12)([fu0, fu1, fu2], fg4)
    check("ff_multi", O.feature_fusion_full([fu0, fu1, fu2], fg4, 12), ref, tol=1e-6)
    out["ff_multi"] = ref.numpy()
    ref = RefFusion([8, 16, 4], 12)([fu0, fu1, fu2], fg4, target_spatial_size=(33, 17))     # up AND down scaling, odd target
    check("ff_target", O.feature_fusion_full([fu0, fu1, fu2], fg4, 12, target_spatial_size=(33, 17)), ref, tol=1e-6)
    out["ff_target"] = ref.numpy()
    fg2 = torch.from_numpy(O.formula_normal("loss/ff/g2", (7, 12), seed=9))
    rmap = torch.from_numpy(O.formula_labels("loss/ff/map", (2, 24, 40), 9, seed=10)) - 1    # ids -1 .. 7: -1 and 7 are invalid
    ref = RefFusion([8, 16], 12)([fu0, fu1], fg2, region_to_pixel_map=rmap)
    check("ff_regions", O.feature_fusion_full([fu0, fu1], fg2, 12, region_to_pixel_map=rmap), ref, tol=1e-6)
    out["ff_regions"] = ref.numpy()
    out["ff_regions_map"] = rmap.numpy().astype(np.int64)
    fadd = torch.from_numpy(O.formula_normal("loss/ff/ga", (2, 24, 6, 10), seed=11))
    ref = RefFusion([8, 16], 24, fusion_method="add")([fu0, fu1], fadd)
    check("ff_add", O.feature_fusion_full([fu0, fu1], fadd, 24, method="add"), ref, tol=1e-6)
    out["ff_add"] = ref.numpy()
    save("losses.npz", **out)


def ref_script_symbol(rel_path, name):
    """TVLoss / dice_loss live in script modules whose import needs cv2 (absent).  Neither definition uses cv2: take the
    class / function node out of the reference file's syntax tree and execute just that node with torch, nn, F in scope --
    the reference's own code, run here in the build container only; nothing of it is stored (fixtures hold numbers)."""
    import ast
    import torch.nn as nn
    import torch.nn.functional as F
    src = open(os.path.join(REF, rel_path)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == name)
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=[node], type_ignores=[]), rel_path, "exec"), ns)
    return ns[name]


def gen_scriptlosses():
    print("[scriptlosses] TVLoss (scripts/train_end_to_end.py:73-89), dice_loss (scripts/train_segmentation.py:29-40): values and "
          "gradients from the reference definitions (syntax-tree extraction, see ref_script_symbol)")
    RefTV = ref_script_symbol("scripts/train_end_to_end.py", "TVLoss")
    ref_dice = ref_script_symbol("scripts/train_segmentation.py", "dice_loss")
    out = {}
    for tag, (shape, weight, seed) in {"tv_a": ((2, 1, 17, 23), 1.0, 1), "tv_b": ((3, 2, 64, 48), 0.37, 2), "tv_c": ((1, 1, 2, 2), 1.0, 3),
                                       "tv_d": ((8, 1, 128, 128), 0.1, 4)}.items():
        x = torch.from_numpy(O.formula_normal(f"sloss/{tag}/x", shape, seed=seed)).requires_grad_(True)
        ref = RefTV(weight)(x)
        ref.backward()
        xo = x.detach().clone().requires_grad_(True)
        mine = O.tv_loss(xo, weight)
        mine.backward()
        check(tag, mine.detach().reshape(1), ref.detach().reshape(1), tol=1e-6 * max(1.0, float(ref)))
        check(tag + ".grad", xo.grad, x.grad, tol=1e-7)
        out[tag] = np.float32(float(ref))
        out[tag + "_grad"] = x.grad.numpy()
        print(f"   {tag}: loss {float(ref):.6f}")
    for tag, (B, C, H, W, smooth, scale, seed) in {"dice_a": (2, 2, 16, 16, 1.0, 2.0, 5), "dice_b": (3, 4, 33, 20, 0.5, 1.0, 6),
                                                   "dice_c": (1, 2, 128, 128, 1.0, 3.0, 7), "dice_d": (2, 3, 8, 8, 1e-3, 8.0, 8)}.items():
        lg = (torch.from_numpy(O.formula_normal(f"sloss/{tag}/x", (B, C, H, W), seed=seed)) * scale).requires_grad_(True)
        y = torch.from_numpy(O.formula_labels(f"sloss/{tag}/y", (B, H, W), C, seed=seed + 10))
        if tag == "dice_d":
            y[0] = 0                                         # a class absent from an image: intersection 0, the smooth term decides
        ref = ref_dice(lg, y, smooth)
        ref.backward()
        lo = lg.detach().clone().requires_grad_(True)
        mine = O.dice_loss(lo, y, smooth)
        mine.backward()
        check(tag, mine.detach().reshape(1), ref.detach().reshape(1), tol=1e-6)
        check(tag + ".grad", lo.grad, lg.grad, tol=1e-8)
        out[tag] = np.float32(float(ref))
        out[tag + "_grad"] = lg.grad.numpy()
        if tag == "dice_d":
            out[tag + "_y"] = y.numpy()
        # the trainer's combined loss (scripts/train_segmentation.py:126-133): CrossEntropyLoss + dice_loss, one backward
        l2 = lg.detach().clone().requires_grad_(True)
        tot = torch.nn.CrossEntropyLoss()(l2, y) + ref_dice(l2, y, smooth)
        tot.backward()
        out[tag + "_cedice"] = np.float32(float(tot))
        out[tag + "_cedice_grad"] = l2.grad.numpy()
        print(f"   {tag}: dice {float(ref):.6f}  ce+dice {float(tot):.6f}")
    # ---- gradients of FeatureConsistencyLoss (importable class; the inputs are those of gen_losses) ----
    for tag, (B, N, D, margin, scale) in {"fc_a": (2, 64, 64, 1.0, 0.1), "fc_b": (3, 1024, 32, 2.5, 0.3), "fc_c": (1, 7, 20, 0.5, 1.0)}.items():
        fu = torch.from_numpy(O.formula_normal(f"loss/{tag}/u", (B, N, D), seed=1)) * scale
        fg = fu + torch.from_numpy(O.formula_normal(f"loss/{tag}/g", (B, N, D), seed=2)) * scale * 0.5
        y = torch.from_numpy(O.formula_labels(f"loss/{tag}/y", (B, N), 2, seed=3))
        fg[0, 0] = fu[0, 0] + 1e-3                            # (gen_losses uses an identical pair; its gradient is 0/0-prone: keep it apart here)
        fu.requires_grad_(True), fg.requires_grad_(True)
        ref = RefFeatLoss(margin=margin)(fu, fg, y)
        ref.backward()
        a, b = fu.detach().clone().requires_grad_(True), fg.detach().clone().requires_grad_(True)
        O.feature_consistency_loss(a, b, y, margin).backward()
        check(tag + ".grad_u", a.grad, fu.grad, tol=1e-7)
        check(tag + ".grad_g", b.grad, fg.grad, tol=1e-7)
        idx = sample_idx(f"sloss/{tag}/idx", fu.numel(), 2048)
        out[tag + "_val"] = np.float32(float(ref))
        out[tag + "_idx"] = idx
        out[tag + "_grad_u"] = fu.grad.reshape(-1)[idx].numpy()
        out[tag + "_grad_g"] = fg.grad.reshape(-1)[idx].numpy()
    # ---- one train step with CE + dice (scripts/train_segmentation.py:121-134) on the reference UNet, train mode ----
    cfg = (3, 2, 8, 2)
    p = O.make_unet_params(*cfg, seed=21)
    x = torch.from_numpy(O.formula_normal("sloss/tr/x", (2, 3, 32, 32), seed=22))
    y = torch.from_numpy(O.formula_labels("sloss/tr/y", (2, 32, 32), 2, seed=23))
    m = ref_unet(cfg, p, train=True)
    lg, _, _ = m(x)
    loss = torch.nn.CrossEntropyLoss()(lg, y) + ref_dice(lg, y)
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in m.named_parameters()}
    oloss, og, _, _, _, _ = O.train_step(p, x, y, cfg[3], loss_kind="ce+dice")
    assert abs(float(oloss) - float(loss)) <= 1e-6 * abs(float(loss)), (float(oloss), float(loss))
    names = list(grads.keys())
    for k in names:
        assert float((og[k] - grads[k]).norm()) <= 2e-3 * float(grads[k].norm()) + 1e-7, k
    m64 = ref_unet(cfg, p, train=True).double()
    lg64, _, _ = m64(x.double())
    (torch.nn.CrossEntropyLoss()(lg64, y) + ref_dice(lg64, y)).backward()
    g64 = {k: v.grad.detach() for k, v in m64.named_parameters()}
    out["tr_loss"] = np.float64(float(loss))
    out["tr_names"] = np.array(names)
    out["tr_grad_norms"] = np.array([float(grads[k].norm()) for k in names], dtype=np.float64)
    out["tr_grad_norms64"] = np.array([float(g64[k].norm()) for k in names], dtype=np.float64)
    out["tr_cond"] = np.array([float((grads[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-300)) for k in names])
    out["tr_grad_flat64"] = torch.cat([g64[k].reshape(-1) for k in names]).numpy().astype(np.float32)
    print(f"   train step CE + dice: loss {float(loss):.6f}, fp32-vs-fp64 reference gradient deviation median {np.median(out['tr_cond']):.2e}")
    save("script_losses.npz", **out)


def gen_graph():
    print("[graph] COO index maps: 128^2/p32, 130x140/p32, 512^2/p16, 1024^2/p16, 16x16/p16 (empty)")
    out = {}
    for tag, (H, W, p) in {"g128": (128, 128, 32), "g130": (130, 140, 32), "g512": (512, 512, 16),
                           "g1024": (1024, 1024, 16), "g1": (16, 16, 16), "grow": (16, 80, 16)}.items():
        pgc = RefPGC(p)
        nph, npw = O.patch_grid(H, W, p)
        feats = torch.zeros(nph * npw, 1)
        _, ei = pgc.construct_patch_graph(torch.zeros(1, H, W), feats)
        oe = O.patch_graph_edges(H, W, p)
        assert ei.dtype == torch.int64 and tuple(ei.shape) == oe.shape, (ei.shape, oe.shape)
        assert np.array_equal(ei.numpy(), oe), tag
        print(f"   {tag}: nodes={nph*npw} edges={oe.shape[1]} bit-exact")
        out[tag] = ei.numpy()
    # image_to_patches + patch-mean on a non-divisible image
    img = torch.from_numpy(O.formula_normal("graph/img", (5, 37, 45), seed=2))
    pt, (nph, npw) = RefPGC(16).image_to_patches(img)
    opt, (onh, onw) = O.image_to_patches(img, 16)
    assert (nph, npw) == (onh, onw) and torch.equal(pt, opt)
    out["patches_37x45_mean"] = pt.mean(dim=(2, 3)).numpy()
    save("patch_graph.npz", **out)


def checksums(ts):
    return np.array([[float(t.double().sum()), float(t.double().abs().sum())] for t in ts], dtype=np.float64)


def gen_c1():
    print("[c1] UNet(1,2,32,4) @1x1x256x256 eval")
    cfg = (1, 2, 32, 4)
    p = O.make_unet_params(*cfg, seed=0)
    x = torch.from_numpy(O.formula_normal("c1/x", (1, 1, 256, 256), seed=0))
    with torch.no_grad():
        lg, sk, ft = ref_unet(cfg, p)(x)
        olg, _, _ = O.unet_forward(p, x, 4)
    check("c1.logits", olg, lg)
    idx = sample_idx("c1/idx", lg.numel())
    save("c1.npz", idx=idx, logits=lg.reshape(-1)[idx].numpy(), sums=checksums([lg] + sk + ft))


def gen_c2():
    print("[c2] UNet(3,2,32,4) @8x3x512x512 eval + patch-mean + GAT(32,128,64,4,1)")
    cfg = (3, 2, 32, 4)
    p = O.make_unet_params(*cfg, seed=0)
    m = ref_unet(cfg, p)
    gp = O.make_gat_params(32, 128, 64, 4, 1, seed=0)
    g = RefGAT(32, 128, 64, 4, 1).eval()
    g.load_state_dict(gp)
    ei = torch.from_numpy(O.patch_graph_edges(512, 512, 16))
    out = {}
    sums = []
    for b in range(8):
        x = torch.from_numpy(O.formula_normal(f"c2/x/{b}", (1, 3, 512, 512), seed=1))
        t0 = time.time()
        with torch.no_grad():
            lg, sk, ft = m(x)
            X = O.patch_mean_features(ft[0][0], 16)
            y = g(X, ei)
            if b == 0:
                olg, _, oft = O.unet_forward(p, x, 4)
                check("c2.logits[0]", olg, lg)
                check("c2.gat[0]", O.gat_network_forward(gp, X, ei, 4), y)
        idx = sample_idx(f"c2/idx/{b}", lg.numel(), 1024)
        gidx = sample_idx(f"c2/gidx/{b}", y.numel(), 512)
        out[f"logits_{b}"] = lg.reshape(-1)[idx].numpy()
        out[f"idx_{b}"] = idx
        out[f"gat_{b}"] = y.reshape(-1)[gidx].numpy()
        out[f"gidx_{b}"] = gidx
        sums.append(checksums([lg] + sk + ft + [X, y]))
        print(f"   image {b}: {time.time()-t0:.1f}s  max|logit|={float(lg.abs().max()):.3f} max|gat|={float(y.abs().max()):.3f}")
    out["sums"] = np.stack(sums)
    save("c2.npz", **out)


def gen_c4():
    print("[c4] synthetic 2048-node in-degree-8 graph, GAT(64,128,64,4,1); 3 images of 1024^2 U-Net")
    N, deg = 2048, 8
    u = O.formula_uniform("c4/src", (N * deg,), 0.0, 1.0, 3).astype(np.float64)
    src = np.minimum((u * N).astype(np.int64), N - 1)
    ei = np.stack([src, np.repeat(np.arange(N, dtype=np.int64), deg)])
    X = torch.from_numpy(O.formula_normal("c4/X", (N, 64), seed=3))
    gp = O.make_gat_params(64, 128, 64, 4, 1, seed=0)
    g = RefGAT(64, 128, 64, 4, 1).eval()
    g.load_state_dict(gp)
    with torch.no_grad():
        y = g(X, torch.from_numpy(ei))
        check("c4.gat", O.gat_network_forward(gp, X, torch.from_numpy(ei), 4), y)
    out = {"gat_out": y.numpy()}
    cfg = (3, 2, 32, 4)
    p = O.make_unet_params(*cfg, seed=0)
    m = ref_unet(cfg, p)
    for b in (0, 31):
        x = torch.from_numpy(O.formula_normal(f"c4/x/{b}", (1, 3, 1024, 1024), seed=2))
        t0 = time.time()
        with torch.no_grad():
            lg, _, _ = m(x)
        idx = sample_idx(f"c4/idx/{b}", lg.numel(), 1024)
        out[f"logits_{b}"] = lg.reshape(-1)[idx].numpy()
        out[f"idx_{b}"] = idx
        print(f"   image {b}: {time.time()-t0:.1f}s max|logit|={float(lg.abs().max()):.3f}")
    save("c4.npz", **out)


def gen_c5():
    print("[c5] train step UNet(3,2,32,4) on a 2x3x128x128 shard and a 4x3x512x512 shard: CE + Adam(1e-3, wd 1e-4)")
    cfg = (3, 2, 32, 4)
    out = {}
    for tag, shape in (("s", (2, 3, 128, 128)), ("f", (4, 3, 512, 512))):
        p = O.make_unet_params(*cfg, seed=0)
        x = torch.from_numpy(O.formula_normal(f"c5/{tag}/x", shape, seed=4))
        y = torch.from_numpy(O.formula_labels(f"c5/{tag}/y", (shape[0], shape[2], shape[3]), 2, seed=5))
        m = ref_unet(cfg, p, train=True)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)  # train_segmentation.py:96
        t0 = time.time()
        opt.zero_grad()
        lg, _, _ = m(x)
        loss = torch.nn.CrossEntropyLoss()(lg, y)  # train_segmentation.py:91,127
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in m.named_parameters()}
        opt.step()
        print(f"   [{tag}] reference step {time.time()-t0:.1f}s loss={float(loss):.6f}")
        oloss, og, onew, ostats, _, _ = O.train_step(p, x, y, 4)
        assert abs(float(oloss) - float(loss)) <= 1e-5 * abs(float(loss)), (float(oloss), float(loss))
        sd = m.state_dict()
        names = list(grads.keys())
        gn = []
        for k in names:
            rn = float(grads[k].norm())
            dn = float((og[k] - grads[k]).norm())
            assert dn <= 2e-3 * rn + 1e-7, (k, dn, rn)
            dp = float((onew[k] - sd[k]).abs().max())
            assert dp <= 2e-5, (k, dp)  # Adam normalises: |step| ~ lr
            gn.append(rn)
        for k, v in ostats.items():
            assert float((v - sd[k]).abs().max()) <= 1e-5, k
        # the same step with the reference module in float64: train-mode BatchNorm + ReLU masks make the
        # fp32 gradients ill-conditioned (a 1e-7 perturbation flips masks), so the fixture records how far
        # the reference's OWN fp32 gradients sit from its fp64 gradients; tests scale their tolerance by it
        m64 = ref_unet(cfg, p, train=True).double()
        lg64, _, _ = m64(x.double())
        loss64 = torch.nn.CrossEntropyLoss()(lg64, y)
        loss64.backward()
        g64 = {k: v.grad.detach() for k, v in m64.named_parameters()}
        cond = np.array([float((grads[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-300)) for k in names])
        print(f"   [{tag}] fp32-vs-fp64 reference gradient deviation: median {np.median(cond):.2e} max(non-bias) "
              f"{max(c for c, k in zip(cond, names) if not (k.endswith('.bias') and 'conv' in k and 'bn' not in k)):.2e}")
        out[f"{tag}_loss64"] = np.float64(float(loss64))
        out[f"{tag}_grad_norms64"] = np.array([float(g64[k].norm()) for k in names], dtype=np.float64)
        out[f"{tag}_cond"] = cond
        out[f"{tag}_loss"] = np.float64(float(loss))
        out[f"{tag}_grad_norms"] = np.array(gn, dtype=np.float64)
        flat_g = torch.cat([grads[k].reshape(-1) for k in names])
        flat_p = torch.cat([sd[k].reshape(-1) for k in names])
        idx = sample_idx(f"c5/{tag}/idx", flat_g.numel(), 4096)
        out[f"{tag}_idx"] = idx
        out[f"{tag}_grad_s"] = flat_g[idx].numpy()
        out[f"{tag}_grad_s64"] = torch.cat([g64[k].reshape(-1) for k in names])[idx].numpy()
        out[f"{tag}_param_s"] = flat_p[idx].numpy()
        out[f"{tag}_bn_rm"] = sd["encoder.encoder_blocks.0.bn1.running_mean"].numpy()
        out[f"{tag}_bn_rv"] = sd["encoder.encoder_blocks.0.bn1.running_var"].numpy()
        out[f"{tag}_bn_rm_b"] = sd["encoder.bottleneck.bn2.running_mean"].numpy()
        out[f"{tag}_bn_rv_b"] = sd["encoder.bottleneck.bn2.running_var"].numpy()
    out["param_names"] = np.array(names)
    save("c5.npz", **out)



GATDROP_CASES = {   # tag: (cfg (in, hidden, out, heads), layers, N, graph, x scale, weight scale, p)
    "edge10": ((8, 16, 16, 4), 1, 10, "edge10", 1.0, 1.0, 0.1),            # the module's own 10-node example, default dropout 0.1
    "patch":  ((32, 128, 64, 4), 1, 36, "grid6", 1.0, 1.0, 0.1),           # the patch GAT (train_end_to_end.py:144-152), 6 x 6 patches
    "seg":    ((64, 64, 2, 2), 1, 36, "grid6", 0.5, 1.0, 0.1),             # the segment predictor (:46-54): 64 -> K = 2, 2 heads
    "l2h1":   ((8, 16, 8, 1), 2, 10, "edge_iso", 1.0, 1.0, 0.3),           # two layers (concat, then mean), isolated node, p = 0.3
}


class _MaskDropout(torch.nn.Module):
    """Stands in for an nn.Dropout of the reference: the same multiplication, with the mask given instead of drawn."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, t):
        return t * self.mask


def gen_gatdrop():
    """The reference GATNetwork in TRAIN mode (graph_attention.py:97, :160) with its nn.Dropout modules replaced by explicit masks --
    the reference's own forward code, only the random draw made reproducible: every head's `dropout` multiplies the attention
    coefficients by that head's (E, 1) mask, every layer's `dropout` the layer output by its (N, F) mask.  Records outputs and the
    gradients torch autograd gives (loss = sum(out * R))."""
    print("[gatdrop] reference GATNetwork.train() with explicit dropout masks: outputs + gradients")
    out = {}
    for tag, (cfg, layers, N, graph, xs, ws, pdrop) in GATDROP_CASES.items():
        if graph == "edge10":
            ei = EDGE10
        elif graph == "edge_iso":
            ei = np.concatenate([EDGE10, np.array([[10, 10], [0, 3]], dtype=np.int64)], axis=1)
            N = 11
        else:
            side = int(graph[4:])
            ei = O.patch_graph_edges(side * 16, side * 16, 16)
        ei = torch.from_numpy(np.ascontiguousarray(ei))
        E = ei.shape[1]
        X = torch.from_numpy(O.formula_normal(f"gatdrop/{tag}/x", (N, cfg[0]), seed=5)) * xs
        R = torch.from_numpy(O.formula_normal(f"gatdrop/{tag}/r", (N, cfg[2]), seed=6))
        p = O.make_gat_params(cfg[0], cfg[1], cfg[2], cfg[3], layers, seed=5, scale=ws)
        g = RefGAT(cfg[0], cfg[1], cfg[2], cfg[3], num_gat_layers=layers, dropout_rate=pdrop).train()
        g.load_state_dict(p)
        masks = []
        for l, layer in enumerate(g.gat_layers):
            width = layer.head_out_features * (cfg[3] if layer.concat else 1)
            em = O.dropout_mask_from_uniform(torch.from_numpy(O.formula_uniform(f"gatdrop/{tag}/e{l}", (cfg[3], E), 0.0, 1.0, 7).astype(np.float32)), pdrop)
            om = O.dropout_mask_from_uniform(torch.from_numpy(O.formula_uniform(f"gatdrop/{tag}/o{l}", (N, width), 0.0, 1.0, 8).astype(np.float32)), pdrop)
            masks.append((em, om))
            for k, head in enumerate(layer.heads):
                head.dropout = _MaskDropout(em[k].reshape(-1, 1))      # graph_attention.py:97
            layer.dropout = _MaskDropout(om)                          # :160
            out[f"{tag}_emask{l}"], out[f"{tag}_omask{l}"] = em.numpy(), om.numpy()
        Xr = X.clone().requires_grad_(True)
        y = g(Xr, ei)
        (y * R).sum().backward()
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        Xo = X.clone().requires_grad_(True)
        yo = O.gat_network_forward(q, Xo, ei, cfg[3], layers, masks=masks)
        (yo * R).sum().backward()
        check(tag + ".out", yo.detach(), y.detach(), tol=2e-6 * max(1.0, float(y.abs().max())))
        check(tag + ".dX", Xo.grad, Xr.grad, tol=2e-6 * max(1.0, float(Xr.grad.abs().max())))
        out[tag + "_out"], out[tag + "_dX"], out[tag + "_ei"] = y.detach().numpy(), Xr.grad.numpy(), ei.numpy()
        for k, v in g.named_parameters():
            check(f"{tag}.d{k}", q[k].grad, v.grad, tol=2e-6 * max(1.0, float(v.grad.abs().max())))
            out[f"{tag}_d_{k}"] = v.grad.numpy()
        kept = float(sum(float((m[0] > 0).float().mean()) for m in masks) / len(masks))
        print(f"   {tag}: N {N}, E {E}, p {pdrop}, kept {kept*100:.1f} % of the coefficients, max|out| {float(y.abs().max()):.3f}")
    save("gat_dropout.npz", **out)


def gen_bf16ref():
    """The REFERENCE U-Net run in bfloat16 on the CPU (module.to(torch.bfloat16), bfloat16 input: every parameter, activation and
    the oneDNN accumulate-then-round path in bf16) against its own fp32 run at the same formula weights: the deviation a bf16
    implementation of this network has BY ITSELF.  tests/test_gpu_bf16.py holds the HIP bf16-storage mode to
    1.25 x these statistics (its storage is coarser in one respect only: none -- weights and activations are bf16 there too, the
    accumulation is fp32 in both), so the bf16 tolerance is pinned by the reference, not argued."""
    print("[bf16ref] reference UNet in bfloat16 vs its own fp32 run")
    out = {}
    cases = [("b", (3, 3, 8, 2), (2, 3, 37, 45), "tiny/b/x", 11, 11), ("c", (3, 2, 8, 3), (2, 3, 64, 48), "tiny/c/x", 11, 11),
             ("c2_0", (3, 2, 32, 4), (1, 3, 512, 512), "c2/x/0", 1, 0)]
    for tag, cfg, shape, xname, xseed, pseed in cases:
        p = O.make_unet_params(*cfg, seed=pseed)
        x = torch.from_numpy(O.formula_normal(xname, shape, seed=xseed))
        with torch.no_grad():
            lf = ref_unet(cfg, p)(x)[0]
            m16 = ref_unet(cfg, p).to(torch.bfloat16)
            lb = m16(x.to(torch.bfloat16))[0].float()
        d = (lb - lf).abs()
        scale = float(lf.abs().max())
        agree = float((lb.argmax(1) == lf.argmax(1)).float().mean())
        flat = d.reshape(-1)
        p999 = float(torch.quantile(flat[:: max(1, flat.numel() // 2_000_000)], 0.999))
        stats = np.array([float(d.max()), float(d.mean()), p999, agree, scale], dtype=np.float64)
        print(f"   {tag}: max-abs {stats[0]:.4e} ({stats[0]/scale*100:.2f} % of max|logit| {scale:.3f}), mean-abs {stats[1]:.3e} "
              f"({stats[1]/scale*100:.3f} %), p99.9 {p999:.3e}, argmax agreement {agree*100:.2f} %")
        out[f"{tag}_stats"] = stats                      # [max-abs, mean-abs, 99.9th percentile, argmax agreement, max|fp32 logit|]
        idx = sample_idx(f"bf16ref/{tag}/idx", lf.numel(), 2048)
        out[f"{tag}_idx"] = idx
        out[f"{tag}_fp32"] = lf.reshape(-1)[idx].numpy()
        out[f"{tag}_bf16"] = lb.reshape(-1)[idx].numpy()
    save("bf16_reference.npz", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="tiny,gat,gatgrad,graph,mincut,mincutgrad,region,dethead,losses,scriptlosses,c1,c2,c4,c5,bf16ref,gatdrop")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    fns = {"tiny": gen_tiny, "gat": gen_gat, "gatgrad": gen_gatgrad, "graph": gen_graph, "mincut": gen_mincut, "mincutgrad": gen_mincutgrad, "region": gen_region, "dethead": gen_dethead, "losses": gen_losses, "scriptlosses": gen_scriptlosses, "c1": gen_c1, "c2": gen_c2, "c4": gen_c4, "c5": gen_c5, "bf16ref": gen_bf16ref, "gatdrop": gen_gatdrop}
    for k in a.only.split(","):
        t0 = time.time()
        fns[k]()
        print(f"   ({time.time()-t0:.1f}s)")


if __name__ == "__main__":
    main()
