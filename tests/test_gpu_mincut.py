"""MinCut stage of the patch-graph branch on the HIP path: `mgunet.PatchSegmentPredictor` +
`mgunet.MinCutRefinement` (mgu_gat_layer_forward / mgu_conv2d_nhwc / mgu_ncut_forward / mgu_ncut_edge_weights)
against the fixtures produced by the reference's own classes and against the oracle.  Tolerances: assignments and
edge weights 1e-5 abs, loss 2e-5 relative (fp32 sums in a different order than the reference's scatter_add)."""
import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O
from test_oracle_golden import MINCUT_CASES, mincut_case

pytestmark = pytest.mark.gpu


def build_predictor(cuda, D, K, hidden, use_gnn, heads, params):
    m = mgunet.PatchSegmentPredictor(D, K, hidden_dim=hidden, use_gnn=use_gnn, num_gnn_layers=1, num_heads=heads)
    missing = m.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys   # same state_dict keys as train_end_to_end.py:40-60
    return m.to(cuda).eval()


@pytest.mark.parametrize("tag", list(MINCUT_CASES))
def test_mincut_forward_vs_reference_fixture(cuda, golden, tag):
    g, X, ei, K, p, use_gnn, heads, hidden, shift = mincut_case(golden, tag)
    pred = build_predictor(cuda, X.shape[1], K, hidden, use_gnn, heads, p)
    Xd, eid = X.to(cuda), ei.to(cuda)
    net = pred if shift is None else (lambda x, e: pred(x, e) + shift.to(cuda))
    mc = mgunet.MinCutRefinement()
    with torch.no_grad():   # the parameters require gradients: without this the outputs are autograd nodes, as in the reference
        loss, soft = mc(Xd, eid, K, net)
    ref_loss = float(g[f"{tag}_loss"])
    assert tuple(soft.shape) == (X.shape[0], K) and loss.dim() == 0
    assert np.abs(soft.cpu().numpy() - g[f"{tag}_soft"]).max() <= 1e-5
    assert abs(float(loss) - ref_loss) <= 2e-5 * max(1.0, ref_loss)
    w = mc.compute_edge_weights_for_ncut(Xd, eid)
    assert np.abs(w.cpu().numpy() - g[f"{tag}_w"]).max() <= 1e-6
    # hard labels (train_end_to_end.py:356): equal wherever the reference's margin is not a rounding tie
    rs = torch.from_numpy(g[f"{tag}_soft"])
    top2 = rs.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(mc.last_hard_labels.cpu()[sure], rs.argmax(1)[sure]) and int(sure.sum()) > 0.9 * len(sure)
    # normalized_cut_loss called directly with the reference's soft assignments (:55-160)
    l2 = mc.normalized_cut_loss(Xd, eid, rs.to(cuda), K)
    assert abs(float(l2) - ref_loss) <= 2e-5 * max(1.0, ref_loss)


def test_mincut_interface_errors(cuda):
    mc = mgunet.MinCutRefinement(gamma_unet_priors=0.5, sigma_intensity=10.0, sigma_features=1.0)
    X = torch.zeros(8, 16, device=cuda)
    ei = torch.tensor([[0, 1], [1, 0]], device=cuda)
    with pytest.raises(ValueError, match="segment_predictor_network is required"):
        mc(X, ei, 2, None)                                                      # mincut_refinement.py:183-186
    with pytest.raises(ValueError, match="shape mismatch"):
        mc.normalized_cut_loss(X, ei, torch.zeros(8, 3, device=cuda), 2)        # :73-74
    gnn = mgunet.PatchSegmentPredictor(16, 2, use_gnn=True, num_heads=1).to(cuda).eval()
    with pytest.raises(ValueError, match="edge_index must be provided"):
        gnn(X)                                                                  # train_end_to_end.py:66-67
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mc.normalized_cut_loss(X.cpu(), ei.cpu(), torch.zeros(8, 2), 2)
    # a graph without edges: every association is 0, every segment is skipped (:152-153)
    assert float(mc.normalized_cut_loss(X, torch.zeros((2, 0), dtype=torch.int64, device=cuda),
                                        torch.full((8, 2), 0.5, device=cuda), 2)) == 0.0


@pytest.mark.parametrize("K", [1, 2, 5, 16])
def test_ncut_uniform_assignment_property_at_batch64_size(cuda, K):
    """Size-independent property on the 64-image patch graph of BASELINE configs[2] (65 536 nodes, 253 952 edges, one
    block-diagonal edge list): with P_ik = 1/K for every node, cut_k = (1/K)(1 - 1/K) sum_e w_e and
    assoc_k = (1/K) sum_e w_e, so the loss is exactly K - 1 whatever the features are."""
    one = torch.from_numpy(O.patch_graph_edges(512, 512, 16))
    ei = torch.cat([one + 1024 * b for b in range(64)], dim=1).to(cuda)
    gen = torch.Generator(device=cuda).manual_seed(K)
    X = torch.randn((64 * 1024, 64), device=cuda, generator=gen) * 0.15
    mc = mgunet.MinCutRefinement()
    loss = mc.normalized_cut_loss(X, ei, torch.full((64 * 1024, K), 1.0 / K, device=cuda), K)
    assert abs(float(loss) - (K - 1)) <= 1e-4 * max(1, K - 1)
    # and the weights of the two directions of every patch-graph edge are equal and in (0, 1]
    w = mc.compute_edge_weights_for_ncut(X, ei)
    assert float(w.min()) > 0.0 and float(w.max()) <= 1.0
    assert torch.equal(w[0::2], w[1::2])   # construct_patch_graph appends (n -> m), (m -> n) pairs (:80-92)
