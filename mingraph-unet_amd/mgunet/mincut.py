"""Host-side mirror of the MinCut stage of the patch-graph branch, routed through libmgunet.so:

* `MinCutRefinement` -- model/graph_partition/mincut_refinement.py:5-205: same constructor, `compute_edge_weights_for_ncut`,
  `normalized_cut_loss`, `forward(features, edge_index, K, segment_predictor_network)` -> `(loss, soft_assignments)`;
* `PatchSegmentPredictor` -- scripts/train_end_to_end.py:40-70: same constructor and state_dict() keys
  (`gnn_predictor.gat_layers...` or `mlp_predictor.{0,2}.{weight,bias}`).

Forward (inference) semantics only: the loss is computed on the HIP path and returned as a device scalar without an
autograd graph (the reference's e2e training loop that would differentiate it does not run: SURVEY appendix A).
The reference sums the weighted degree over the SOURCE index of the COO list (:96), so the kernels take a CSR by
source; it is derived once per edge_index tensor and cached, like the GAT's CSR by target.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .gat import GATNetwork, _context, coo_to_csr_device


def _check_features(x: torch.Tensor, what: str) -> None:
    if not isinstance(x, torch.Tensor) or x.dim() != 2:
        raise ValueError(f"{what} must be a (N, D) tensor")
    if not x.is_cuda:
        raise RuntimeError("mgunet MinCut runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
    if x.dtype != torch.float32:
        raise TypeError(f"expected float32 {what}, got {x.dtype}")


class PatchSegmentPredictor(nn.Module):
    """train_end_to_end.py:40-70: logits (N, num_segments) from the GAT-refined patch features."""

    def __init__(self, in_dim, num_segments, hidden_dim=None, use_gnn=False, num_gnn_layers=1, num_heads=1):
        super().__init__()
        self.use_gnn = use_gnn
        self.in_dim, self.num_segments = in_dim, num_segments
        if use_gnn:
            self.gnn_predictor = GATNetwork(node_feature_dim=in_dim, hidden_dim=hidden_dim if hidden_dim else in_dim,
                                            output_dim=num_segments, num_heads=num_heads, num_gat_layers=num_gnn_layers,
                                            dropout_rate=0.1, alpha=0.2)  # :46-54
        else:
            if hidden_dim is None:
                hidden_dim = in_dim * 2  # :57
            self.mlp_predictor = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, num_segments))

    def _linear(self, x: torch.Tensor, lin: nn.Linear, relu: bool) -> torch.Tensor:
        """y = x W^T + b (optionally ReLU) as a 1x1 convolution over an (N, 1) image: the MFMA implicit-GEMM kernel."""
        N, Cin = x.shape
        Cout = lin.out_features
        if Cin % 4:
            raise ValueError("MLP predictor widths must be multiples of 4 (16-byte NHWC pixels)")
        ld = (Cout + 3) // 4 * 4
        out = torch.empty((N, ld), device=x.device, dtype=torch.float32)
        ctx = _context(x.device)
        w = lin.weight.detach().contiguous()
        b = lin.bias.detach().contiguous()
        with torch.cuda.device(x.device):
            rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, x.data_ptr(), 1, N, 1, Cin, w.data_ptr(), b.data_ptr(), None, None, Cout, 1,
                                            1 if relu else 0, out.data_ptr(), ld, 0, _lib.current_stream_ptr(x.device))
        _lib.check(rc, ctx.handle)
        return out[:, :Cout]

    def forward(self, x, edge_index=None):
        _check_features(x, "x")
        if self.use_gnn:
            if edge_index is None:
                raise ValueError("edge_index must be provided for GNN-based segment predictor.")  # :66-67
            return self.gnn_predictor(x, edge_index)
        if self.training:
            raise RuntimeError("the HIP path implements the predictor's forward only: call .eval()")
        h = self._linear(x.detach().contiguous(), self.mlp_predictor[0], relu=True)
        return self._linear(h.contiguous(), self.mlp_predictor[2], relu=False).contiguous()


class MinCutRefinement(nn.Module):
    """mincut_refinement.py:5-205.  The constructor arguments parameterise the energy E(S) of a solver the reference
    never implements (:9-16); like there, they are stored and unused."""

    def __init__(self, gamma_unet_priors=0.5, sigma_intensity=10.0, sigma_features=1.0):
        super().__init__()
        self.gamma_unet_priors = gamma_unet_priors
        self.sigma_intensity = sigma_intensity
        self.sigma_features = sigma_features
        self._csr = None

    def _csr_by_source(self, edge_index: torch.Tensor, N: int, dev):
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, N, str(dev))
        if self._csr is None or self._csr[0] != key:
            rowptr, col = coo_to_csr_device(edge_index.to(dev).flip(0), N)   # rows = sources, col = targets
            self._csr = (key, rowptr, col, edge_index)
        return self._csr[1], self._csr[2]

    def compute_edge_weights_for_ncut(self, node_features, edge_index):
        """(E,) weights exp(-|f_i - f_j|^2 / 2) in edge order (:30-52)."""
        _check_features(node_features, "node_features")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be an int64 (2, E) tensor")
        dev = node_features.device
        f = node_features.detach().contiguous()
        ei = edge_index.to(dev).contiguous()
        N, D = f.shape
        E = ei.shape[1]
        if E:
            # the reference raises on an out-of-range index, so this check has to complete before the call returns:
            # ONE reduction and ONE host read (the kernel itself never reads out of range: it trusts this check)
            lo, hi = torch.stack(torch.aminmax(ei)).tolist()
            if lo < 0 or hi >= N:
                raise IndexError(f"edge_index values must be in [0, {N}); got [{lo}, {hi}]")
        w = torch.empty(E, device=dev, dtype=torch.float32)
        ctx = _context(dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_ncut_edge_weights(ctx.handle, f.data_ptr(), N, D, ei.data_ptr() if E else None, E,
                                                  w.data_ptr() if E else None, _lib.current_stream_ptr(dev))
        _lib.check(rc, ctx.handle)
        return w

    def _ncut(self, node_features, edge_index, assign, K, is_logits):
        _check_features(node_features, "node_features")
        dev = node_features.device
        f = node_features.detach().contiguous()
        N, D = f.shape
        a = assign.detach().to(torch.float32).contiguous()
        rowptr, col = self._csr_by_source(edge_index, N, dev)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        soft = torch.empty((N, K), device=dev, dtype=torch.float32) if is_logits else None
        hard = torch.empty(N, device=dev, dtype=torch.int32) if is_logits else None
        ctx = _context(dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_ncut_forward(ctx.handle, f.data_ptr(), N, D, rowptr.data_ptr(), col.data_ptr() if col.numel() else None,
                                             col.numel(), a.data_ptr(), K, 1 if is_logits else 0,
                                             soft.data_ptr() if is_logits else None, hard.data_ptr() if is_logits else None,
                                             loss.data_ptr(), _lib.current_stream_ptr(dev))
        _lib.check(rc, ctx.handle)
        return loss[0], soft, hard

    def forward_batched(self, patch_features, edge_index_single, B, num_expected_segments, segment_logits):
        """The per-image loop of train_end_to_end.py:347-356 for B images that share one patch-graph topology:
        patch_features (B*Np, D), segment_logits (B*Np, K) -> (losses (B,), soft (B*Np, K), hard labels (B*Np,) int64).
        The loss is a quotient per image, so each image is one mgu_ncut_forward call on its slice."""
        N = patch_features.size(0) // B
        losses, softs, hards = [], [], []
        for b in range(B):
            l, s, h = self._ncut(patch_features[b * N:(b + 1) * N], edge_index_single, segment_logits[b * N:(b + 1) * N],
                                 num_expected_segments, True)
            losses.append(l), softs.append(s), hards.append(h)
        return torch.stack(losses), torch.cat(softs, 0), torch.cat(hards, 0).to(torch.int64)

    def normalized_cut_loss(self, node_features, edge_index, segment_assignments_soft, num_segments_k):
        """sum_k cut(A_k, V \\ A_k) / assoc(A_k, V) with soft assignments (:55-160)."""
        N = node_features.size(0)
        if tuple(segment_assignments_soft.shape) != (N, num_segments_k):
            raise ValueError("segment_assignments_soft shape mismatch.")  # :73-74
        return self._ncut(node_features, edge_index, segment_assignments_soft, num_segments_k, False)[0]

    def forward(self, gat_refined_patch_features, patch_graph_edge_index, num_expected_segments, segment_predictor_network=None):
        """(L_partition, soft segment assignments (N, K)): :163-205.  The arg-max labels of train_end_to_end.py:356 come
        out of the same kernel and are kept in `self.last_hard_labels`."""
        if segment_predictor_network is None:
            raise ValueError("segment_predictor_network is required to get segment assignments for Ncut loss.")  # :183-186
        logits = segment_predictor_network(gat_refined_patch_features, patch_graph_edge_index)  # :190
        N = gat_refined_patch_features.size(0)
        if tuple(logits.shape) != (N, num_expected_segments):
            raise ValueError("segment_assignments_soft shape mismatch.")
        loss, soft, hard = self._ncut(gat_refined_patch_features, patch_graph_edge_index, logits, num_expected_segments, True)
        self.last_hard_labels = hard.to(torch.int64)
        return loss, soft
