"""Host-side mirror of the MinCut stage of the patch-graph branch, routed through libmgunet.so:

* `MinCutRefinement` -- model/graph_partition/mincut_refinement.py:5-205: same constructor, `compute_edge_weights_for_ncut`,
  `normalized_cut_loss`, `forward(features, edge_index, K, segment_predictor_network)` -> `(loss, soft_assignments)`;
* `PatchSegmentPredictor` -- scripts/train_end_to_end.py:40-70: same constructor and state_dict() keys
  (`gnn_predictor.gat_layers...` or `mlp_predictor.{0,2}.{weight,bias}`).

The loss is a node of the autograd graph when its inputs require gradients: `loss.backward()` (train_end_to_end.py:472-479)
runs `mgu_ncut_backward` (gradients w.r.t. the segment logits / soft assignments AND the node features, through the edge
weights, as the reference's autograd does) and, for the MLP predictor, the Linear layers' data / weight gradients on the
library's 1x1-convolution backward kernels.  Under `torch.no_grad()` nothing is recorded.
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


def _linear_run(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, relu: bool) -> torch.Tensor:
    """y = x W^T + b (optionally ReLU) as a 1x1 convolution over an (N, 1) image: the MFMA implicit-GEMM kernel.
    Returns the (N, ceil4(Cout)) buffer (pad columns are zero)."""
    N, Cin = x.shape
    Cout = w.shape[0]
    if Cin % 4:
        raise ValueError("MLP predictor widths must be multiples of 4 (16-byte NHWC pixels)")
    ld = (Cout + 3) // 4 * 4
    out = torch.empty((N, ld), device=x.device, dtype=torch.float32)
    ctx = _context(x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, x.data_ptr(), 1, N, 1, Cin, w.data_ptr(), b.data_ptr(), None, None, Cout, 1,
                                        1 if relu else 0, out.data_ptr(), ld, 0, _lib.current_stream_ptr(x.device))
    _lib.check(rc, ctx.handle)
    return out


class _LinearFn(torch.autograd.Function):
    """nn.Linear (+ ReLU) of the MLP segment predictor (train_end_to_end.py:59-63) with its backward on the HIP path:
    dX = dZ W (mgu_conv2d_dgrad_nhwc, k = 1), dW = dZ^T X (mgu_conv2d_wgrad_nhwc), db = column sums of dZ, dZ = dY where Y > 0."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        xc, w, b = x.detach().contiguous(), weight.detach().contiguous(), bias.detach().contiguous()
        y = _linear_run(xc, w, b, relu)
        ctx.relu, ctx.cout = relu, w.shape[0]
        ctx.save_for_backward(xc, w, y)
        return y[:, :w.shape[0]]

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        dev = x.device
        N, Cin = x.shape
        Cout, ld = ctx.cout, y.shape[1]
        dz = torch.zeros((N, ld), device=dev, dtype=torch.float32)
        dz[:, :Cout] = gy
        c = _context(dev)
        L = _lib.lib()
        st = _lib.current_stream_ptr(dev)
        dx = torch.empty((N, Cin), device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = None
        with torch.cuda.device(dev):
            if ctx.relu:
                _lib.check(L.mgu_relu_backward(c.handle, dz.data_ptr(), y.data_ptr(), dz.numel(), dz.data_ptr(), st), c.handle)
            if dx is not None:
                _lib.check(L.mgu_conv2d_dgrad_nhwc(c.handle, dz.data_ptr(), w.data_ptr(), 1, N, 1, Cin, Cout, 1, dx.data_ptr(), Cin, st), c.handle)
            if dw is not None:
                _lib.check(L.mgu_conv2d_wgrad_nhwc(c.handle, x.data_ptr(), Cin, dz.data_ptr(), 1, N, 1, Cin, Cout, 1, dw.data_ptr(), st), c.handle)
            if ctx.needs_input_grad[2]:
                dbp = torch.empty(ld, device=dev, dtype=torch.float32)
                _lib.check(L.mgu_channel_sum_nhwc(c.handle, dz.data_ptr(), ld, N, ld, dbp.data_ptr(), st), c.handle)
                db = dbp[:Cout]
        return dx, dw, db, None


class _NcutFn(torch.autograd.Function):
    """normalized_cut_loss / MinCutRefinement.forward as one autograd node: forward mgu_ncut_forward, backward mgu_ncut_backward."""

    @staticmethod
    def forward(ctx, feats, assign, mod, edge_index, K, is_logits):
        loss, soft, hard = mod._ncut_run(feats, edge_index, assign, K, is_logits)
        P = soft if is_logits else assign.detach().to(torch.float32).contiguous()
        ctx.mod, ctx.edge_index, ctx.K, ctx.is_logits = mod, edge_index, K, is_logits
        ctx.save_for_backward(feats.detach().contiguous(), P)
        if is_logits:
            ctx.mark_non_differentiable(hard)
            return loss, soft, hard
        return loss

    @staticmethod
    def backward(ctx, gloss, gsoft=None, ghard=None):
        f, P = ctx.saved_tensors
        dev = f.device
        N, D = f.shape
        K = ctx.K
        rp_s, col_t = ctx.mod._csr_by_source(ctx.edge_index, N, dev)
        rp_t, col_s = ctx.mod._csr_by_target(ctx.edge_index, N, dev)
        E = col_t.numel()
        gl = gloss.detach().to(torch.float32).reshape(1).contiguous()
        gs = gsoft.detach().to(torch.float32).contiguous() if (ctx.is_logits and gsoft is not None) else None
        dA = torch.empty((N, K), device=dev, dtype=torch.float32)
        dF = torch.empty((N, D), device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        c = _context(dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_ncut_backward(c.handle, f.data_ptr(), N, D, rp_s.data_ptr(), col_t.data_ptr() if E else None,
                                              rp_t.data_ptr(), col_s.data_ptr() if E else None, E, P.data_ptr(), K,
                                              1 if ctx.is_logits else 0, gl.data_ptr(), gs.data_ptr() if gs is not None else None,
                                              dA.data_ptr(), dF.data_ptr() if dF is not None else None, _lib.current_stream_ptr(dev))
        _lib.check(rc, c.handle)
        return dF, (dA if ctx.needs_input_grad[1] else None), None, None, None, None


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
        if torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad or lin.bias.requires_grad):
            return _LinearFn.apply(x, lin.weight, lin.bias, relu)
        return _linear_run(x.detach().contiguous(), lin.weight.detach().contiguous(), lin.bias.detach().contiguous(), relu)[:, :lin.out_features]

    def forward(self, x, edge_index=None):
        _check_features(x, "x")
        if self.use_gnn:
            if edge_index is None:
                raise ValueError("edge_index must be provided for GNN-based segment predictor.")  # :66-67
            return self.gnn_predictor(x, edge_index)
        h = self._linear(x, self.mlp_predictor[0], relu=True)   # no dropout in the MLP branch: train == eval
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

    def _csr_by_target(self, edge_index: torch.Tensor, N: int, dev):
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, N, str(dev))
        ent = self.__dict__.get("_csr_t")
        if ent is None or ent[0] != key:
            rowptr, col = coo_to_csr_device(edge_index.to(dev), N)   # rows = targets, col = sources (the backward's second gather)
            ent = self.__dict__["_csr_t"] = (key, rowptr, col, edge_index)
        return ent[1], ent[2]

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
        if torch.is_grad_enabled() and (node_features.requires_grad or assign.requires_grad):
            if node_features.shape[1] > 1024:   # refused here, not in the middle of loss.backward()
                raise ValueError(f"mgu_ncut_backward keeps a feature row in registers: D <= 1024, got {node_features.shape[1]}")
            r = _NcutFn.apply(node_features, assign, self, edge_index, K, is_logits)
            return r if is_logits else (r, None, None)
        return self._ncut_run(node_features, edge_index, assign, K, is_logits)

    def _ncut_run(self, node_features, edge_index, assign, K, is_logits):
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
