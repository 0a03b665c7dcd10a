"""Host-side mirror of model/gat/graph_attention.py routed through libmgunet.so.

Same classes, constructor signatures and state_dict() keys (`gat_layers.{l}.heads.{h}.W.weight`,
`...a.weight`).  forward(node_features, edge_index) takes the reference's COO int64 (2,E) edge_index;
the CSR-by-target the HIP kernels consume is derived once per edge_index tensor and cached.
All heads of a layer run in ONE kernel sequence (the reference loops over heads in Python,
graph_attention.py:151).  Train mode with dropout_rate > 0 (the reference's default 0.1: nn.Dropout on every head's
attention coefficients, :97, and on the layer output, :160) draws its masks on the device from the library's own
counter-based generator (mgu_dropout_mask; `mgunet.gat.seed_dropout(seed)` restarts the stream) -- or takes them from
`layer.dropout_masks = (edge_masks (H, E) in COO order, out_mask (N, F_out))`, the hook through which a test feeds the
same draw to the reference (tests/golden/gat_dropout.npz) -- and runs mgu_gat_layer_forward_train / _backward_train.
The layers are differentiable: when a parameter or the node features require grad, the call becomes a
torch.autograd.Function whose backward is mgu_gat_layer_backward (dX, dW, da per head), so the graph
branch trains under loss.backward() as in scripts/train_end_to_end.py:219-226, :478.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def coo_to_csr_device(edge_index: torch.Tensor, num_nodes: int):
    """Stable COO -> CSR-by-target on the tensor's device (mgu_coo_to_csr_device: radix sort by target, callers cache the result).
    Ids outside [0, num_nodes) raise IndexError here, where the reference's h[edge_index[0]] would (graph_attention.py:57): ONE
    synchronisation per new edge_index tensor, none per forward."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must have shape (2, E)")
    if edge_index.dtype != torch.int64:
        raise TypeError("edge_index must be int64 (torch.long) like the reference's")
    if not edge_index.is_cuda:
        raise RuntimeError("mgunet GAT runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
    dev = edge_index.device
    ei = edge_index.contiguous()
    E = ei.shape[1]
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    ctx = _context(dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mgu_coo_to_csr_device(ctx.handle, ei.data_ptr() if E else None, E, num_nodes, rowptr.data_ptr(),
                                              col.data_ptr() if E else None, status.data_ptr(), _lib.current_stream_ptr(dev))
    _lib.check(rc, ctx.handle)
    if int(status.item()):
        raise IndexError(f"edge_index values must be in [0, {num_nodes})")
    return rowptr, col


class GraphAttentionLayer(nn.Module):
    """One attention head (graph_attention.py:5-118): parameter holder + single-head forward."""

    def __init__(self, in_features, out_features, dropout_rate, alpha, concat=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.dropout_rate, self.alpha, self.concat = dropout_rate, alpha, concat
        self.W = nn.Linear(in_features, out_features, bias=False)
        self.a = nn.Linear(2 * out_features, 1, bias=False)
        self.leakyrelu = nn.LeakyReLU(self.alpha)
        self.dropout = nn.Dropout(self.dropout_rate)
        nn.init.xavier_uniform_(self.W.weight, gain=1.414)  # :36-37
        nn.init.xavier_uniform_(self.a.weight, gain=1.414)

    def forward(self, node_features, edge_index, graph_ptr=None):
        return _gat_layer_forward([self], node_features, edge_index, True, self.alpha, self.training,
                                  self.dropout_rate, graph_ptr, _csr_cache(self), owner=self, out_dropout=False)


def _csr_cache(mod):
    c = mod.__dict__.get("_mgu_csr_cache")
    if c is None:
        c = {}
        mod.__dict__["_mgu_csr_cache"] = c
    return c


_CTX = {}


def _context(device: torch.device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ctx = _CTX.get(idx)
    if ctx is None:
        ctx = _CTX[idx] = _lib.Context(idx)
    return ctx


def stacked_head_weights(heads, cache):
    """(H*Fh4, Fin4) W panel and (H, 2*Fh4) a panel of a layer's heads, rebuilt only when a parameter changes.
    Fh4 = the per-head width rounded up to a multiple of 4 (16-byte lanes): the extra output features have zero weights
    and zero attention coefficients, so they are exactly ELU(0) = 0 and the caller slices them off."""
    sig = tuple((h.W.weight.data_ptr(), h.W.weight._version, h.a.weight.data_ptr(), h.a.weight._version) for h in heads)
    ent = cache.get("weights")
    if ent is None or ent[0] != sig:
        Fh = heads[0].out_features
        pad = (-Fh) % 4
        Ws = [F.pad(h.W.weight.detach(), (0, 0, 0, pad)) for h in heads]
        As = [torch.cat([F.pad(h.a.weight.detach()[:, :Fh], (0, pad)), F.pad(h.a.weight.detach()[:, Fh:], (0, pad))], 1) for h in heads]
        W = torch.cat(Ws, 0)
        a = torch.cat(As, 0).contiguous()
        if W.shape[1] % 4:
            W = F.pad(W, (0, 4 - W.shape[1] % 4))
        ent = cache["weights"] = (sig, W.contiguous(), a)
    return ent[1], ent[2]


class _Prepared:
    """A mgu_gat_weights handle (the layer's weight-only preparation) tied to the weight versions it was built from."""

    def __init__(self, ctx, handle, sig):
        self.ctx, self.handle, self.sig = ctx, handle, sig

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().mgu_gat_release(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:
            pass


def prepared_head_weights(heads, cache, ctx, dev, has_edges: bool):
    """mgu_gat_prepare once per (weight versions, device, has_edges): W^T a rows + fragment-order W^T (or the GEMM panel)."""
    import ctypes as C
    W, a = stacked_head_weights(heads, cache)
    sig = (cache["weights"][0], str(dev), bool(has_edges))
    ent = cache.get("prepared")
    if ent is None or ent.sig != sig:
        Fh = (heads[0].out_features + 3) // 4 * 4
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgu_gat_prepare(ctx.handle, W.data_ptr(), a.data_ptr(), len(heads), Fh, W.shape[1], 1 if has_edges else 0,
                                                  C.byref(h), _lib.current_stream_ptr(dev)), ctx.handle)
        ent = cache["prepared"] = _Prepared(ctx, h, sig)
    return ent.handle


_DROPOUT = {"seed": 0x6D67756E6574, "stream": 0}   # the library's dropout generator: one stream id per mask drawn


def seed_dropout(seed: int) -> None:
    """Restart the device-side dropout generator (train-mode GAT masks): the same seed gives the same masks call for call."""
    _DROPOUT["seed"], _DROPOUT["stream"] = int(seed) & (2 ** 64 - 1), 0


def _rank_key() -> int:
    """(rank of a data-parallel run, 0 otherwise): folded into the Philox key so that the ranks of one job draw DIFFERENT masks on
    their different shards from the same seed (torch's per-process generators differ the same way)."""
    try:
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    except Exception:
        return 0


def _draw_mask(ctx, dev, shape, p: float) -> torch.Tensor:
    """nn.Dropout's mask (0 or 1 / (1 - p)) from Philox-4x32-10 on the device: element i of stream s under the key
    seed + 0x9E3779B97F4A7C15 * (rank, device index): ranks and devices of one process group never share a mask sequence."""
    m = torch.empty(shape, device=dev, dtype=torch.float32)
    _DROPOUT["stream"] += 1
    key = (_DROPOUT["seed"] + 0x9E3779B97F4A7C15 * (_rank_key() * 64 + (dev.index or 0))) & (2 ** 64 - 1)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mgu_dropout_mask(ctx.handle, key, _DROPOUT["stream"], m.numel(), float(p), m.data_ptr(),
                                               _lib.current_stream_ptr(dev)), ctx.handle)
    return m


def _gat_layer_forward(heads, X, edge_index, concat, alpha, training, dropout_rate, graph_ptr, cache, owner=None, out_dropout=True):
    if not X.is_cuda:
        raise RuntimeError("mgunet GAT runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
    if X.dtype != torch.float32:
        raise TypeError(f"expected float32 node features, got {X.dtype}")
    if X.dim() != 2:
        raise ValueError("node_features must be (N, F)")
    if heads[0].W.weight.shape[1] != X.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({X.shape[0]}x{X.shape[1]} and "
                           f"{heads[0].W.weight.shape[1]}x{heads[0].out_features})")
    meta = _LayerCall(heads, edge_index, concat, alpha, graph_ptr, cache)
    params = [t for h in heads for t in (h.W.weight, h.a.weight)]
    if training and dropout_rate > 0:
        meta.dropout_rate = float(dropout_rate)
        meta.injected = getattr(owner, "dropout_masks", None) if owner is not None else None
        meta.out_dropout = out_dropout   # a lone GraphAttentionLayer drops coefficients only (:97); :160 belongs to the multi-head layer
        return _GatLayerTrainFn.apply(X, meta, *params)
    if torch.is_grad_enabled() and (X.requires_grad or any(t.requires_grad for t in params)):
        # a node of the autograd graph whose backward is mgu_gat_layer_backward (gat_bwd.hip): loss.backward() reaches the GAT
        # parameters as it does in the reference's loop (scripts/train_end_to_end.py:219-226, :478)
        return _GatLayerFn.apply(X, meta, *params)
    return _gat_layer_run(X, meta)[0]


class _LayerCall:
    """The non-tensor arguments of one layer call."""

    def __init__(self, heads, edge_index, concat, alpha, graph_ptr, cache):
        self.heads, self.edge_index, self.concat, self.alpha, self.graph_ptr, self.cache = heads, edge_index, concat, alpha, graph_ptr, cache
        self.dropout_rate, self.injected, self.out_dropout = 0.0, None, True   # train mode (see _GatLayerTrainFn)


def _gat_layer_run(X, m):
    """The forward through libmgunet; returns (out, saved) with `saved` what a backward needs."""
    heads, cache = m.heads, m.cache
    dev = X.device
    N, Fin = X.shape
    H = len(heads)
    Fh_true = heads[0].out_features
    Fh = (Fh_true + 3) // 4 * 4   # narrow heads (e.g. the 2-segment predictor) run zero-padded, see stacked_head_weights
    W, a = stacked_head_weights(heads, cache)
    Xc = X.detach().contiguous()
    if Fin % 4:  # zero-pad K to a multiple of 4: exact (W is padded the same way in stacked_head_weights)
        Xc = F.pad(Xc, (0, 4 - Fin % 4))
    edge_index = m.edge_index
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, N, str(edge_index.device))
    ent = cache.get("csr")
    if ent is None or ent[0] != key:
        rowptr, col = coo_to_csr_device(edge_index.to(dev), N)
        ent = cache["csr"] = (key, rowptr, col, edge_index)  # keep the key tensor alive so data_ptr stays unique
        cache.pop("csr_t", None)
    rowptr, col = ent[1], ent[2]
    G, gp = 1, None
    if m.graph_ptr is not None:
        gp = m.graph_ptr.to(device=dev, dtype=torch.int32).contiguous()
        G = gp.numel() - 1
    out = torch.empty((N, H * Fh if m.concat else Fh), device=dev, dtype=torch.float32)
    ctx = _context(dev)
    if W.device != dev:
        raise RuntimeError(f"GAT parameters are on {W.device}, node features on {dev}")
    handle = prepared_head_weights(heads, cache, ctx, dev, col.numel() > 0)
    with torch.cuda.device(dev):
        rc = _lib.lib().mgu_gat_layer_forward_prepared(ctx.handle, handle, Xc.data_ptr(), N, rowptr.data_ptr(),
                                                       col.data_ptr() if col.numel() else None, col.numel(), gp.data_ptr() if gp is not None else None,
                                                       G, 1 if m.concat else 0, float(m.alpha), out.data_ptr(),
                                                       _lib.current_stream_ptr(dev))
    _lib.check(rc, ctx.handle)
    saved = (Xc, W, a, rowptr, col, gp, G, Fh, Fh_true, Fin)
    if Fh != Fh_true:
        out = out.view(N, -1, Fh)[:, :, :Fh_true].reshape(N, -1).contiguous()
    return out, saved


def _transposed_csr(cache, rowptr, col, N, dev, ctx):
    """CSR by source of the cached CSR by target (mgu_csr_transpose_device), built once per graph."""
    ent = cache.get("csr_t")
    if ent is None or ent[0] is not rowptr:
        E = col.numel()
        rp = torch.empty(N + 1, dtype=torch.int32, device=dev)
        eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        tgt = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgu_csr_transpose_device(ctx.handle, rowptr.data_ptr(), col.data_ptr() if E else None, E, N, rp.data_ptr(),
                                                           eid.data_ptr(), tgt.data_ptr(), _lib.current_stream_ptr(dev)), ctx.handle)
        ent = cache["csr_t"] = (rowptr, rp, eid, tgt)
    return ent[1], ent[2], ent[3]


class _GatLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, X, meta, *params):
        out, saved = _gat_layer_run(X, meta)
        ctx_.meta, ctx_.saved = meta, saved
        return out

    @staticmethod
    def backward(ctx_, gout):
        m = ctx_.meta
        Xc, W, a, rowptr, col, gp, G, Fh, Fh_true, Fin = ctx_.saved
        heads = m.heads
        H, N, dev = len(heads), Xc.shape[0], Xc.device
        g = gout.detach().float()
        if Fh != Fh_true:   # the padded output features never reach the caller: their gradient is 0
            g = F.pad(g.view(N, -1, Fh_true), (0, Fh - Fh_true)).reshape(N, -1)
        g = g.contiguous()
        ctx = _context(dev)
        rps, eid, tgt = _transposed_csr(m.cache, rowptr, col, N, dev, ctx)
        need_x = ctx_.needs_input_grad[0]
        dX = torch.empty_like(Xc) if need_x else None
        dW, da = torch.empty_like(W), torch.empty_like(a)
        E = col.numel()
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_gat_layer_backward(ctx.handle, Xc.data_ptr(), N, Xc.shape[1], rowptr.data_ptr(), col.data_ptr() if E else None, E,
                                                   rps.data_ptr(), eid.data_ptr(), tgt.data_ptr(), gp.data_ptr() if gp is not None else None, G,
                                                   W.data_ptr(), a.data_ptr(), H, Fh, 1 if m.concat else 0, float(m.alpha), g.data_ptr(),
                                                   dX.data_ptr() if need_x else None, dW.data_ptr(), da.data_ptr(),
                                                   _lib.current_stream_ptr(dev))
        _lib.check(rc, ctx.handle)
        grads = []
        for h in range(H):
            grads.append(dW[h * Fh:h * Fh + Fh_true, :Fin].contiguous())
            grads.append(torch.cat([da[h:h + 1, :Fh_true], da[h:h + 1, Fh:Fh + Fh_true]], 1).contiguous())
        return (dX[:, :Fin].contiguous() if need_x else None, None, *grads)


def _csr_with_perm(m, N, dev):
    """(rowptr, col, perm): the cached CSR by target and, for injected masks, the stable COO -> CSR edge permutation."""
    cache, edge_index = m.cache, m.edge_index
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, N, str(edge_index.device))
    ent = cache.get("csr")
    if ent is None or ent[0] != key:
        rowptr, col = coo_to_csr_device(edge_index.to(dev), N)
        ent = cache["csr"] = (key, rowptr, col, edge_index)
        cache.pop("csr_t", None)
        cache.pop("perm", None)
    perm = None
    if m.injected is not None:
        pe = cache.get("perm")
        if pe is None or pe[0] != key:
            pe = cache["perm"] = (key, torch.sort(edge_index.to(dev)[1], stable=True).indices)   # the order mgu_coo_to_csr_device produces
        perm = pe[1]
    return ent[1], ent[2], perm


class _GatLayerTrainFn(torch.autograd.Function):
    """One multi-head layer in TRAIN mode with dropout (graph_attention.py:97, :160): mgu_gat_layer_forward_train with explicit masks
    (drawn by mgu_dropout_mask, or injected through `layer.dropout_masks`), backward mgu_gat_layer_backward_train with the same masks."""

    @staticmethod
    def forward(ctx_, X, m, *params):
        heads = m.heads
        dev = X.device
        N, Fin = X.shape
        H = len(heads)
        Fh_true = heads[0].out_features
        Fh = (Fh_true + 3) // 4 * 4
        W, a = stacked_head_weights(heads, m.cache)
        Xc = X.detach().contiguous()
        if Fin % 4:
            Xc = F.pad(Xc, (0, 4 - Fin % 4))
        rowptr, col, perm = _csr_with_perm(m, N, dev)
        E = col.numel()
        G, gp = 1, None
        if m.graph_ptr is not None:
            gp = m.graph_ptr.to(device=dev, dtype=torch.int32).contiguous()
            G = gp.numel() - 1
        c = _context(dev)
        Fo_true = H * Fh_true if m.concat else Fh_true
        if m.injected is not None:   # (H, E) in COO order, (N, F_out) or None: the hook a test feeds the reference's draw through
            em_coo, om_true = m.injected
            if tuple(em_coo.shape) != (H, E) or (om_true is not None and tuple(om_true.shape) != (N, Fo_true)):
                raise ValueError(f"dropout_masks must be ((heads, E) = {(H, E)}, (N, F_out) = {(N, Fo_true)} or None)")
            edge_mask = em_coo.to(device=dev, dtype=torch.float32)[:, perm].t().contiguous()
            om_true = om_true.to(device=dev, dtype=torch.float32) if om_true is not None else None
        else:
            edge_mask = _draw_mask(c, dev, (max(E, 1), H), m.dropout_rate)
            om_true = _draw_mask(c, dev, (N, Fo_true), m.dropout_rate) if m.out_dropout else None
        if om_true is None:
            out_mask = None
        elif Fh != Fh_true:   # heads run zero-padded to 16-byte lanes: the pad features are ELU(0) = 0 whatever their mask
            out_mask = F.pad(om_true.view(N, -1, Fh_true), (0, Fh - Fh_true), value=1.0).reshape(N, -1).contiguous()
        else:
            out_mask = om_true.contiguous()
        out = torch.empty((N, H * Fh if m.concat else Fh), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_gat_layer_forward_train(c.handle, Xc.data_ptr(), N, Xc.shape[1], rowptr.data_ptr(), col.data_ptr() if E else None, E,
                                                        gp.data_ptr() if gp is not None else None, G, W.data_ptr(), a.data_ptr(), H, Fh,
                                                        1 if m.concat else 0, float(m.alpha), edge_mask.data_ptr(),
                                                        out_mask.data_ptr() if out_mask is not None else None, out.data_ptr(),
                                                        _lib.current_stream_ptr(dev))
        _lib.check(rc, c.handle)
        ctx_.meta = m
        ctx_.saved = (Xc, W, a, rowptr, col, gp, G, Fh, Fh_true, Fin, edge_mask, out_mask)
        if Fh != Fh_true:
            out = out.view(N, -1, Fh)[:, :, :Fh_true].reshape(N, -1).contiguous()
        return out

    @staticmethod
    def backward(ctx_, gout):
        m = ctx_.meta
        Xc, W, a, rowptr, col, gp, G, Fh, Fh_true, Fin, edge_mask, out_mask = ctx_.saved
        heads = m.heads
        H, N, dev = len(heads), Xc.shape[0], Xc.device
        g = gout.detach().float()
        if Fh != Fh_true:
            g = F.pad(g.view(N, -1, Fh_true), (0, Fh - Fh_true)).reshape(N, -1)
        g = g.contiguous()
        c = _context(dev)
        rps, eid, tgt = _transposed_csr(m.cache, rowptr, col, N, dev, c)
        need_x = ctx_.needs_input_grad[0]
        dX = torch.empty_like(Xc) if need_x else None
        dW, da = torch.empty_like(W), torch.empty_like(a)
        E = col.numel()
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_gat_layer_backward_train(c.handle, Xc.data_ptr(), N, Xc.shape[1], rowptr.data_ptr(), col.data_ptr() if E else None, E,
                                                         rps.data_ptr(), eid.data_ptr(), tgt.data_ptr(), gp.data_ptr() if gp is not None else None, G,
                                                         W.data_ptr(), a.data_ptr(), H, Fh, 1 if m.concat else 0, float(m.alpha),
                                                         edge_mask.data_ptr(), out_mask.data_ptr() if out_mask is not None else None, g.data_ptr(),
                                                         dX.data_ptr() if need_x else None, dW.data_ptr(), da.data_ptr(), _lib.current_stream_ptr(dev))
        _lib.check(rc, c.handle)
        grads = []
        for h in range(H):
            grads.append(dW[h * Fh:h * Fh + Fh_true, :Fin].contiguous())
            grads.append(torch.cat([da[h:h + 1, :Fh_true], da[h:h + 1, Fh:Fh + Fh_true]], 1).contiguous())
        return (dX[:, :Fin].contiguous() if need_x else None, None, *grads)


class MultiHeadGATLayer(nn.Module):
    """graph_attention.py:120-160: all heads in one launch sequence; concat (:155) or mean (:158)."""

    def __init__(self, in_features, out_features, num_heads, dropout_rate, alpha, concat=True):
        super().__init__()
        self.num_heads, self.concat = num_heads, concat
        self.alpha, self.dropout_rate = alpha, dropout_rate
        if concat:
            assert out_features % num_heads == 0, "out_features must be divisible by num_heads if concatenating"
            self.head_out_features = out_features // num_heads
        else:
            self.head_out_features = out_features
        self.heads = nn.ModuleList(
            [GraphAttentionLayer(in_features, self.head_out_features, dropout_rate, alpha) for _ in range(num_heads)])
        self.dropout = nn.Dropout(dropout_rate)

    def forward(self, node_features, edge_index, graph_ptr=None):
        return _gat_layer_forward(list(self.heads), node_features, edge_index, self.concat, self.alpha,
                                  self.training, self.dropout_rate, graph_ptr, _csr_cache(self), owner=self)


class GATNetwork(nn.Module):
    """Drop-in for graph_attention.py:162-192 (same layer wiring, including the reference's
    multi-layer width mismatch: num_gat_layers >= 2 only works with num_heads == 1)."""

    def __init__(self, node_feature_dim, hidden_dim, output_dim, num_heads, num_gat_layers=1, dropout_rate=0.1,
                 alpha=0.2):
        super().__init__()
        self.num_gat_layers = num_gat_layers
        self.gat_layers = nn.ModuleList()
        if num_gat_layers == 1:
            self.gat_layers.append(MultiHeadGATLayer(node_feature_dim, output_dim, num_heads, dropout_rate, alpha, concat=False))
        else:
            self.gat_layers.append(MultiHeadGATLayer(node_feature_dim, hidden_dim, num_heads, dropout_rate, alpha, concat=True))
            for _ in range(num_gat_layers - 2):
                self.gat_layers.append(MultiHeadGATLayer(hidden_dim * num_heads, hidden_dim, num_heads, dropout_rate, alpha, concat=True))
            self.gat_layers.append(MultiHeadGATLayer(hidden_dim * num_heads, output_dim, num_heads, dropout_rate, alpha, concat=False))

    def forward(self, node_features, edge_index, graph_ptr=None):
        h = node_features
        for layer in self.gat_layers:
            h = layer(h, edge_index, graph_ptr)
        return h
