"""Step loops around the HIP path: replaces the inline loops of
experiments/segmentation_performance.py:119-144 (eval) and scripts/train_end_to_end.py:270-332
(U-Net + patch-graph GAT forward), with the batch sharded over ranks (one process per GPU)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .gat import GATNetwork, _context, _csr_cache, stacked_head_weights
from .patch_graph import PatchGraphConstructor
from .unet import UNet


class MinGraphUNet(nn.Module):
    """The 'full MinGraph-UNet forward' as one module (the reference has it only as inline code in its
    training loop, train_end_to_end.py:270-332, fed with torch.randn placeholder node features).
    Here: U-Net -> per-patch mean of the shallowest decoder feature -> block-diagonal patch graph of
    the whole batch -> GAT.  Returns (logits, skips, decoder_feats, node_embeddings (B*Np, out_dim))."""

    def __init__(self, unet: UNet, gat: GATNetwork, patch_size: int = 16):
        super().__init__()
        self.unet, self.gat = unet, gat
        self.graph = PatchGraphConstructor(patch_size)

    def forward(self, x):
        logits, skips, feats = self.unet(x)
        B, _, H, W = x.shape
        X = self.graph.patch_mean_features(feats[0])
        rowptr, col, gp, N, E = self.graph.batched_csr(H, W, B, x.device)
        emb = gat_forward_csr(self.gat, X, rowptr, col, gp)
        return logits, skips, feats, emb


def gat_forward_csr(gat: GATNetwork, X, rowptr, col, graph_ptr):
    """GATNetwork.forward on a prebuilt device CSR (skips the COO->CSR conversion of the COO API)."""
    h = X
    dev = X.device
    ctx = _context(dev)
    G = graph_ptr.numel() - 1 if graph_ptr is not None else 1
    for layer in gat.gat_layers:
        if layer.training and layer.dropout_rate > 0:
            raise RuntimeError("GAT HIP path implements eval mode: call .eval() (see mgunet.gat)")
        heads = list(layer.heads)
        Fh, H = heads[0].out_features, len(heads)
        W, a = stacked_head_weights(heads, _csr_cache(layer))
        if h.shape[1] % 4 or W.shape[1] != h.shape[1]:
            raise ValueError("node feature width must be a multiple of 4 and match W")
        h = h.contiguous()
        out = torch.empty((h.shape[0], H * Fh if layer.concat else Fh), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_gat_layer_forward(ctx.handle, h.data_ptr(), h.shape[0], h.shape[1], rowptr.data_ptr(),
                                                  col.data_ptr() if col.numel() else None, col.numel(),
                                                  graph_ptr.data_ptr() if graph_ptr is not None else None, G,
                                                  W.data_ptr(), a.data_ptr(), H, Fh, 1 if layer.concat else 0,
                                                  float(layer.alpha), out.data_ptr(), _lib.current_stream_ptr(dev))
        _lib.check(rc, ctx.handle)
        h = out
    return h


def argmax_classes(logits_nchw: torch.Tensor) -> torch.Tensor:
    """torch.argmax(seg_logits, dim=1) of segmentation_performance.py:141 on the GPU (NHWC logits)."""
    if not logits_nchw.is_cuda:
        raise RuntimeError("argmax_classes runs only on a HIP device")
    B, Cc, H, W = logits_nchw.shape
    nhwc = logits_nchw.permute(0, 2, 3, 1).contiguous()
    pred = torch.empty((B, H, W), device=logits_nchw.device, dtype=torch.int64)
    ctx = _context(logits_nchw.device)
    with torch.cuda.device(logits_nchw.device):
        rc = _lib.lib().mgu_argmax_classes(ctx.handle, nhwc.data_ptr(), B * H * W, Cc, pred.data_ptr(),
                                           _lib.current_stream_ptr(logits_nchw.device))
    _lib.check(rc, ctx.handle)
    return pred


@torch.no_grad()
def segment_batch(model: UNet, images: torch.Tensor):
    """One iteration of the eval loop (segmentation_performance.py:125-141): logits, argmax mask."""
    logits, _, _ = model(images)
    return logits, argmax_classes(logits)


def shard_batch(global_batch: int, rank: int, world_size: int):
    """Contiguous image range [lo, hi) of this rank: images (and their graphs) are independent in
    forward, so inference shards with no collective."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
