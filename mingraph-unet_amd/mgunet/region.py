"""Region stage and feature fusion of the e2e forward (first half), routed through libmgunet.so.

The reference has no class for the region stage: it is loop-body code (scripts/train_end_to_end.py:358-421), run once
per image.  `region_stage` does the same steps for a whole batch -- label-mean pooling into K region nodes per image,
the region GAT on the fully connected K-node graphs (one block-diagonal launch), region embedding -> patches ->
nearest-upsampled pixels -> concat with the U-Net feature -- and `FeatureFusion` mirrors
model/fusion_detection/feature_fusion.py for the spatially aligned inputs that loop produces.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .gat import GATNetwork, _context


_REGION_GRAPHS = {}


def region_edge_index(K: int, device=None) -> torch.Tensor:
    """The fully connected placeholder region graph (train_end_to_end.py:375-380): (2, K(K-1)) int64."""
    if K > 1:
        src, tgt = torch.triu_indices(K, K, offset=1)
        ei = torch.stack([torch.cat([src, tgt]), torch.cat([tgt, src])], dim=0)
    else:
        ei = torch.empty((2, 0), dtype=torch.long)
    return ei.to(device) if device is not None else ei


def _f32_cuda(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: the region stage runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
    if x.dtype != torch.float32:
        raise TypeError(f"{what} must be float32, got {x.dtype}")


def region_mean_pool(patch_feats: torch.Tensor, hard_labels: torch.Tensor, B: int, K: int) -> torch.Tensor:
    """(B*Np, D) patch features + (B*Np,) labels in [0, K) -> (B*K, D) segment means, zero rows for empty segments
    (train_end_to_end.py:369-373)."""
    _f32_cuda(patch_feats, "patch_feats")
    N, D = patch_feats.shape
    if N % B:
        raise ValueError("patch_feats rows must be B * Np")
    dev = patch_feats.device
    f = patch_feats.detach().contiguous()
    h = hard_labels.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty((B * K, D), device=dev, dtype=torch.float32)
    ctx = _context(dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mgu_region_mean_pool(ctx.handle, f.data_ptr(), h.data_ptr(), B, N // B, D, K, out.data_ptr(),
                                             _lib.current_stream_ptr(dev))
    _lib.check(rc, ctx.handle)
    return out


def region_fuse(f_u, region_emb: torch.Tensor, hard_labels: torch.Tensor, B: int, H: int, W: int, nph: int, npw: int, K: int):
    """Fused feature map (B, Cu + D, H, W) (NHWC storage, NCHW view): channels [0, Cu) = f_u, the rest the embedding of the
    segment of the patch each pixel maps to under nearest interpolation (train_end_to_end.py:403-421 + feature_fusion.py:150)."""
    _f32_cuda(region_emb, "region_emb")
    dev = region_emb.device
    D = region_emb.shape[1]
    Cu, fu_ptr = 0, None
    if f_u is not None:
        _f32_cuda(f_u, "f_u")
        if tuple(f_u.shape[0:1] + f_u.shape[2:]) != (B, H, W):
            raise ValueError(f"f_u must be (B, C, {H}, {W})")
        Cu = f_u.shape[1]
        fu_nhwc = f_u.detach().permute(0, 2, 3, 1).contiguous()   # a no-op for mgunet's NHWC-stored feature maps
        fu_ptr = fu_nhwc.data_ptr()
    h = hard_labels.to(device=dev, dtype=torch.int32).contiguous()
    emb = region_emb.detach().contiguous()
    out = torch.empty((B, H, W, Cu + D), device=dev, dtype=torch.float32)
    ctx = _context(dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mgu_region_fuse_nhwc(ctx.handle, fu_ptr, Cu, emb.data_ptr(), h.data_ptr(), B, H, W, nph, npw, K, D,
                                             out.data_ptr(), _lib.current_stream_ptr(dev))
    _lib.check(rc, ctx.handle)
    return out.permute(0, 3, 1, 2)


def region_stage(patch_feats, hard_labels, B, K, region_gat: GATNetwork, nph, npw, H, W, f_u=None):
    """train_end_to_end.py:366-437 for a batch of B images with nph*npw patches each.  Returns
    (region embeddings (B*K, D'), fused (B, Cu + D', H, W))."""
    reg = region_mean_pool(patch_feats, hard_labels, B, K)
    if K > 1:
        key = (B, K, str(reg.device))
        ent = _REGION_GRAPHS.get(key)
        if ent is None:   # the block-diagonal batch of K-node graphs is a constant of (B, K): build it (and, inside the
            one = region_edge_index(K, reg.device)   # GAT mirror, its CSR) once
            ei = torch.cat([one + K * b for b in range(B)], dim=1)
            gp = torch.arange(B + 1, device=reg.device, dtype=torch.int32) * K
            ent = _REGION_GRAPHS[key] = (ei, gp)
        with torch.no_grad():   # this stage produces forward values (the pooling / fuse kernels around it carry no autograd graph)
            emb = region_gat(reg, ent[0], graph_ptr=ent[1])   # :383-384; exp(e - max e) is per graph (graph_attention.py:86)
    else:
        emb = reg                                                            # :385-387: no edges, unrefined features
    return emb, region_fuse(f_u, emb, hard_labels, B, H, W, nph, npw, K)


class FeatureFusion(nn.Module):
    """feature_fusion.py:5-162 in full: every U-Net scale is brought to the target size (bilinear, align_corners=False, :69-76)
    and written STRAIGHT into its channel slice of the fused NHWC tensor (mgu_resize_bilinear_nhwc); F_g is gathered per pixel
    from the per-region table (:84-138, invalid ids stay zero: mgu_region_map_gather_nhwc) or resized like a scale (:140-144).
    'concat' never materialises the intermediate torch.cat([...]) tensors; 'add' sums the two assembled maps."""

    def __init__(self, unet_feature_dims, gat_feature_dim, fusion_method="concat"):
        super().__init__()
        self.unet_feature_dims = unet_feature_dims
        self.gat_feature_dim = gat_feature_dim
        self.fusion_method = fusion_method.lower()

    @staticmethod
    def _place(ctx, src_nchw, out_nhwc, c_off, H, W):
        """src (B, C, h, w) -> channels [c_off, c_off + C) of out (B, H, W, ld)."""
        B, Cs, h, w = src_nchw.shape
        if Cs % 4:
            raise ValueError("feature widths must be multiples of 4 (16-byte NHWC lanes)")
        if (h, w) == (H, W):
            out_nhwc[..., c_off:c_off + Cs] = src_nchw.permute(0, 2, 3, 1)      # same size: a strided copy, no arithmetic
            return
        src = src_nchw.detach().float().permute(0, 2, 3, 1).contiguous()        # a no-op for mgunet's NHWC-stored feature maps
        dev = src.device
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_resize_bilinear_nhwc(ctx.handle, src.data_ptr(), Cs, B, h, w, Cs, out_nhwc.data_ptr(), out_nhwc.shape[3],
                                                     c_off, H, W, _lib.current_stream_ptr(dev))
        _lib.check(rc, ctx.handle)

    def forward(self, f_u_list, f_g, target_spatial_size=None, region_to_pixel_map=None):
        B = f_u_list[0].size(0)
        if target_spatial_size is None:
            target_spatial_size = (f_u_list[0].size(2), f_u_list[0].size(3))
        H, W = int(target_spatial_size[0]), int(target_spatial_size[1])
        if self.fusion_method not in ("concat", "add"):
            raise NotImplementedError(f"Fusion method '{self.fusion_method}' not implemented.")  # :157
        dev = f_u_list[0].device
        if not f_u_list[0].is_cuda:
            raise RuntimeError("mgunet.FeatureFusion runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
        per_region = f_g.ndim == 2 and region_to_pixel_map is not None
        if not per_region and f_g.ndim != 4:
            raise ValueError(f"f_g has unsupported shape {f_g.shape}. "
                             "Expected (Num_regions, D_gat) with region_map or (B, D_gat, H, W).")  # :144-146
        Cu = sum(int(t.size(1)) for t in f_u_list)
        Dg = self.gat_feature_dim if per_region else int(f_g.size(1))
        add = self.fusion_method == "add"
        if add and Cu != Dg:
            raise ValueError("Channel dimensions must match for 'add' fusion or implement adaptation.")  # :153-154
        ctx = _context(dev)
        fused = torch.empty((B, H, W, Cu if add else Cu + Dg), device=dev, dtype=torch.float32)
        gbuf = torch.empty((B, H, W, Dg), device=dev, dtype=torch.float32) if add else fused
        g_off = 0 if add else Cu
        off = 0
        for t in f_u_list:                                                      # :67-78
            self._place(ctx, t, fused, off, H, W)
            off += int(t.size(1))
        if per_region:                                                          # :84-138
            if Dg % 4 or f_g.shape[1] != Dg:
                raise ValueError("gat_feature_dim must be a multiple of 4 and match f_g's width")
            ids = region_to_pixel_map.to(device=dev, dtype=torch.int64).contiguous()
            if tuple(ids.shape) != (B, H, W):
                raise ValueError(f"region_to_pixel_map must be (B, H, W) = ({B}, {H}, {W})")
            table = f_g.detach().float().contiguous()
            with torch.cuda.device(dev):
                rc = _lib.lib().mgu_region_map_gather_nhwc(ctx.handle, table.data_ptr(), table.shape[0], Dg, ids.data_ptr(), B * H * W,
                                                           gbuf.data_ptr(), gbuf.shape[3], g_off, _lib.current_stream_ptr(dev))
            _lib.check(rc, ctx.handle)
        else:                                                                   # :140-144
            self._place(ctx, f_g, gbuf, g_off, H, W)
        if add:
            fused += gbuf
        return fused.permute(0, 3, 1, 2)
