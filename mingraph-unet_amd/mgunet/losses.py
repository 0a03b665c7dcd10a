"""Host-side mirrors of the reference's auxiliary losses, routed through libmgunet.so: same names,
constructor arguments, call signatures and error messages; FORWARD values (a 0-dim float32 tensor on the input's device, no
autograd graph -- the reference's loop that would differentiate them does not run).

    TVLoss                   scripts/train_end_to_end.py:73-89
    dice_loss                scripts/train_segmentation.py:29-40
    FeatureConsistencyLoss   model/unet/feature_loss.py:5-125
    EllipticalShapeLoss      model/unet/shape_loss.py:6-180
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .gat import _context


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"mgunet {what} runs only on a HIP device (MI355X); there is deliberately no CPU fallback")


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach() if t.dtype == torch.float32 else t.detach().float()


class TVLoss(nn.Module):
    def __init__(self, weight=1.0):
        super().__init__()
        self.weight = weight

    def forward(self, x):
        _need_cuda(x, "TVLoss")
        if x.dim() != 4:
            raise ValueError("expected (B, C, H, W)")
        x = _f32(x)
        B, C_, H, W = x.shape
        out = torch.empty((), device=x.device, dtype=torch.float32)
        ctx = _context(x.device)
        sn, sc, sh, sw = x.stride()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mgu_tv_loss(ctx.handle, x.data_ptr(), B, C_, H, W, sn, sc, sh, sw, float(self.weight), out.data_ptr(),
                                              _lib.current_stream_ptr(x.device)), ctx.handle)
        return out


def dice_loss(pred, target, smooth=1.):
    """pred (B, C, H, W) logits (any strides: NHWC storage is read in place), target (B, H, W) int64."""
    _need_cuda(pred, "dice_loss")
    if pred.dim() != 4 or target.dim() != 3 or target.dtype != torch.int64:
        raise ValueError("expected logits (B, C, H, W) and an int64 target (B, H, W)")
    pred = _f32(pred)
    B, C_, H, W = pred.shape
    if pred.stride(2) != W * pred.stride(3):          # the pixel index must be one stride: both NCHW and NHWC storage qualify
        pred = pred.contiguous()
    target = target.contiguous()
    out = torch.empty((), device=pred.device, dtype=torch.float32)
    ctx = _context(pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(_lib.lib().mgu_dice_loss(ctx.handle, pred.data_ptr(), target.data_ptr(), B, H * W, C_, pred.stride(0), pred.stride(1),
                                            pred.stride(3), float(smooth), out.data_ptr(), _lib.current_stream_ptr(pred.device)), ctx.handle)
    return out


class FeatureConsistencyLoss(nn.Module):
    def __init__(self, margin=1.0):
        super().__init__()
        self.margin = margin

    def forward(self, f_unet, f_graph, correspondence_map_y, regions_unet=None, regions_graph=None):
        _need_cuda(f_unet, "FeatureConsistencyLoss")
        B, N, D = f_unet.shape                                                  # (N, D) inputs raise here, as in the reference (:88)
        if f_unet.shape != f_graph.shape:
            raise ValueError(f"f_unet ({f_unet.shape}) and f_graph ({f_graph.shape}) must have same dimensions for this loss version.")
        if correspondence_map_y.shape != (B, N):
            raise ValueError(f"correspondence_map_y (patch_region_labels_y) shape ({correspondence_map_y.shape}) "
                             f"is not (Batch, Num_Patches) = ({B}, {N}).")
        if D % 4:
            raise ValueError("the feature width must be a multiple of 4 (16-byte lanes)")
        fu, fg, y = _f32(f_unet).contiguous(), _f32(f_graph).contiguous(), correspondence_map_y.detach().float().contiguous()
        out = torch.empty((), device=fu.device, dtype=torch.float32)
        ctx = _context(fu.device)
        with torch.cuda.device(fu.device):
            _lib.check(_lib.lib().mgu_feature_consistency_loss(ctx.handle, fu.data_ptr(), fg.data_ptr(), y.data_ptr(), B, N, D,
                                                               float(self.margin), out.data_ptr(), _lib.current_stream_ptr(fu.device)),
                       ctx.handle)
        return out


class EllipticalShapeLoss(nn.Module):
    def __init__(self, epsilon=1e-6):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, segmentation_probs, object_masks_list=None):
        L = _lib.lib()
        if object_masks_list is None:
            _need_cuda(segmentation_probs, "EllipticalShapeLoss")
            p = _f32(segmentation_probs)
            B, C_, H, W = p.shape
            if p.stride(2) != W * p.stride(3):
                p = p.contiguous()
            out = torch.empty((), device=p.device, dtype=torch.float32)
            ctx = _context(p.device)
            with torch.cuda.device(p.device):
                _lib.check(L.mgu_elliptical_shape_loss_probs(ctx.handle, p.data_ptr(), B, C_, H, W, p.stride(0), p.stride(1), p.stride(3),
                                                             float(self.epsilon), out.data_ptr(), _lib.current_stream_ptr(p.device)), ctx.handle)
            return out
        masks = [m for img in object_masks_list for m in img]
        if not masks:
            dev = segmentation_probs.device if segmentation_probs is not None else torch.device("cuda")
            return torch.tensor(0.0, device=dev)
        _need_cuda(masks[0], "EllipticalShapeLoss")
        stack = torch.stack([(m != 0) for m in masks]).to(torch.uint8).contiguous()   # (M, H, W): one pass over all objects
        M, H, W = stack.shape
        out = torch.empty((), device=stack.device, dtype=torch.float32)
        ctx = _context(stack.device)
        with torch.cuda.device(stack.device):
            _lib.check(L.mgu_elliptical_shape_loss_masks(ctx.handle, stack.data_ptr(), M, H, W, float(self.epsilon), out.data_ptr(),
                                                         _lib.current_stream_ptr(stack.device)), ctx.handle)
        return out
