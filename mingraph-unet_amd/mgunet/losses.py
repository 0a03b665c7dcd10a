"""Host-side mirrors of the reference's auxiliary losses, routed through libmgunet.so: same names,
constructor arguments, call signatures and error messages.  Each call returns a 0-dim float32 tensor on the input's device.
TVLoss, dice_loss and FeatureConsistencyLoss are differentiable (torch.autograd.Function nodes whose backward is a HIP kernel:
mgu_*_backward), so `loss.backward()` as at scripts/train_segmentation.py:133 works on them; EllipticalShapeLoss has no gradient
w.r.t. its input (a function of arg-max pixel coordinates; the reference loop pins loss_shape to 0, train_end_to_end.py:287).
check_labels(device) synchronises and raises where F.one_hot would have raised on an out-of-range label.

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


def check_labels(device) -> None:
    """Synchronise and raise ValueError if a loss kernel on `device` met a label outside [0, C) since the last check (the place
    F.one_hot raises in the reference, scripts/train_segmentation.py:34); the loss calls themselves never block the host."""
    ctx = _context(torch.device(device))
    with torch.cuda.device(device):
        _lib.check(_lib.lib().mgu_loss_sync_check(ctx.handle, _lib.current_stream_ptr(torch.device(device))), ctx.handle)


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, x, weight):
        xd = _f32(x)
        B, C_, H, W = xd.shape
        out = torch.empty((), device=xd.device, dtype=torch.float32)
        ctx = _context(xd.device)
        with torch.cuda.device(xd.device):
            _lib.check(_lib.lib().mgu_tv_loss(ctx.handle, xd.data_ptr(), B, C_, H, W, *xd.stride(), float(weight), out.data_ptr(),
                                              _lib.current_stream_ptr(xd.device)), ctx.handle)
        ctx_.save_for_backward(xd)
        ctx_.weight, ctx_.in_dtype = float(weight), x.dtype
        return out

    @staticmethod
    def backward(ctx_, g):
        (xd,) = ctx_.saved_tensors
        B, C_, H, W = xd.shape
        dx = torch.empty_like(xd)
        g = g.detach().float().contiguous()
        ctx = _context(xd.device)
        with torch.cuda.device(xd.device):
            _lib.check(_lib.lib().mgu_tv_loss_backward(ctx.handle, xd.data_ptr(), B, C_, H, W, *xd.stride(), ctx_.weight, 1.0, g.data_ptr(),
                                                       dx.data_ptr(), *dx.stride(), _lib.current_stream_ptr(xd.device)), ctx.handle)
        return dx.to(ctx_.in_dtype), None


class TVLoss(nn.Module):
    def __init__(self, weight=1.0):
        super().__init__()
        self.weight = weight

    def forward(self, x):
        _need_cuda(x, "TVLoss")
        if x.dim() != 4:
            raise ValueError("expected (B, C, H, W)")
        return _TVFn.apply(x, self.weight)


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, pred, target, smooth):
        pd = _f32(pred)
        B, C_, H, W = pd.shape
        if pd.stride(2) != W * pd.stride(3):          # the pixel index must be one stride: both NCHW and NHWC storage qualify
            pd = pd.contiguous()
        target = target.contiguous()
        out = torch.empty((), device=pd.device, dtype=torch.float32)
        ctx = _context(pd.device)
        with torch.cuda.device(pd.device):
            _lib.check(_lib.lib().mgu_dice_loss(ctx.handle, pd.data_ptr(), target.data_ptr(), B, H * W, C_, pd.stride(0), pd.stride(1),
                                                pd.stride(3), float(smooth), out.data_ptr(), _lib.current_stream_ptr(pd.device)), ctx.handle)
        ctx_.save_for_backward(pd, target)
        ctx_.smooth, ctx_.in_dtype = float(smooth), pred.dtype
        return out

    @staticmethod
    def backward(ctx_, g):
        pd, target = ctx_.saved_tensors
        B, C_, H, W = pd.shape
        d = torch.empty_like(pd)                      # same strides (preserve_format): NHWC storage stays NHWC
        if d.stride() != pd.stride():
            d = torch.empty_strided(pd.shape, pd.stride(), device=pd.device, dtype=torch.float32)
        g = g.detach().float().contiguous()
        ctx = _context(pd.device)
        with torch.cuda.device(pd.device):
            _lib.check(_lib.lib().mgu_dice_loss_backward(ctx.handle, pd.data_ptr(), target.data_ptr(), B, H * W, C_, pd.stride(0), pd.stride(1),
                                                         pd.stride(3), ctx_.smooth, 1.0, g.data_ptr(), d.data_ptr(), d.stride(0), d.stride(1),
                                                         d.stride(3), 0, None, _lib.current_stream_ptr(pd.device)), ctx.handle)
        return d.to(ctx_.in_dtype), None, None


def dice_loss(pred, target, smooth=1.):
    """pred (B, C, H, W) logits (any strides: NHWC storage is read in place), target (B, H, W) int64.  Differentiable w.r.t. pred.
    A label outside [0, C) -- F.one_hot raises on it -- is reported by check_labels(pred.device) (no host sync inside the call)."""
    _need_cuda(pred, "dice_loss")
    if pred.dim() != 4 or target.dim() != 3 or target.dtype != torch.int64:
        raise ValueError("expected logits (B, C, H, W) and an int64 target (B, H, W)")
    return _DiceFn.apply(pred, target, smooth)


class _FeatConsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, f_unet, f_graph, y, margin):
        fu, fg = _f32(f_unet).contiguous(), _f32(f_graph).contiguous()
        B, N, D = fu.shape
        out = torch.empty((), device=fu.device, dtype=torch.float32)
        ctx = _context(fu.device)
        with torch.cuda.device(fu.device):
            _lib.check(_lib.lib().mgu_feature_consistency_loss(ctx.handle, fu.data_ptr(), fg.data_ptr(), y.data_ptr(), B, N, D,
                                                               float(margin), out.data_ptr(), _lib.current_stream_ptr(fu.device)), ctx.handle)
        ctx_.save_for_backward(fu, fg, y)
        ctx_.margin = float(margin)
        return out

    @staticmethod
    def backward(ctx_, g):
        fu, fg, y = ctx_.saved_tensors
        B, N, D = fu.shape
        need_u, need_g = ctx_.needs_input_grad[0], ctx_.needs_input_grad[1]
        du = torch.empty_like(fu) if need_u else None
        dg = torch.empty_like(fg) if need_g else None
        g = g.detach().float().contiguous()
        ctx = _context(fu.device)
        with torch.cuda.device(fu.device):
            _lib.check(_lib.lib().mgu_feature_consistency_loss_backward(
                ctx.handle, fu.data_ptr(), fg.data_ptr(), y.data_ptr(), B, N, D, ctx_.margin, 1.0, g.data_ptr(),
                du.data_ptr() if need_u else None, dg.data_ptr() if need_g else None, _lib.current_stream_ptr(fu.device)), ctx.handle)
        return du, dg, None, None


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
        return _FeatConsFn.apply(f_unet, f_graph, correspondence_map_y.detach().float().contiguous(), self.margin)


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
