"""Host-side mirror of model/fusion_detection/detection_head.py routed through libmgunet.so.

Same class name, constructor signature and state_dict() keys (`conv_block.{0,3}.{weight,bias}`,
`conv_block.{2,5}.{weight,bias,running_mean,running_var,num_batches_tracked}`, `fc_layers.{0,3}.*`, `fc_bbox.*`,
`fc_confidence.*`, `fc_class_scores.*`).  Eval-mode forward: dropout is the identity, BatchNorm uses its running
statistics.  The module order is Conv -> ReLU -> BatchNorm (:33-38), so each BatchNorm is an affine map AFTER the
ReLU: the first one runs as a streaming pass between the two convolutions (zero padding makes it unfoldable), the
second one commutes with the global average pool and is folded into the first Linear layer's weights.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .gat import _context


class DetectionHead(nn.Module):
    def __init__(self, in_features_channels, num_classes, num_detection_outputs=5, fc_hidden_dim=256, input_is_flat=False):
        super().__init__()
        self.input_is_flat = input_is_flat
        self.num_classes = num_classes
        if not input_is_flat:
            c = in_features_channels
            self.conv_block = nn.Sequential(nn.Conv2d(c, c // 2, kernel_size=3, padding=1), nn.ReLU(), nn.BatchNorm2d(c // 2),
                                            nn.Conv2d(c // 2, c // 4, kernel_size=3, padding=1), nn.ReLU(), nn.BatchNorm2d(c // 4),
                                            nn.AdaptiveAvgPool2d((1, 1)))  # :32-40 (parameter holders)
            fc_in = c // 4
        else:
            fc_in = in_features_channels
        self.fc_layers = nn.Sequential(nn.Linear(fc_in, fc_hidden_dim), nn.ReLU(), nn.Dropout(0.5),
                                       nn.Linear(fc_hidden_dim, fc_hidden_dim // 2), nn.ReLU(), nn.Dropout(0.5))  # :45-52
        self.fc_bbox = nn.Linear(fc_hidden_dim // 2, 4)  # :56
        self.fc_confidence = nn.Linear(fc_hidden_dim // 2, 1)  # :59
        if num_classes > 1:
            self.fc_class_scores = nn.Linear(fc_hidden_dim // 2, num_classes)  # :65-66

    # ---- packed weights: built once per (device, parameter versions), not per forward ------------------------------------
    def _prepared(self, ctx, dev):
        """Everything the forward derives from the parameters alone: the two convolutions' packed panels / Winograd
        transforms (mgu_conv2d_prepare), the BatchNorm affines, the stacked output heads.  Rebuilt only when a parameter or
        buffer has been modified (tensor._version) or the module moved to another device."""
        tensors = list(self.parameters()) + list(self.buffers())
        key = (str(dev), tuple((t.data_ptr(), t._version) for t in tensors))
        cache = self.__dict__.get("_mgu_prepared")
        if cache is not None and cache["key"] == key:
            return cache
        if cache is not None:
            for h in cache["handles"]:
                _lib.lib().mgu_conv2d_release(cache["ctx"].handle, h)
        L = _lib.lib()
        stream = _lib.current_stream_ptr(dev)
        out = {"key": key, "ctx": ctx, "handles": []}

        def prep(conv):
            import ctypes as C
            w = conv.weight.detach().contiguous()
            h = C.c_void_p()
            _lib.check(L.mgu_conv2d_prepare(ctx.handle, w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], C.byref(h), stream), ctx.handle)
            out["handles"].append(h)
            return h, conv.bias.detach().contiguous(), w.shape[0]

        if not self.input_is_flat:
            cb = self.conv_block
            out["conv1"], out["conv2"] = prep(cb[0]), prep(cb[3])
            out["bn1"], out["bn2"] = self._bn_affine(cb[2]), self._bn_affine(cb[5])
        ws, bs = [self.fc_bbox.weight, self.fc_confidence.weight], [self.fc_bbox.bias, self.fc_confidence.bias]
        if self.num_classes > 1:
            ws.append(self.fc_class_scores.weight)
            bs.append(self.fc_class_scores.bias)
        wo, bo = torch.cat(ws, 0).detach().contiguous(), torch.cat(bs, 0).detach().contiguous()
        pad = (-wo.shape[0]) % 4
        if pad:
            wo = torch.cat([wo, torch.zeros(pad, wo.shape[1], device=dev)], 0)
            bo = torch.cat([bo, torch.zeros(pad, device=dev)], 0)
        out["heads"] = (wo.contiguous(), bo.contiguous())
        self.__dict__["_mgu_prepared"] = out
        return out

    def __del__(self):
        try:
            cache = self.__dict__.get("_mgu_prepared")
            if cache:
                for h in cache["handles"]:
                    _lib.lib().mgu_conv2d_release(cache["ctx"].handle, h)
        except Exception:
            pass

    @staticmethod
    def _conv_prepared(ctx, x_nhwc, prepared, relu):
        handle, bias, Cout = prepared
        B, H, W, _ = x_nhwc.shape
        ld = (Cout + 3) // 4 * 4
        out = torch.empty((B, H, W, ld), device=x_nhwc.device, dtype=torch.float32)
        rc = _lib.lib().mgu_conv2d_prepared_nhwc(ctx.handle, handle, x_nhwc.data_ptr(), B, H, W, bias.data_ptr(), None, None,
                                                 1 if relu else 0, out.data_ptr(), ld, 0, _lib.current_stream_ptr(x_nhwc.device))
        _lib.check(rc, ctx.handle)
        return out if ld == Cout else out[..., :Cout]

    # ---- building blocks (all on the stream of the input's device) -------------------------------------------------
    @staticmethod
    def _conv(ctx, x_nhwc, w, b, k, relu):
        B, H, W, Cin = x_nhwc.shape
        Cout = w.shape[0]
        ld = (Cout + 3) // 4 * 4
        out = torch.empty((B, H, W, ld), device=x_nhwc.device, dtype=torch.float32)
        rc = _lib.lib().mgu_conv2d_nhwc(ctx.handle, x_nhwc.data_ptr(), B, H, W, Cin, w.data_ptr(), b.data_ptr(), None, None, Cout, k,
                                        1 if relu else 0, out.data_ptr(), ld, 0, _lib.current_stream_ptr(x_nhwc.device))
        _lib.check(rc, ctx.handle)
        return out if ld == Cout else out[..., :Cout]

    @staticmethod
    def _affine(ctx, x2d, scale, shift, act):
        M, C = x2d.shape
        y = torch.empty_like(x2d)
        rc = _lib.lib().mgu_channel_affine_nhwc(ctx.handle, x2d.data_ptr(), C, M, C, scale.data_ptr() if scale is not None else None,
                                                shift.data_ptr() if shift is not None else None, act, y.data_ptr(), C,
                                                _lib.current_stream_ptr(x2d.device))
        _lib.check(rc, ctx.handle)
        return y

    def _linear(self, ctx, x, w, b, relu):
        """(N, Cin) @ w^T + b as a 1x1 convolution over an (N, 1) image; widths padded to multiples of 4 by the caller."""
        N, Cin = x.shape
        return self._conv(ctx, x.reshape(1, N, 1, Cin), w.reshape(w.shape[0], Cin, 1, 1).contiguous(), b, 1, relu).reshape(N, -1)

    @staticmethod
    def _bn_affine(bn: nn.BatchNorm2d):
        a = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
        return a.contiguous(), (bn.bias - bn.running_mean * a).detach().contiguous()

    def forward(self, f_fused):
        if self.training:
            raise RuntimeError("the HIP path implements DetectionHead's eval-mode forward: call .eval()")
        if not f_fused.is_cuda:
            raise RuntimeError("mgunet.DetectionHead runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
        if f_fused.dtype != torch.float32:
            raise TypeError(f"expected float32 features, got {f_fused.dtype}")
        dev = f_fused.device
        ctx = _context(dev)
        w1, b1 = self.fc_layers[0].weight.detach(), self.fc_layers[0].bias.detach()
        with torch.cuda.device(dev):
            prep = self._prepared(ctx, dev)
            if not self.input_is_flat:
                if f_fused.dim() != 4:
                    raise ValueError("expected (B, C, H, W) fused features")
                B, C, H, W = f_fused.shape
                if C % 16:
                    raise ValueError("in_features_channels must be a multiple of 16 (C/4 is a 16-byte NHWC pixel)")
                x = f_fused.detach().permute(0, 2, 3, 1).contiguous()        # a no-op for mgunet's NHWC-stored feature maps
                x = self._conv_prepared(ctx, x, prep["conv1"], True)                                                    # :33-34
                a1, c1 = prep["bn1"]
                x = self._affine(ctx, x.reshape(-1, C // 2), a1, c1, 0).reshape(B, H, W, C // 2)                      # :35
                x = self._conv_prepared(ctx, x, prep["conv2"], True)                                                    # :36-37
                C4 = C // 4
                sums = torch.empty((B, C4), device=dev, dtype=torch.float32)
                rc = _lib.lib().mgu_channel_sum_images_nhwc(ctx.handle, x.data_ptr(), C4, B, H * W, C4, sums.data_ptr(),
                                                            _lib.current_stream_ptr(dev))                               # :39
                _lib.check(rc, ctx.handle)
                # BatchNorm (:38) after the mean: Linear1(a2 * mean + c2) = (W1 diag(a2 / HW)) sums + (W1 c2 + b1); the
                # fold depends on H*W, so it is cached per spatial size next to the packed weights
                fold = prep.setdefault("fold", {})
                if (H, W) not in fold:
                    a2, c2 = prep["bn2"]
                    fold[(H, W)] = ((w1 * (a2 / float(H * W)).unsqueeze(0)).contiguous(), (b1 + w1 @ c2).contiguous())
                w1, b1 = fold[(H, W)]
                feat = sums
            else:
                if f_fused.dim() != 2:
                    raise ValueError("expected (B, FlatFeatureDim) features")
                feat = f_fused.detach().contiguous()
            if feat.shape[1] % 4:
                raise ValueError("the flattened feature width must be a multiple of 4")
            h = self._linear(ctx, feat, w1.contiguous(), b1.contiguous(), True)                                           # :45-47
            h = self._linear(ctx, h.contiguous(), self.fc_layers[3].weight.detach(), self.fc_layers[3].bias.detach(), True)  # :49-51
            # the output heads in ONE GEMM: rows [0,4) boxes, [4] confidence, [5, 5+ncls) class scores (:56-66)
            wo, bo = prep["heads"]
            o = self._linear(ctx, h.contiguous(), wo, bo, False).contiguous()
            sg = self._affine(ctx, o, None, None, 2)                                                                      # :101, :104
        bboxes, conf = sg[:, 0:4], sg[:, 4:5]
        if self.num_classes > 1:
            return bboxes, conf, o[:, 5:5 + self.num_classes]                                                            # :107-111 (raw scores)
        return bboxes, conf
