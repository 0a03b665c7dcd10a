"""Host-side mirror of the reference's model/unet package, routed through libmgunet.so.

Same class names, constructor signatures, module tree and state_dict() keys as
model/unet/unet_encoder.py, unet_decoder.py and unet_model.py, so reference checkpoints load with
load_state_dict() and the reference's callers (`logits, skips, feats = model(x)`) keep working.
The torch modules here only HOLD parameters; every FLOP of forward() runs in hand-written HIP
kernels behind the C-ABI (include/mgunet.h).  Inputs must live on a HIP device: there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib


class ConvBlock(nn.Module):
    """Parameter holder for Conv3x3-BN-ReLU x2 (model/unet/unet_encoder.py:4-25)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=True):
        super().__init__()
        if kernel_size != 3 or padding != 1 or not use_batchnorm:
            raise ValueError("the HIP path implements the configuration the reference uses: k=3, p=1, BatchNorm on")
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.use_batchnorm = True
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        raise RuntimeError("ConvBlock is executed as part of mgunet.UNet.forward (fused HIP schedule)")


class UNetEncoder(nn.Module):
    """model/unet/unet_encoder.py:27-53 (module tree only)."""

    def __init__(self, in_channels=3, init_features=32, depth=4):
        super().__init__()
        self.depth = depth
        self.encoder_blocks = nn.ModuleList()
        self.pool_layers = nn.ModuleList()
        features, cin = init_features, in_channels
        for _ in range(depth):
            self.encoder_blocks.append(ConvBlock(cin, features))
            self.pool_layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            cin, features = features, features * 2
        self.bottleneck = ConvBlock(cin, features)


class DecoderBlock(nn.Module):
    """model/unet/unet_decoder.py:6-28 (module tree only)."""

    def __init__(self, in_channels_skip, in_channels_prev, out_channels, use_batchnorm=True):
        super().__init__()
        self.upsample = nn.ConvTranspose2d(in_channels_prev, in_channels_prev // 2, kernel_size=2, stride=2)
        self.conv_block = ConvBlock(in_channels_skip + in_channels_prev // 2, out_channels, use_batchnorm=use_batchnorm)


class UNetDecoder(nn.Module):
    """model/unet/unet_decoder.py:58-117 (module tree only)."""

    def __init__(self, num_classes, init_features=32, depth=4):
        super().__init__()
        self.depth = depth
        self.decoder_blocks = nn.ModuleList()
        prev = init_features * (2 ** depth)
        for i in reversed(range(depth)):
            c = init_features * (2 ** i)
            self.decoder_blocks.append(DecoderBlock(c, prev, c))
            prev = c
        self.final_conv = nn.Conv2d(prev, num_classes, kernel_size=1)


class UNet(nn.Module):
    """Drop-in for model/unet/unet_model.py:6-36.  forward(x) -> (logits, skips, decoder_feats)."""

    def __init__(self, in_channels=3, num_classes=2, init_features=32, depth=4, compute_dtype=torch.float32):
        super().__init__()
        if init_features % 4 != 0:
            raise ValueError("init_features must be a multiple of 4 (NHWC 16-byte lanes)")
        self.in_channels, self.num_classes = in_channels, num_classes
        self.init_features, self.depth = init_features, depth
        self.compute_dtype = torch.float32
        self.set_compute_dtype(compute_dtype)
        self.encoder = UNetEncoder(in_channels=in_channels, init_features=init_features, depth=depth)
        self.decoder = UNetDecoder(num_classes=num_classes, init_features=init_features, depth=depth)
        self._ctx = {}          # device index -> _lib.Context
        self._loaded_sig = {}   # device index -> signature of the parameters last packed
        self._slots = None      # [(state_dict key, owning dict, name)]: survives .to() / load_state_dict

    # ---- plumbing -------------------------------------------------------------------------
    def set_compute_dtype(self, dtype) -> "UNet":
        """torch.float32: exact-fp32 MFMA (default, the reference's precision).  torch.bfloat16: bf16 storage of
        activations and packed weights with fp32 accumulation (BASELINE config 3, inference only); parameters stay
        fp32 masters, inputs and logits stay fp32, skips / decoder features come back as bfloat16 tensors."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        if dtype == torch.bfloat16 and (getattr(self, "init_features", 8) % 8 or getattr(self, "num_classes", 1) > 4):
            raise ValueError("bfloat16 storage needs init_features % 8 == 0 and at most 4 classes")
        self.compute_dtype = dtype
        return self

    def __getstate__(self):  # contexts hold device handles: never pickled / deep-copied
        d = self.__dict__.copy()
        d["_ctx"], d["_loaded_sig"], d["_slots"] = {}, {}, None
        d.pop("_train_outputs", None)
        return d

    def _context(self, device: torch.device) -> "_lib.Context":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        code = 1 if self.compute_dtype == torch.bfloat16 else 0
        key = (idx, code)
        ctx = self._ctx.get(key)
        if ctx is None:
            ctx = _lib.Context(idx)
            ctx.key = key
            _lib.check(_lib.lib().mgu_unet_configure(ctx.handle, self.in_channels, self.num_classes,
                                                     self.init_features, self.depth, code), ctx.handle)
            self._ctx[key] = ctx
        return ctx

    def _named_tensors(self):
        if self._slots is None:
            slots = []
            for prefix, mod in self.named_modules():
                pre = prefix + "." if prefix else ""
                for n in mod._parameters:
                    slots.append((pre + n, mod._parameters, n))
                for n in mod._buffers:
                    if n not in mod._non_persistent_buffers_set:
                        slots.append((pre + n, mod._buffers, n))
            self._slots = slots
        return [(k, d[n]) for k, d, n in self._slots]

    def _sync_weights(self, ctx, device) -> None:
        tensors = self._named_tensors()
        sig = tuple((v.data_ptr(), v._version) for _, v in tensors if v.dtype.is_floating_point)   # not num_batches_tracked
        if self._loaded_sig.get(ctx.key) == sig:
            return
        descs, keep = [], []
        for k, v in tensors:
            if not v.dtype.is_floating_point:
                continue  # num_batches_tracked
            if v.device != device or v.dtype != torch.float32:
                raise RuntimeError(f"parameter {k} is {v.dtype} on {v.device}; expected float32 on {device} "
                                   "(call model.to(device) first)")
            v = v.detach().contiguous()
            keep.append(v)
            descs.append(_lib.TensorDesc(k.encode(), v.data_ptr(), v.numel()))
        arr = (_lib.TensorDesc * len(descs))(*descs)
        _lib.check(_lib.lib().mgu_unet_load_weights(ctx.handle, arr, len(descs), _lib.current_stream_ptr(device)),
                   ctx.handle)
        self._loaded_sig[ctx.key] = sig

    def mark_parameters_changed(self) -> None:
        """Parameters/buffers were modified behind torch's back (through raw pointers, without a version bump): hand the whole
        state_dict to the library again on the next forward."""
        self._loaded_sig.clear()

    def refresh_packed_weights(self, device) -> None:
        """The parameter tensors the library already knows were updated IN PLACE (the Trainer's Adam kernel writes the flat
        buffer they are views of): rebuild the packed weight forms from the recorded pointers (mgu_unet_refresh_weights: one
        launch for all Winograd sets) instead of a full mgu_unet_load_weights.  Falls back to the full path if this context
        has not been loaded yet."""
        ctx = self._context(device)
        # every OTHER context this model was loaded into (another device, the other storage dtype) still holds the old packed
        # weights under a matching (data_ptr, _version) signature -- the Adam kernel bumps no version: drop their signatures so
        # that they reload on their next forward
        for key in list(self._loaded_sig):
            if key != ctx.key:
                del self._loaded_sig[key]
        if self._loaded_sig.get(ctx.key) is None:
            return
        with torch.cuda.device(device):
            _lib.check(_lib.lib().mgu_unet_refresh_weights(ctx.handle, _lib.current_stream_ptr(device)), ctx.handle)

    def _bn_counters(self):
        return [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d)]

    def flops(self, B, H, W) -> float:
        """2*MAC of the convolutions of one forward."""
        ctx = next(iter(self._ctx.values()), None)
        if ctx is None:
            raise RuntimeError("run a forward on a HIP device first")
        return float(_lib.lib().mgu_unet_flops(ctx.handle, B, H, W))

    # ---- the hot path -----------------------------------------------------------------------
    def forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError("expected a (B,C,H,W) tensor")
        if not x.is_cuda:
            raise RuntimeError("mgunet.UNet runs only on a HIP device (MI355X); move the model and input to "
                               "'cuda' -- there is deliberately no CPU fallback")
        if x.dtype != torch.float32:
            raise TypeError(f"expected float32 input, got {x.dtype}")
        if self.training and self.compute_dtype != torch.float32:
            raise RuntimeError("bfloat16 storage is an inference mode: call .eval() or set_compute_dtype(torch.float32)")
        B, Cin, H, W = x.shape
        if Cin != self.in_channels:
            raise RuntimeError(f"expected {self.in_channels} input channels, got {Cin}")
        dev = x.device
        ctx = self._context(dev)
        self._sync_weights(ctx, dev)
        f, d = self.init_features, self.depth
        hs, ws = [H], [W]
        for _ in range(d):
            hs.append(hs[-1] // 2)
            ws.append(ws[-1] // 2)
        with torch.cuda.device(dev):
            # NHWC storage, NCHW logical view (channels_last semantics without its size-1 ambiguities)
            adt = self.compute_dtype  # activations the forward stores; logits are always fp32
            logits = torch.empty((B, H, W, self.num_classes), device=dev, dtype=torch.float32)
            cats = [torch.empty((B, hs[i], ws[i], 2 * (f << i)), device=dev, dtype=adt) for i in range(d)]
            feats = [torch.empty((B, hs[i], ws[i], f << i), device=dev, dtype=adt) for i in range(d)]
            cat_ptrs = (C.c_void_p * d)(*[t.data_ptr() for t in cats])
            feat_ptrs = (C.c_void_p * d)(*[t.data_ptr() for t in feats])
            sn, sc, sh, sw = x.stride()
            rc = _lib.lib().mgu_unet_forward(ctx.handle, x.data_ptr(), B, H, W, sn, sc, sh, sw, logits.data_ptr(),
                                            cat_ptrs, feat_ptrs, 1 if self.training else 0,
                                            _lib.current_stream_ptr(dev))
        _lib.check(rc, ctx.handle)
        if self.training:
            # the library updated running_mean/var in place through raw pointers: bump the counters the
            # reference's BatchNorm bumps (num_batches_tracked) and drop the folded eval scale/shift
            # (the library marks its own folded eval scale / shift stale: nothing has to be re-sent through the state_dict)
            torch._foreach_add_(self._bn_counters(), 1)
            self._train_outputs = (logits, cats, feats)  # kept alive for mgu_unet_backward
        skips = [cats[i].permute(0, 3, 1, 2)[:, : (f << i)] for i in range(d)]
        return logits.permute(0, 3, 1, 2), skips, [t.permute(0, 3, 1, 2) for t in feats]
