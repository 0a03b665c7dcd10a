"""Host-side mirrors of the reference's input / output pipeline, routed through libmgunet.so:

    ImagePreprocessor        preprocessing/image_preprocessing/image_preprocess.py:6-126
    EdgeDetector             preprocessing/graph_feature_processing/edge_detection.py:4-44
    HistogramEqualizer       preprocessing/graph_feature_processing/histogram_equalization.py:4-49
    patch_features_u8        scripts/graph_refinement.py:97-104   (image_to_patches(...).mean(...))
    postprocess_segmentation scripts/infer_segmentation.py:20-51

Same class names, constructor arguments and method names.  The reference works on host numpy arrays with cv2 / PIL; here the
pixels go to the device once (uint8) and every step is a HIP kernel that reproduces the library's integer arithmetic exactly.
Methods accept a numpy array (returned type: what the reference returns) or a uint8 CUDA tensor (returned: CUDA tensors, no host
round trip).  File paths are decoded with PIL (RGB order): cv2 is not a dependency of this package.  The random augmentations of
ImagePreprocessor (flip / rotation through torchvision's RNG) are not reproduced: apply_augmentation=True raises."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .gat import _context

_DEV = None


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("mgunet.preprocess runs only on a HIP device (MI355X); there is deliberately no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev_u8(a):
    """numpy uint8 array | uint8 tensor -> (contiguous uint8 CUDA tensor, came_from_numpy)"""
    if isinstance(a, np.ndarray):
        if a.dtype != np.uint8:
            raise TypeError(f"expected a uint8 image, got {a.dtype}")
        return torch.from_numpy(np.ascontiguousarray(a)).to(_device()), True
    if isinstance(a, torch.Tensor):
        if a.dtype != torch.uint8:
            raise TypeError(f"expected a uint8 image, got {a.dtype}")
        return (a if a.is_cuda else a.to(_device())).contiguous(), False
    raise TypeError("Input must be an image path (str) or a NumPy array.")


class ImagePreprocessor:
    def __init__(self, resize_dim=(128, 128), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), apply_augmentation=False):
        self.resize_dim = resize_dim  # H, W
        self.mean, self.std = mean, std
        self.apply_augmentation = apply_augmentation
        if apply_augmentation:
            raise NotImplementedError("the random flip / rotation of image_preprocess.py:34-51 draws from torchvision's RNG and is not "
                                      "reproduced; augment on the host or pass apply_augmentation=False")

    def preprocess(self, image_path_or_array, out: torch.Tensor = None):
        """-> (3, H, W) float32 CUDA tensor (or fills `out`, any (3, H, W) view -- e.g. one image of an NHWC batch)."""
        bgr = 1                                   # arrays are BGR, as cv2.imread delivers them (:76-78)
        if isinstance(image_path_or_array, str):
            from PIL import Image
            try:
                image = np.asarray(Image.open(image_path_or_array).convert("RGB"))
            except FileNotFoundError:
                raise FileNotFoundError(f"Image not found at {image_path_or_array}")
            bgr = 0
        else:
            image = image_path_or_array
        img, _ = _to_dev_u8(image)
        if img.dim() == 2:
            img = img.unsqueeze(-1)               # grey -> three equal channels (:79-80)
        if img.dim() != 3 or img.shape[2] not in (1, 3):
            raise ValueError("expected an (H, W, 3) or (H, W) uint8 image")
        Hs, Ws, ch = img.shape
        H, W = int(self.resize_dim[0]), int(self.resize_dim[1])
        if out is None:
            out = torch.empty((3, H, W), device=img.device, dtype=torch.float32)
        elif tuple(out.shape) != (3, H, W) or out.dtype != torch.float32 or out.device != img.device:
            raise ValueError("`out` must be a float32 (3, H, W) view on the image's device")
        mean, std = (C.c_float * 3)(*self.mean), (C.c_float * 3)(*self.std)
        ctx = _context(img.device)
        with torch.cuda.device(img.device):
            rc = _lib.lib().mgu_preprocess_image_u8(ctx.handle, img.data_ptr(), Hs, Ws, ch, bgr, H, W, mean, std, out.data_ptr(), out.stride(0),
                                                    out.stride(1), out.stride(2), _lib.current_stream_ptr(img.device))
        _lib.check(rc, ctx.handle)
        return out

    def preprocess_mask(self, mask_path_or_array, num_classes):
        if isinstance(mask_path_or_array, str):
            from PIL import Image
            try:
                mask = np.asarray(Image.open(mask_path_or_array).convert("L"))
            except FileNotFoundError:
                raise FileNotFoundError(f"Mask not found at {mask_path_or_array}")
        else:
            mask = mask_path_or_array
            if isinstance(mask, np.ndarray) and mask.ndim == 3:
                mask = mask.squeeze(2) if mask.shape[2] == 1 else np.argmax(mask, axis=2).astype(np.uint8)   # :108-114
        m, _ = _to_dev_u8(mask)
        if m.dim() != 2:
            raise ValueError("expected an (H, W) mask")
        H, W = int(self.resize_dim[0]), int(self.resize_dim[1])
        out = torch.empty((H, W), device=m.device, dtype=torch.int64)
        ctx = _context(m.device)
        with torch.cuda.device(m.device):
            rc = _lib.lib().mgu_preprocess_mask_u8(ctx.handle, m.data_ptr(), m.shape[0], m.shape[1], H, W, int(num_classes), out.data_ptr(),
                                                   _lib.current_stream_ptr(m.device))
        _lib.check(rc, ctx.handle)
        return out


def _rgb_op(fn_name, image_array_rgb, out_channels):
    if isinstance(image_array_rgb, (np.ndarray, torch.Tensor)) and (image_array_rgb.ndim != 3 or image_array_rgb.shape[2] != 3):
        raise ValueError("Input image must be an RGB image (H, W, 3).")
    img, was_np = _to_dev_u8(image_array_rgb)
    H, W, _ = img.shape
    out = torch.empty((H, W, 3) if out_channels == 3 else (H, W), device=img.device, dtype=torch.uint8)
    ctx = _context(img.device)
    with torch.cuda.device(img.device):
        rc = getattr(_lib.lib(), fn_name)(ctx.handle, img.data_ptr(), H, W, out.data_ptr(), _lib.current_stream_ptr(img.device))
    _lib.check(rc, ctx.handle)
    return out.cpu().numpy() if was_np else out


class EdgeDetector:
    def __init__(self, kernel_size=3):
        if kernel_size != 3:
            raise NotImplementedError("the HIP path implements the 3x3 Sobel operator (configs/preprocessing.yaml: sobel_kernel_size 3)")
        self.kernel_size = kernel_size

    def sobel_edges(self, image_array_rgb):
        """(H, W, 3) RGB uint8 -> (H, W) uint8 edge magnitude normalised to [0, 255]."""
        return _rgb_op("mgu_sobel_edges_u8", image_array_rgb, 1)


class HistogramEqualizer:
    def equalize_histogram_rgb(self, image_array_rgb):
        """(H, W, 3) RGB uint8 -> (H, W, 3): luminance histogram equalised in YUV."""
        return _rgb_op("mgu_equalize_hist_rgb_u8", image_array_rgb, 3)


def patch_features_u8(image, patch_size: int, per_channel: bool = False) -> torch.Tensor:
    """image_to_patches(map).mean(...) of scripts/graph_refinement.py:97-104 for a uint8 (H, W) / (H, W, C) map: per-patch means
    (zero padded bottom / right) -> (Np, 1) or (Np, C) float32 on the device."""
    img, _ = _to_dev_u8(image)
    if img.dim() == 2:
        img = img.unsqueeze(-1)
    H, W, ch = img.shape
    nph, npw = (H + patch_size - 1) // patch_size, (W + patch_size - 1) // patch_size
    out = torch.empty((nph * npw, ch if per_channel else 1), device=img.device, dtype=torch.float32)
    ctx = _context(img.device)
    with torch.cuda.device(img.device):
        rc = _lib.lib().mgu_patch_mean_u8(ctx.handle, img.data_ptr(), H, W, ch, patch_size, 1 if per_channel else 0, out.data_ptr(),
                                          _lib.current_stream_ptr(img.device))
    _lib.check(rc, ctx.handle)
    return out


DEFAULT_COLORS_BGR = [(0, 0, 0), (0, 255, 0), (0, 0, 255), (255, 0, 0)]   # infer_segmentation.py:40-45


def postprocess_segmentation(seg_logits_or_probs, num_classes, colors=None):
    """infer_segmentation.py:20-51: (C, H, W) / (1, C, H, W) scores or (H, W) labels -> (labels (H, W) numpy, colour map (H, W, 3) uint8
    numpy).  Classes beyond the four fixed colours get random colours in the reference (np.random): pass `colors` to fix them."""
    t = seg_logits_or_probs
    if not t.is_cuda:
        raise RuntimeError("mgunet.postprocess_segmentation runs only on a HIP device")
    if t.ndim == 4:
        t = t.squeeze(0)
    if t.shape[0] == num_classes and t.ndim == 3:
        from .engine import argmax_classes
        labels = argmax_classes(t.unsqueeze(0).float())[0]
    else:
        labels = t.long()
    labels = labels.contiguous()
    pal = list(colors if colors is not None else DEFAULT_COLORS_BGR)
    if len(pal) < num_classes:
        raise ValueError(f"{num_classes} classes need {num_classes} colours (the reference draws the extra ones at random)")
    palette = torch.tensor(pal[:num_classes], dtype=torch.uint8, device=t.device).contiguous()
    H, W = labels.shape
    vis = torch.empty((H, W, 3), device=t.device, dtype=torch.uint8)
    ctx = _context(t.device)
    with torch.cuda.device(t.device):
        rc = _lib.lib().mgu_colorize_labels(ctx.handle, labels.data_ptr(), H * W, palette.data_ptr(), int(num_classes), vis.data_ptr(), None,
                                            _lib.current_stream_ptr(t.device))
    _lib.check(rc, ctx.handle)
    return labels.cpu().numpy(), vis.cpu().numpy()
