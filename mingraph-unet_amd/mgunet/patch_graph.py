"""Host-side mirror of preprocessing/graph_construction/patch_graph_construction.py.

construct_patch_graph() returns the reference's COO edge_index bit-exactly (int64, same emission
order); the index maps come from the C-ABI host routine mgu_patch_graph_build.  Additions for the
HIP path: block-diagonal batched CSR for B images, and patch_mean_features(), the deterministic node
features of the 'full forward' (what :104-136 describes and leaves NotImplemented)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib


def _build(H: int, W: int, patch: int):
    L = _lib.lib()
    E, nph, npw = C.c_int64(), C.c_int(), C.c_int()
    _lib.check(L.mgu_patch_graph_build(H, W, patch, None, None, None, C.byref(E), C.byref(nph), C.byref(npw)))
    n = nph.value * npw.value
    coo = np.empty((2, E.value), dtype=np.int64)
    rowptr = np.empty(n + 1, dtype=np.int32)
    col = np.empty(E.value, dtype=np.int32)
    _lib.check(L.mgu_patch_graph_build(H, W, patch, coo.ctypes.data_as(C.c_void_p), rowptr.ctypes.data_as(C.c_void_p),
                                       col.ctypes.data_as(C.c_void_p), None, None, None))
    return coo, rowptr, col, nph.value, npw.value


class PatchGraphConstructor:
    def __init__(self, patch_size=16):
        self.patch_size = patch_size
        self._cache = {}

    def image_to_patches(self, image_tensor_chw):
        """(C,H,W) -> ((Np,C,p,p), (nph,npw)); zero pad bottom/right (:26-47).  Layout-only op."""
        p = self.patch_size
        Cc, H, W = image_tensor_chw.shape
        if H % p != 0 or W % p != 0:
            image_tensor_chw = F.pad(image_tensor_chw, (0, (p - W % p) % p, 0, (p - H % p) % p))
            Cc, H, W = image_tensor_chw.shape
        pt = image_tensor_chw.unfold(1, p, p).unfold(2, p, p)
        nph, npw = pt.shape[1], pt.shape[2]
        return pt.permute(1, 2, 0, 3, 4).contiguous().view(-1, Cc, p, p), (nph, npw)

    def _maps(self, H, W):
        key = (H, W, self.patch_size)
        if key not in self._cache:
            self._cache[key] = _build(H, W, self.patch_size)
        return self._cache[key]

    def construct_patch_graph(self, image_tensor_chw, patch_features_flat):
        """-> (node_features, edge_index (2,E) int64 COO on the features' device) (:49-102)."""
        _, H, W = image_tensor_chw.shape
        coo, _, _, nph, npw = self._maps(H, W)
        if patch_features_flat.shape[0] != nph * npw:
            raise ValueError(f"Number of patch features ({patch_features_flat.shape[0]}) "
                             f"does not match expected number of patches ({nph * npw}) "
                             f"for image {H}x{W} and patch size {self.patch_size}.")
        ei = torch.from_numpy(coo.copy()).to(patch_features_flat.device)
        return patch_features_flat, ei

    def grid(self, H, W):
        """(nph, npw) of an H x W image (ceil division, patch_graph_construction.py:67-68)."""
        _, _, _, nph, npw = self._maps(H, W)
        return nph, npw

    def edge_index(self, H, W, device, B=1):
        """The reference's COO int64 (2, E) edge list of one H x W image (B = 1) or the block-diagonal list of B images,
        cached per (H, W, B, device) so that CSR caches keyed on the tensor keep hitting."""
        key = ("coo", H, W, self.patch_size, B, str(device))
        if key not in self._cache:
            coo, _, _, nph, npw = self._maps(H, W)
            one = torch.from_numpy(coo.copy())
            N = nph * npw
            self._cache[key] = (torch.cat([one + N * b for b in range(B)], dim=1) if B > 1 else one).to(device)
        return self._cache[key]

    def batched_csr(self, H, W, B, device):
        """Block-diagonal CSR-by-target of B copies of the H x W patch graph, on `device`:
        (rowptr int32[B*N+1], col int32[B*E], graph_ptr int32[B+1], N, E)."""
        key = ("csr", H, W, self.patch_size, B, str(device))
        if key not in self._cache:
            _, rowptr, col, nph, npw = self._maps(H, W)
            N, E = nph * npw, col.shape[0]
            rp = np.concatenate([rowptr[:-1].astype(np.int64) + b * E for b in range(B)] + [np.array([B * E])])
            cl = np.concatenate([col.astype(np.int64) + b * N for b in range(B)]) if E else np.zeros(0, np.int64)
            gp = np.arange(B + 1, dtype=np.int64) * N
            self._cache[key] = (torch.from_numpy(rp.astype(np.int32)).to(device),
                                torch.from_numpy(cl.astype(np.int32)).to(device),
                                torch.from_numpy(gp.astype(np.int32)).to(device), N, E)
        return self._cache[key]

    def patch_mean_features(self, feat_nchw: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) feature map (NHWC storage preferred) -> (B*nph*npw, C) patch means on the GPU."""
        if not feat_nchw.is_cuda:
            raise RuntimeError("patch_mean_features runs only on a HIP device")
        B, Cc, H, W = feat_nchw.shape
        if feat_nchw.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"expected float32 or bfloat16 features, got {feat_nchw.dtype}")
        code = 1 if feat_nchw.dtype == torch.bfloat16 else 0
        nhwc = feat_nchw.permute(0, 2, 3, 1).contiguous()  # no copy when storage is already NHWC
        nph, npw = (H + self.patch_size - 1) // self.patch_size, (W + self.patch_size - 1) // self.patch_size
        out = torch.empty((B * nph * npw, Cc), device=feat_nchw.device, dtype=torch.float32)
        from .gat import _context
        ctx = _context(feat_nchw.device)
        with torch.cuda.device(feat_nchw.device):
            rc = _lib.lib().mgu_patch_mean(ctx.handle, nhwc.data_ptr(), code, B, H, W, Cc, self.patch_size, out.data_ptr(),
                                           _lib.current_stream_ptr(feat_nchw.device))
        _lib.check(rc, ctx.handle)
        return out

    def get_patch_features_from_unet_encoder(self, unet_encoder_features, patches_coords_info=None):
        """The reference raises NotImplementedError here (:104-136); this build defines it as the
        per-patch mean of the given U-Net feature map."""
        return self.patch_mean_features(unet_encoder_features)
