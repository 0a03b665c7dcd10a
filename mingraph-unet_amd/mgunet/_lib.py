"""ctypes binding of libmgunet.so (include/mgunet.h).  No torch types cross this boundary: only
device pointers (tensor.data_ptr()), sizes and the HIP stream handle.  There is no CPU fallback:
if the shared library is missing or no HIP device exists, every call raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                      # .../mingraph-unet_amd
CSRC = os.path.join(PKG_ROOT, "csrc")
# MGU_LIB_PATH: developer hook for A/B runs of two builds of the SAME sources tree (tools/gpu_r03_ab.sh); the default is the in-tree build
LIB_PATH = os.environ.get("MGU_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libmgunet.so")

MGU_OK, MGU_ERR_INVALID, MGU_ERR_HIP, MGU_ERR_STATE, MGU_ERR_NOMEM = 0, -1, -2, -3, -4


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p), ("numel", C.c_int64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("flops_alg", C.c_double), ("flops_mfma", C.c_double),
                ("launches", C.c_int), ("pipe", C.c_int)]


def read_kernel_stats(ctx) -> list:
    """mgu_profile_read_kernels as a list of dicts (per kernel family since mgu_profile_enable(ctx, 1))."""
    arr = (KernelStat * 64)()
    n = C.c_int()
    check(lib().mgu_profile_read_kernels(ctx.handle, arr, 64, C.byref(n)), ctx.handle)
    return [{"name": arr[i].name.decode(), "ms": arr[i].ms, "flops_alg": arr[i].flops_alg, "flops_mfma": arr[i].flops_mfma,
             "launches": arr[i].launches, "pipe": arr[i].pipe} for i in range(n.value)]


_lib = None
_lock = threading.Lock()

# name -> (restype, argtypes); must list every symbol include/mgunet.h declares (tests check this)
_PROTOS = {
    "mgu_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mgu_destroy": (None, [C.c_void_p]),
    "mgu_last_error": (C.c_char_p, [C.c_void_p]),
    "mgu_version": (C.c_char_p, []),
    "mgu_unet_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mgu_unet_param_count": (C.c_int64, [C.c_void_p]),
    "mgu_unet_load_weights": (C.c_int, [C.c_void_p, C.POINTER(TensorDesc), C.c_int, C.c_void_p]),
    "mgu_unet_refresh_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgu_unet_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "mgu_unet_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mgu_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                                   C.c_void_p]),
    "mgu_unet_param_offset": (C.c_int64, [C.c_void_p, C.c_char_p]),
    "mgu_cross_entropy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "mgu_sync_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgu_unet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "mgu_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "mgu_comm_destroy": (C.c_int, [C.c_void_p]),
    "mgu_comm_handle": (C.c_void_p, [C.c_void_p]),
    "mgu_comm_world_size": (C.c_int, [C.c_void_p]),
    "mgu_allreduce_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgu_unet_backward_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_conv2d_wgrad_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "mgu_conv2d_dgrad_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "mgu_conv_transpose2x2_wgrad_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_conv_transpose2x2_dgrad_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_bn_relu_train_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_bn_relu_backward_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_maxpool2x2_backward_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p]),
    "mgu_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                               C.c_float, C.c_void_p]),
    "mgu_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "mgu_conv2d_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                  C.c_void_p]),
    "mgu_conv2d_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "mgu_conv2d_release": (None, [C.c_void_p, C.c_void_p]),
    "mgu_conv2d_prepared_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mgu_conv_transpose2x2_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mgu_maxpool2x2_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p]),
    "mgu_argmax_classes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_patch_graph_build": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mgu_coo_to_csr": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_coo_to_csr_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_csr_transpose_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_gat_layer_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_gat_layer_backward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_gat_layer_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "mgu_dropout_mask": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_patch_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "mgu_gat_layer_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_gat_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "mgu_gat_release": (None, [C.c_void_p, C.c_void_p]),
    "mgu_gat_layer_forward_prepared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                 C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_ncut_edge_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgu_ncut_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_ncut_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_relu_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgu_region_mean_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_region_fuse_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_tv_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                              C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_dice_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_feature_consistency_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                               C.c_void_p, C.c_void_p]),
    "mgu_tv_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "mgu_dice_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "mgu_feature_consistency_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                                        C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_loss_sync_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgu_elliptical_shape_loss_masks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_elliptical_shape_loss_probs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                                  C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "mgu_resize_bilinear_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p]),
    "mgu_region_map_gather_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                             C.c_void_p]),
    "mgu_preprocess_image_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float), C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "mgu_preprocess_mask_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_sobel_edges_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_equalize_hist_rgb_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_patch_mean_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_colorize_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgu_channel_affine_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p]),
    "mgu_channel_sum_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_channel_sum_images_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mgu_unet_request_patch_mean": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgu_unet_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mgu_unet_mfma_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mgu_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mgu_profile_read_kernels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mgu_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
}


def build(verbose: bool = False) -> str:
    """Compile csrc/ for gfx950 into lib/libmgunet.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libmgunet.so failed (see output above)")
    return LIB_PATH


def lib() -> C.CDLL:
    """Load (once) and return the shared library with prototypes installed."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(or `make -C mingraph-unet_amd/csrc`).  There is no CPU fallback.")
            L = C.CDLL(LIB_PATH)
            for name, (res, args) in _PROTOS.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            _lib = L
    return _lib


def check(rc: int, ctx=None) -> None:
    """Map a C-ABI return code to the exception type the reference's Python would raise."""
    if rc == MGU_OK:
        return
    msg = lib().mgu_last_error(ctx)
    msg = msg.decode() if msg else f"libmgunet error {rc}"
    if rc == MGU_ERR_INVALID:
        raise ValueError(msg)
    if rc == MGU_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


class Context:
    """One mgu_ctx per (object, device)."""

    def __init__(self, device_index: int):
        self.handle = C.c_void_p()
        self.device_index = device_index
        check(lib().mgu_create(device_index, C.byref(self.handle)), None)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                lib().mgu_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


def current_stream_ptr(device) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
