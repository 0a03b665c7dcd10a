"""CPU oracle for the MinGraph-UNet hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch-functional + numpy, no reference imports) of the
algorithms on the north-star path of agent-charon/MinGraph-UNet.  It exists so that the HIP
path can be checked on a box where the reference's Python does not exist.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
(`mingraph-unet_amd/`) never does.

Parity status: PINNED.  `oracle/make_golden.py` (run in the build container, where
`/root/reference` is importable) loads the same formula weights into the reference's own
`model.unet.unet_model.UNet`, `model.gat.graph_attention.GATNetwork`,
`model.graph_partition.mincut_refinement.MinCutRefinement`,
`model.fusion_detection.feature_fusion.FeatureFusion`,
`model.fusion_detection.detection_head.DetectionHead`, `model.unet.feature_loss.FeatureConsistencyLoss`,
`model.unet.shape_loss.EllipticalShapeLoss` and
`preprocessing.graph_construction.patch_graph_construction.PatchGraphConstructor`, asserts
this restatement agrees (<= 1e-5 abs on O(1) logits; index maps bit-exact) and writes the
reference's outputs to `tests/golden/`.  `tests/test_oracle_golden.py` re-checks this file against
those fixtures on every run.  The reference's own tests hold no numeric expectations
(SURVEY.md section 4), so the fixtures generated from the reference are the pin.

NOT pinned by import (stated here and in DESIGN.md): `TVLoss` (scripts/train_end_to_end.py:73-89) and `dice_loss`
(scripts/train_segmentation.py:29-40) live in script modules whose import needs cv2 / torchvision (absent, no network);
`tv_loss` / `dice_loss` below restate the few lines of arithmetic and are pinned by hand-computed known answers
(tests/test_oracle_golden.py).  The input-pipeline functions (resize / normalise / mask resize / Sobel / histogram equalisation /
colour mask) restate cv2 and torchvision calls of modules that cannot be imported either; they are written against the
published definitions of those operations and carry no reference-generated fixture.

All citations are relative to /root/reference/MinGraph-UNet/.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Formula ("counter based") tensors: identical on every box, no torch RNG involved.
# --------------------------------------------------------------------------------------
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return x


def _key(name: str, seed: int) -> np.uint64:
    return np.uint64((zlib.crc32(name.encode()) << 32) ^ (seed & 0xFFFFFFFF))


def formula_uniform(name: str, shape, lo: float = 0.0, hi: float = 1.0, seed: int = 0) -> np.ndarray:
    """float32 array of `shape`, element i = lo + (hi-lo) * U(hash(name, seed, i))."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix64(idx * _GOLD + _key(name, seed))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def formula_normal(name: str, shape, seed: int = 0) -> np.ndarray:
    """float32 N(0,1) by Box-Muller on two formula uniforms."""
    u1 = formula_uniform(name + "/u1", shape, 0.0, 1.0, seed).astype(np.float64)
    u2 = formula_uniform(name + "/u2", shape, 0.0, 1.0, seed).astype(np.float64)
    u1 = np.maximum(u1, 1e-12)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def formula_labels(name: str, shape, num_classes: int, seed: int = 0) -> np.ndarray:
    u = formula_uniform(name, shape, 0.0, 1.0, seed)
    return np.minimum((u * num_classes).astype(np.int64), num_classes - 1)


# --------------------------------------------------------------------------------------
# U-Net parameters (state_dict names of model/unet/unet_model.py:18-19, dumped in SURVEY 8b)
# --------------------------------------------------------------------------------------
def _convblock_shapes(prefix, cin, cout, out):
    # model/unet/unet_encoder.py:7-13 (registration order: conv1, conv2, bn1, bn2)
    out[prefix + "conv1.weight"] = (cout, cin, 3, 3)
    out[prefix + "conv1.bias"] = (cout,)
    out[prefix + "conv2.weight"] = (cout, cout, 3, 3)
    out[prefix + "conv2.bias"] = (cout,)
    for bn in ("bn1", "bn2"):
        out[prefix + bn + ".weight"] = (cout,)
        out[prefix + bn + ".bias"] = (cout,)
        out[prefix + bn + ".running_mean"] = (cout,)
        out[prefix + bn + ".running_var"] = (cout,)
        out[prefix + bn + ".num_batches_tracked"] = ()


def unet_param_shapes(in_channels=3, num_classes=2, init_features=32, depth=4) -> "OrderedDict[str, tuple]":
    """state_dict() key -> shape, in the reference's registration order."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    feats, cin = init_features, in_channels
    for i in range(depth):  # unet_encoder.py:46-50
        _convblock_shapes(f"encoder.encoder_blocks.{i}.", cin, feats, out)
        cin, feats = feats, feats * 2
    _convblock_shapes("encoder.bottleneck.", cin, feats, out)  # unet_encoder.py:53
    prev = init_features * (2 ** depth)
    for bi, i in enumerate(reversed(range(depth))):  # unet_decoder.py:103-114
        c = init_features * (2 ** i)
        p = f"decoder.decoder_blocks.{bi}."
        out[p + "upsample.weight"] = (prev, prev // 2, 2, 2)  # ConvTranspose2d: (Cin, Cout, 2, 2)
        out[p + "upsample.bias"] = (prev // 2,)
        _convblock_shapes(p + "conv_block.", c + prev // 2, c, out)
        prev = c
    out["decoder.final_conv.weight"] = (num_classes, prev, 1, 1)  # unet_decoder.py:117
    out["decoder.final_conv.bias"] = (num_classes,)
    return out


def make_unet_params(in_channels=3, num_classes=2, init_features=32, depth=4, seed=0) -> "OrderedDict[str, torch.Tensor]":
    """Formula weights scaled so activations stay O(1) through the net (SURVEY 8d 'Weights')."""
    params: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in unet_param_shapes(in_channels, num_classes, init_features, depth).items():
        if name.endswith("num_batches_tracked"):
            params[name] = torch.zeros((), dtype=torch.int64)
            continue
        if name.endswith("conv1.weight") or name.endswith("conv2.weight") or name.endswith("final_conv.weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            a = 0.9 * float(np.sqrt(6.0 / fan_in))  # 0.9: keeps |logit| ~ 1.5 mean / 8 max at 512^2
            arr = formula_uniform(name, shape, -a, a, seed)
        elif name.endswith("upsample.weight"):
            a = float(np.sqrt(6.0 / shape[0]))  # each output pixel sees Cin inputs through one tap
            arr = formula_uniform(name, shape, -a, a, seed)
        elif ".bn" in name and name.endswith(".weight"):
            arr = formula_uniform(name, shape, 0.5, 1.5, seed)
        elif name.endswith("running_var"):
            arr = formula_uniform(name, shape, 0.5, 1.5, seed)
        elif name.endswith("running_mean"):
            arr = formula_uniform(name, shape, -0.2, 0.2, seed)
        elif ".bn" in name and name.endswith(".bias"):
            arr = formula_uniform(name, shape, -0.2, 0.2, seed)
        else:  # conv / upsample / final biases
            arr = formula_uniform(name, shape, -0.1, 0.1, seed)
        params[name] = torch.from_numpy(arr)
    return params


# --------------------------------------------------------------------------------------
# U-Net forward (model/unet/*.py)
# --------------------------------------------------------------------------------------
def _conv_block(p, prefix, x, training, momentum, eps, new_stats):
    """ConvBlock.forward, model/unet/unet_encoder.py:15-25 (conv1-bn1-relu-conv2-bn2-relu)."""
    for c, bn in (("conv1", "bn1"), ("conv2", "bn2")):
        x = F.conv2d(x, p[prefix + c + ".weight"], p[prefix + c + ".bias"], padding=1)  # :7-8
        rm, rv = p[prefix + bn + ".running_mean"], p[prefix + bn + ".running_var"]
        if training:
            rm, rv = rm.clone(), rv.clone()
        x = F.batch_norm(x, rm, rv, p[prefix + bn + ".weight"], p[prefix + bn + ".bias"],
                         training=training, momentum=momentum, eps=eps)  # :12-13 defaults
        if training and new_stats is not None:
            new_stats[prefix + bn + ".running_mean"] = rm
            new_stats[prefix + bn + ".running_var"] = rv
        x = F.relu(x)  # :9
    return x


def unet_forward(p, x, depth=4, training=False, momentum=0.1, eps=1e-5, new_stats=None):
    """UNet.forward, model/unet/unet_model.py:34-36.

    Returns (logits, skips shallow->deep, decoder feats shallow->deep) exactly as the
    reference's 3-tuple.  `new_stats` (dict) receives updated BN running stats in training.
    """
    skips = []
    cur = x
    for i in range(depth):  # unet_encoder.py:67-70
        cur = _conv_block(p, f"encoder.encoder_blocks.{i}.", cur, training, momentum, eps, new_stats)
        skips.append(cur)
        cur = F.max_pool2d(cur, kernel_size=2, stride=2)  # :48
    cur = _conv_block(p, "encoder.bottleneck.", cur, training, momentum, eps, new_stats)  # :72
    feats = []
    for bi in range(depth):  # unet_decoder.py:139-141
        skip = skips[depth - 1 - bi]  # reversed_skips, :134
        pre = f"decoder.decoder_blocks.{bi}."
        up = F.conv_transpose2d(cur, p[pre + "upsample.weight"], p[pre + "upsample.bias"], stride=2)  # :36
        dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]  # :41-42
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])  # :46-47
        cur = torch.cat([skip, up], dim=1)  # :53 skip FIRST
        cur = _conv_block(p, pre + "conv_block.", cur, training, momentum, eps, new_stats)  # :55
        feats.append(cur)
    logits = F.conv2d(cur, p["decoder.final_conv.weight"], p["decoder.final_conv.bias"])  # :143
    return logits, skips, feats[::-1]  # :149


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


def conv_block_bf16_storage(p, prefix, cur, fp32_first_weights=False, eps=1e-5):
    """ConvBlock (model/unet/unet_encoder.py:15-25, eval) with the storage roundings of the HIP bf16 mode: bf16 weights
    (fp32 for the first convolution when it runs on the VALU kernel), >= fp32 accumulation, BatchNorm as the folded fp32
    scale / shift of the epilogue, each of the two outputs rounded to bf16.  `cur` holds bf16-representable values."""
    for ci, (c, bn) in enumerate((("conv1", "bn1"), ("conv2", "bn2"))):
        w = p[prefix + c + ".weight"]
        if not (fp32_first_weights and ci == 0):
            w = _bf16(w)
        z = F.conv2d(cur, w, None, padding=1)
        scale = p[prefix + bn + ".weight"] / torch.sqrt(p[prefix + bn + ".running_var"] + eps)
        shift = p[prefix + bn + ".bias"] + (p[prefix + c + ".bias"] - p[prefix + bn + ".running_mean"]) * scale
        cur = _bf16(F.relu(z * scale[None, :, None, None] + shift[None, :, None, None]))
    return cur


def decoder_block_bf16_storage(p, bi, cur, skip):
    """DecoderBlock (model/unet/unet_decoder.py:30-56) with the same storage roundings: bf16 transposed-conv weights, its
    output rounded to bf16, skip first in the concat, then the ConvBlock."""
    pre = f"decoder.decoder_blocks.{bi}."
    up = _bf16(F.conv_transpose2d(cur, _bf16(p[pre + "upsample.weight"]), p[pre + "upsample.bias"], stride=2))
    dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
    up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return conv_block_bf16_storage(p, pre + "conv_block.", torch.cat([skip, up], dim=1))


def unet_forward_bf16_storage(p, x, depth=4, first_fp32=True):
    """UNet.forward (model/unet/unet_model.py:34-36, eval) with the STORAGE roundings of the HIP bf16 mode written out:
    the input and every stored activation are rounded to bf16, the 3x3 / transposed convolution weights are rounded to
    bf16, products accumulate in (at least) fp32, the first convolution (when it runs on the VALU kernel, `first_fp32`)
    and the 1x1 head keep fp32 weights and the logits stay fp32.  This is not the reference's arithmetic (the reference
    has no bf16 path of its own for this comparison); it is the yardstick that separates "bf16 storage costs this much"
    from "the HIP bf16 kernels compute something else".  Note that two correct implementations of THIS arithmetic still
    drift apart over 23 layers: a different fp32 summation order flips about 1 % of a layer's bf16 roundings, and every
    flipped input perturbs all the outputs it feeds, so only short segments (one or two layers from a common input) can
    be held to the one-ulp level -- tests/test_gpu_bf16.py does exactly that."""
    skips = []
    cur = _bf16(x)
    for i in range(depth):
        cur = conv_block_bf16_storage(p, f"encoder.encoder_blocks.{i}.", cur, first_fp32 and i == 0)
        skips.append(cur)
        cur = F.max_pool2d(cur, kernel_size=2, stride=2)
    cur = conv_block_bf16_storage(p, "encoder.bottleneck.", cur)
    feats = []
    for bi in range(depth):
        cur = decoder_block_bf16_storage(p, bi, cur, skips[depth - 1 - bi])
        feats.append(cur)
    logits = F.conv2d(cur, p["decoder.final_conv.weight"], p["decoder.final_conv.bias"])
    return logits, skips, feats[::-1]


def unet_flops(in_channels, num_classes, init_features, depth, H, W) -> float:
    """2*MAC count of the convolutions only, per image."""
    fl, cin, f, h, w = 0.0, in_channels, init_features, H, W
    dims = []
    for _ in range(depth):
        fl += 2.0 * h * w * 9 * (cin * f + f * f)
        dims.append((h, w, f))
        cin, f, h, w = f, f * 2, h // 2, w // 2
    fl += 2.0 * h * w * 9 * (cin * f + f * f)
    prev = f
    for i in reversed(range(depth)):
        sh, sw, c = dims[i]
        fl += 2.0 * h * w * prev * (prev // 2) * 4  # convT 2x2 s2
        h, w = sh, sw
        fl += 2.0 * h * w * 9 * ((c + prev // 2) * c + c * c)
        prev = c
    fl += 2.0 * h * w * prev * num_classes
    return fl


# --------------------------------------------------------------------------------------
# Patch graph (preprocessing/graph_construction/patch_graph_construction.py)
# --------------------------------------------------------------------------------------
def patch_grid(H: int, W: int, patch: int):
    """ceil-div grid, patch_graph_construction.py:67-68."""
    return (H + patch - 1) // patch, (W + patch - 1) // patch


def patch_graph_edges(H: int, W: int, patch: int) -> np.ndarray:
    """COO edge_index (2,E) int64 in the reference's emission order (:77-92, :97)."""
    nph, npw = patch_grid(H, W, patch)
    src, dst = [], []
    for r in range(nph):
        for c in range(npw):
            n = r * npw + c
            if c + 1 < npw:  # right neighbour, both directions (:81-84)
                src += [n, n + 1]
                dst += [n + 1, n]
            if r + 1 < nph:  # down neighbour, both directions (:86-89)
                src += [n, n + npw]
                dst += [n + npw, n]
    if not src:
        return np.empty((2, 0), dtype=np.int64)  # :95
    return np.asarray([src, dst], dtype=np.int64)


def coo_to_csr(edge_index: np.ndarray, num_nodes: int):
    """CSR by TARGET, preserving the COO order of the sources of each target (stable)."""
    tgt = edge_index[1]
    order = np.argsort(tgt, kind="stable")
    col = edge_index[0][order].astype(np.int32)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(tgt, minlength=num_nodes), out=rowptr[1:])
    return rowptr, col, order.astype(np.int64)


def image_to_patches(img_chw: torch.Tensor, patch: int):
    """patch_graph_construction.py:26-47: zero-pad bottom/right, unfold twice."""
    C, H, W = img_chw.shape
    ph, pw = (patch - H % patch) % patch, (patch - W % patch) % patch
    if ph or pw:
        img_chw = F.pad(img_chw, (0, pw, 0, ph))
        C, H, W = img_chw.shape
    pt = img_chw.unfold(1, patch, patch).unfold(2, patch, patch)
    nph, npw = pt.shape[1], pt.shape[2]
    pt = pt.permute(1, 2, 0, 3, 4).contiguous().view(-1, C, patch, patch)
    return pt, (nph, npw)


def patch_mean_features(feat_chw: torch.Tensor, patch: int) -> torch.Tensor:
    """Deterministic node features for the 'full forward' (SURVEY 8a row L3): the mean of the
    shallowest decoder feature over each patch produced by image_to_patches -> (Np, C).
    This is what patch_graph_construction.py:104-136 describes and leaves NotImplemented;
    the zero padding of :28-33 is included in the mean (divide by patch*patch)."""
    pt, _ = image_to_patches(feat_chw, patch)
    return pt.mean(dim=(2, 3))


# --------------------------------------------------------------------------------------
# GAT (model/gat/graph_attention.py)
# --------------------------------------------------------------------------------------
def gat_param_shapes(node_feature_dim, hidden_dim, output_dim, num_heads, num_gat_layers=1):
    """state_dict keys of GATNetwork (graph_attention.py:162-186)."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    layers = []
    if num_gat_layers == 1:
        layers.append((node_feature_dim, output_dim, False))  # :168-172
    else:
        layers.append((node_feature_dim, hidden_dim, True))  # :175-177
        for _ in range(num_gat_layers - 2):
            layers.append((hidden_dim * num_heads, hidden_dim, True))  # :179-182
        layers.append((hidden_dim * num_heads, output_dim, False))  # :184-186
    for l, (fin, fout, concat) in enumerate(layers):
        fh = fout // num_heads if concat else fout  # :137-141
        for h in range(num_heads):
            out[f"gat_layers.{l}.heads.{h}.W.weight"] = (fh, fin)  # :28
            out[f"gat_layers.{l}.heads.{h}.a.weight"] = (1, 2 * fh)  # :31
    return out, layers


def make_gat_params(node_feature_dim, hidden_dim, output_dim, num_heads, num_gat_layers=1, seed=0, scale=1.0):
    shapes, _ = gat_param_shapes(node_feature_dim, hidden_dim, output_dim, num_heads, num_gat_layers)
    params = OrderedDict()
    for name, shape in shapes.items():
        fan_in, fan_out = shape[1], shape[0]
        a = 1.414 * float(np.sqrt(6.0 / (fan_in + fan_out))) * scale  # xavier_uniform gain 1.414, :36-37
        params[name] = torch.from_numpy(formula_uniform(name, shape, -a, a, seed))
    return params


def gat_head_forward(X, edge_index, W, a, alpha=0.2, edge_mask=None):
    """GraphAttentionLayer.forward, graph_attention.py:40-118, literally.  edge_mask (E, 1): the train-mode dropout of the attention
    coefficients (:97) as an explicit mask, values 0 or 1 / (1 - p) in the COO edge order (None = eval mode: identity)."""
    N = X.shape[0]
    h = X @ W.t()  # :53
    hs, ht = h[edge_index[0]], h[edge_index[1]]  # :57-58
    e = F.leaky_relu(torch.cat([hs, ht], dim=1) @ a.t(), alpha)  # :61-65
    if e.numel() == 0:
        return F.elu(torch.zeros_like(h))
    exp_e = torch.exp(e - torch.max(e))  # :86 GLOBAL max
    den = torch.zeros(N, 1, dtype=X.dtype).scatter_add_(0, edge_index[1].unsqueeze(1), exp_e)  # :90-91
    att = exp_e / (den[edge_index[1]] + 1e-10)  # :94-96
    if edge_mask is not None:
        att = att * edge_mask.to(att.dtype)  # :97
    hp = torch.zeros_like(h)
    hp.scatter_add_(0, edge_index[1].unsqueeze(1).repeat(1, h.shape[1]), att * hs)  # :104-112
    return F.elu(hp)  # :118


def gat_network_forward(p, X, edge_index, num_heads, num_gat_layers=1, alpha=0.2, masks=None):
    """GATNetwork.forward, graph_attention.py:150-160, 188-192.  masks (train mode): per layer a pair (edge_masks (H, E) -- head k's
    dropout mask of its attention coefficients, :97 -- , out_mask (N, F_out) -- the layer's output dropout, :160), values 0 or
    1 / (1 - p); None = eval mode (dropout is the identity)."""
    h = X
    for l in range(num_gat_layers):
        em, om = masks[l] if masks is not None else (None, None)
        outs = [gat_head_forward(h, edge_index, p[f"gat_layers.{l}.heads.{k}.W.weight"],
                                 p[f"gat_layers.{l}.heads.{k}.a.weight"], alpha,
                                 None if em is None else em[k].reshape(-1, 1)) for k in range(num_heads)]
        concat = (num_gat_layers > 1) and (l < num_gat_layers - 1)
        h = torch.cat(outs, dim=1) if concat else torch.mean(torch.stack(outs, 0), 0)  # :153-158
        if om is not None:
            h = h * om.to(h.dtype)  # :160
    return h


def dropout_mask_from_uniform(u, p):
    """nn.Dropout's mask from uniform numbers: keep where u >= p, scaled by 1 / (1 - p)."""
    return (u >= p).to(torch.float32) / (1.0 - p)


# --------------------------------------------------------------------------------------
# MinCut stage: segment predictor + normalized-cut loss
# (model/graph_partition/mincut_refinement.py, scripts/train_end_to_end.py:40-70,154-164,347-356)
# --------------------------------------------------------------------------------------
def ncut_edge_weights(node_features, edge_index):
    """MinCutRefinement.compute_edge_weights_for_ncut, mincut_refinement.py:30-52: w = exp(-|f_s - f_t|^2 / 2)."""
    d = node_features[edge_index[0]] - node_features[edge_index[1]]  # :43-44
    return torch.exp(-torch.sum(d ** 2, dim=1) / 2.0)  # :46-51 (sigma = 1)


def normalized_cut_loss(node_features, edge_index, soft, num_segments):
    """MinCutRefinement.normalized_cut_loss, mincut_refinement.py:55-160, literally: degree = scatter_add of the edge
    weights over the SOURCE index (:96), assoc_k = sum_i P_ik deg_i (:104), cut_k = sum_e w_e P_src,k (1 - P_tgt,k)
    (:150), a segment contributes cut/assoc only if assoc > 1e-8 (:152-153)."""
    N = node_features.shape[0]
    if tuple(soft.shape) != (N, num_segments):
        raise ValueError("segment_assignments_soft shape mismatch.")  # :73-74
    w = ncut_edge_weights(node_features, edge_index)  # :77
    total = torch.zeros((), dtype=node_features.dtype)
    for k in range(num_segments):  # :83
        pk = soft[:, k]
        deg = torch.zeros(N, dtype=node_features.dtype).scatter_add_(0, edge_index[0], w)  # :93-96
        assoc = torch.sum(pk * deg)  # :104
        cut = torch.sum(w * pk[edge_index[0]] * (1 - pk[edge_index[1]]))  # :116-117,150
        if assoc > 1e-8:  # :152
            total = total + cut / assoc
    return total


def normalized_cut_loss_grad(node_features, edge_index, soft, num_segments, gloss=1.0):
    """Analytic gradient of normalized_cut_loss (the function above = mincut_refinement.py:55-160) w.r.t. the soft assignments
    and the node features, in the inputs' dtype -- what torch autograd computes for the reference when total_loss.backward()
    (train_end_to_end.py:478) reaches L_partition.  With c_k = cut_k, a_k = assoc_k, kept_k = (a_k > 1e-8) (:152):
        alpha_k = g / a_k,   beta_k = -g c_k / a_k^2                               (0 for a skipped segment)
        dL/dP_ik = sum_{e: src=i} w_e (alpha_k (1 - P_tgt,k) + beta_k) - sum_{e: tgt=i} w_e alpha_k P_src,k
        dL/dw_e  = sum_k P_src,k (alpha_k (1 - P_tgt,k) + beta_k)
        dL/df    : w_e = exp(-|f_s - f_t|^2 / 2)  =>  dw_e/df_s = -w_e (f_s - f_t) = -dw_e/df_t      (:43-51)
    Returns (dP (N, K), dF (N, D)).  tests/ check it against the reference's autograd fixture; the HIP kernel
    (csrc/ncut.hip: ncut_bwd_node_kernel) evaluates exactly these sums as two gathers per node."""
    src, tgt = edge_index[0], edge_index[1]
    N, D = node_features.shape
    dt = node_features.dtype
    diff = node_features[src] - node_features[tgt]
    w = torch.exp(-torch.sum(diff ** 2, dim=1) / 2.0)
    Ps, Pt = soft[src], soft[tgt]                                   # (E, K)
    cut = torch.sum(w[:, None] * Ps * (1 - Pt), dim=0)              # (K,)
    assoc = torch.sum(w[:, None] * Ps, dim=0)
    kept = assoc > 1e-8
    safe = torch.where(kept, assoc, torch.ones_like(assoc))
    alpha = torch.where(kept, gloss / safe, torch.zeros_like(assoc))
    beta = torch.where(kept, -gloss * cut / (safe * safe), torch.zeros_like(assoc))
    t = alpha[None, :] * (1 - Pt) + beta[None, :]                   # (E, K)
    dP = torch.zeros(N, num_segments, dtype=dt)
    dP.index_add_(0, src, w[:, None] * t)
    dP.index_add_(0, tgt, -w[:, None] * alpha[None, :] * Ps)
    q = torch.sum(Ps * t, dim=1)                                    # dL/dw_e
    gsrc = -(q * w)[:, None] * diff
    dF = torch.zeros(N, D, dtype=dt)
    dF.index_add_(0, src, gsrc)
    dF.index_add_(0, tgt, -gsrc)
    return dP, dF


# cases of tests/golden/mincut_grad.npz (oracle/make_golden.py gen_mincutgrad runs the REFERENCE classes under autograd on them)
MINCUTGRAD_CASES = {   # tag: (graph, nodes, D, K, hidden, use_gnn, heads, seed, logit shift, feature scale)
    "a": ("patch128", 64, 64, 2, 32, True, 2, 5, None, 0.15),          # the configuration of train_end_to_end.py:155-163
    "b": ("directed", 50, 24, 3, None, False, 1, 6, None, 0.2),        # MLP predictor, directed graph, a node without edges
    "c": ("directed", 50, 24, 3, None, False, 1, 6, (0.0, -60.0, 0.0), 0.2),   # a skipped segment (:152-153): no gradient through it
    "d": ("patch512", 1024, 64, 2, 32, True, 2, 7, None, 0.15),       # the headline patch graph
    "e": ("directed", 50, 24, 16, 40, False, 1, 8, None, 0.3),         # K = 16, the widest the kernels take
}


def mincutgrad_inputs(tag):
    """(edge_index (2,E) int64 tensor, X (N,D), R (N,K) weights of the soft-assignment side loss, predictor params, K, use_gnn, heads, shift)."""
    graph, N, D, K, hidden, use_gnn, heads, seed, shift, xs = MINCUTGRAD_CASES[tag]
    if graph == "patch128":
        ei = patch_graph_edges(128, 128, 16)
    elif graph == "patch512":
        ei = patch_graph_edges(512, 512, 16)
    else:   # the directed random graph of the forward fixture (make_golden.gen_mincut case b): node 49 has no outgoing edge, 48 none
        u = formula_uniform("mincut/b/e", (2, 200), 0.0, 1.0, 3)
        ei = np.stack([np.minimum((u[0] * 48).astype(np.int64), 47), np.minimum((u[1] * 50).astype(np.int64), 49)])
    X = torch.from_numpy(formula_normal(f"mincutgrad/{tag}/x", (N, D), seed=seed)) * xs
    R = torch.from_numpy(formula_normal(f"mincutgrad/{tag}/r", (N, K), seed=seed + 100))
    p = make_segment_predictor_params(D, K, hidden, use_gnn, heads, seed=seed)
    sh = None if shift is None else torch.tensor(shift, dtype=torch.float32)
    return torch.from_numpy(np.ascontiguousarray(ei)), X, R, p, K, use_gnn, heads, sh


def segment_predictor_param_shapes(in_dim, num_segments, hidden_dim=None, use_gnn=False, num_heads=1):
    """state_dict keys of PatchSegmentPredictor (train_end_to_end.py:40-60; one GAT layer as configured at :156-163)."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    if use_gnn:
        shapes, _ = gat_param_shapes(in_dim, hidden_dim if hidden_dim else in_dim, num_segments, num_heads, 1)  # :46-54
        for k, v in shapes.items():
            out["gnn_predictor." + k] = v
    else:
        hd = hidden_dim if hidden_dim is not None else in_dim * 2  # :57
        out["mlp_predictor.0.weight"] = (hd, in_dim)  # :58-62
        out["mlp_predictor.0.bias"] = (hd,)
        out["mlp_predictor.2.weight"] = (num_segments, hd)
        out["mlp_predictor.2.bias"] = (num_segments,)
    return out


def make_segment_predictor_params(in_dim, num_segments, hidden_dim=None, use_gnn=False, num_heads=1, seed=0):
    params = OrderedDict()
    for name, shape in segment_predictor_param_shapes(in_dim, num_segments, hidden_dim, use_gnn, num_heads).items():
        if len(shape) == 2:
            a = 1.414 * float(np.sqrt(6.0 / (shape[0] + shape[1])))
        else:
            a = 0.1
        params[name] = torch.from_numpy(formula_uniform(name, shape, -a, a, seed))
    return params


def segment_predictor_forward(p, X, edge_index, use_gnn, num_heads=1, alpha=0.2):
    """PatchSegmentPredictor.forward in eval mode, train_end_to_end.py:64-70."""
    if use_gnn:
        if edge_index is None:
            raise ValueError("edge_index must be provided for GNN-based segment predictor.")  # :66-67
        gp = OrderedDict((k[len("gnn_predictor."):], v) for k, v in p.items())
        return gat_network_forward(gp, X, edge_index, num_heads, 1, alpha)  # :68
    h = F.relu(X @ p["mlp_predictor.0.weight"].t() + p["mlp_predictor.0.bias"])  # :58-62
    return h @ p["mlp_predictor.2.weight"].t() + p["mlp_predictor.2.bias"]


def mincut_forward(node_features, edge_index, num_segments, segment_logits):
    """MinCutRefinement.forward after the predictor call, mincut_refinement.py:188-205: softmax over segments, the
    loss on the same features, and the hard labels train_end_to_end.py:356 takes from the soft assignments."""
    soft = F.softmax(segment_logits, dim=1)  # :190
    loss = normalized_cut_loss(node_features, edge_index, soft, num_segments)  # :193-198
    return loss, soft, torch.argmax(soft, dim=1)


# --------------------------------------------------------------------------------------
# Region stage + feature fusion (first half): scripts/train_end_to_end.py:358-437,
# model/fusion_detection/feature_fusion.py:43-162
# --------------------------------------------------------------------------------------
def region_edge_index(K: int) -> torch.Tensor:
    """The fully connected placeholder region graph, train_end_to_end.py:375-380."""
    if K > 1:
        src, tgt = torch.triu_indices(K, K, offset=1)
        return torch.stack([torch.cat([src, tgt]), torch.cat([tgt, src])], dim=0)
    return torch.empty((2, 0), dtype=torch.long)


def region_stage(patch_feats, hard_labels, K, region_gat_params, num_heads, nph, npw, H, W):
    """One image: label-mean pooling (:369-373), region GAT on the K-node graph (:382-390), map back to patches
    (:403-406), patch grid -> pixels by nearest interpolation (:410-421).  Returns (region embeddings (K, D'),
    f_g_pixel (D', H, W))."""
    D = patch_feats.shape[1]
    reg = torch.zeros(K, D, dtype=patch_feats.dtype)
    for k in range(K):
        m = hard_labels == k
        if m.sum() > 0:
            reg[k] = patch_feats[m].mean(dim=0)
    ei = region_edge_index(K)
    if K > 0 and ei.numel() > 0:
        emb = gat_network_forward(region_gat_params, reg, ei, num_heads, 1)  # :383-384
    else:
        emb = reg  # :385-387
    mapped = emb[hard_labels]  # :403-404
    grid = mapped.t().reshape(emb.shape[1], nph, npw)  # :410
    pix = F.interpolate(grid.unsqueeze(0), size=(H, W), mode="nearest").squeeze(0)  # :416-420
    return emb, pix


def feature_fusion(f_u_list, f_g, method="concat"):
    """FeatureFusion.forward for spatially aligned inputs (feature_fusion.py:64-78, 145-160): the case the e2e loop
    uses (train_end_to_end.py:433-437: one F_u scale, F_g already (B, D, H, W) at the same size)."""
    f_u = torch.cat(list(f_u_list), dim=1)  # :78
    if method == "concat":
        return torch.cat([f_u, f_g], dim=1)  # :150
    if method == "add":
        if f_u.shape[1] != f_g.shape[1]:
            raise ValueError("Channel dimensions must match for 'add' fusion or implement adaptation.")  # :153-154
        return f_u + f_g
    raise NotImplementedError(f"Fusion method '{method}' not implemented.")  # :157


def feature_fusion_full(f_u_list, f_g, gat_feature_dim, method="concat", target_spatial_size=None, region_to_pixel_map=None):
    """FeatureFusion.forward in full (feature_fusion.py:43-162): multi-scale F_u brought to the target size with
    F.interpolate(bilinear, align_corners=False) (:69-76); F_g per region gathered through region_to_pixel_map with invalid
    ids left zero (:84-138) or per pixel and bilinearly resized (:140-144); concat | add (:149-157)."""
    B = f_u_list[0].size(0)
    if target_spatial_size is None:
        target_spatial_size = (f_u_list[0].size(2), f_u_list[0].size(3))  # :64-65
    proc = []
    for fu in f_u_list:
        if (fu.size(2), fu.size(3)) != tuple(target_spatial_size):
            fu = F.interpolate(fu, size=tuple(target_spatial_size), mode="bilinear", align_corners=False)  # :71
        proc.append(fu)
    f_u = torch.cat(proc, dim=1)  # :78
    if f_g.ndim == 2 and region_to_pixel_map is not None:
        Ht, Wt = target_spatial_size
        pix = torch.zeros(B, gat_feature_dim, Ht, Wt, dtype=f_g.dtype)  # :88
        for b in range(B):
            flat = region_to_pixel_map[b].reshape(-1).long()  # :120
            valid = (flat >= 0) & (flat < f_g.shape[0])  # :123
            tmp = torch.zeros(gat_feature_dim, Ht * Wt, dtype=f_g.dtype)
            if int(valid.sum()) > 0:
                tmp[:, torch.arange(Ht * Wt)[valid]] = f_g[flat[valid]].T  # :126-134
            pix[b] = tmp.view(gat_feature_dim, Ht, Wt)  # :136
        f_g_al = pix
    elif f_g.ndim == 4:
        if (f_g.size(2), f_g.size(3)) != tuple(target_spatial_size):
            f_g_al = F.interpolate(f_g, size=tuple(target_spatial_size), mode="bilinear", align_corners=False)  # :142
        else:
            f_g_al = f_g
    else:
        raise ValueError(f"f_g has unsupported shape {f_g.shape}. Expected (Num_regions, D_gat) with region_map or (B, D_gat, H, W).")
    if method == "concat":
        return torch.cat([f_u, f_g_al], dim=1)
    if method == "add":
        if f_u.shape[1] != f_g_al.shape[1]:
            raise ValueError("Channel dimensions must match for 'add' fusion or implement adaptation.")
        return f_u + f_g_al
    raise NotImplementedError(f"Fusion method '{method}' not implemented.")


# --------------------------------------------------------------------------------------
# Auxiliary losses, forward values
# --------------------------------------------------------------------------------------
def tv_loss(x: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
    """TVLoss.forward, scripts/train_end_to_end.py:78-89: squared differences of vertical / horizontal neighbours, each sum
    divided by its count (H-1)W / H(W-1), the total divided by the batch size."""
    B, _, H, W = x.shape
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()   # :86
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()   # :87
    return weight * (h_tv / ((H - 1) * W) + w_tv / (H * (W - 1))) / B   # :84-85, :88


def dice_loss(pred: torch.Tensor, target: torch.Tensor, smooth: float = 1.0) -> torch.Tensor:
    """dice_loss, scripts/train_segmentation.py:29-40: softmax over classes, one-hot target, per (image, class) Dice with
    additive smoothing, 1 - mean."""
    p = torch.softmax(pred, dim=1)  # :30
    onehot = F.one_hot(target, num_classes=p.shape[1]).permute(0, 3, 1, 2).float()  # :34
    inter = (p * onehot).sum(dim=(2, 3))  # :36
    union = p.sum(dim=(2, 3)) + onehot.sum(dim=(2, 3))  # :37
    return 1.0 - ((2.0 * inter + smooth) / (union + smooth)).mean()  # :39-40


def feature_consistency_loss(f_unet: torch.Tensor, f_graph: torch.Tensor, y: torch.Tensor, margin: float = 1.0) -> torch.Tensor:
    """FeatureConsistencyLoss.forward, model/unet/feature_loss.py:88-125 -- the (B, N, D) / (B, N) form:
    sum_p [ y ||a-b||^2 + (1-y) relu(m - sqrt(||a-b||^2 + 1e-8))^2 ], mean over the batch."""
    if f_unet.shape != f_graph.shape:
        raise ValueError(f"f_unet ({f_unet.shape}) and f_graph ({f_graph.shape}) must have same dimensions for this loss version.")  # :96
    B, N, _ = f_unet.shape
    if tuple(y.shape) != (B, N):
        raise ValueError(f"correspondence_map_y (patch_region_labels_y) shape ({y.shape}) is not (Batch, Num_Patches) = ({B}, {N}).")  # :99-100
    yp = y.float()  # :103
    d2 = ((f_unet - f_graph) ** 2).sum(dim=2)  # :106
    dist = torch.sqrt(d2 + 1e-8)  # :115
    per = yp * d2 + (1 - yp) * F.relu(margin - dist) ** 2  # :109, :117-118
    return per.sum(dim=1).mean()  # :123


def _ellipse_object_term(mask: torch.Tensor, eps: float):
    """One object of EllipticalShapeLoss (shape_loss.py:100-144 / :155-175): mean over its pixels of (p^T S^-1 p - 1)^2 with S the
    sample covariance (torch.cov: divisor N - 1) of the centred (row, col) coordinates plus eps I; None if it is skipped."""
    if mask.sum() < 10:  # :96, :100, :157
        return None
    coords = torch.nonzero(mask, as_tuple=False).float()  # :104
    centred = coords - coords.mean(dim=0)  # :110-113
    cov = torch.cov(centred.T)  # :131
    inv = torch.inverse(cov + eps * torch.eye(2))  # :137
    mah = torch.diag(centred @ inv @ centred.T)  # :143
    return torch.mean((mah - 1.0) ** 2)  # :145


def elliptical_shape_loss(segmentation_probs: torch.Tensor = None, object_masks_list=None, eps: float = 1e-6) -> torch.Tensor:
    """EllipticalShapeLoss.forward, model/unet/shape_loss.py:17-180: with masks, every listed object; without, the arg-max == 1
    region of each image as ONE object (:61-98); objects under 10 pixels are skipped; mean over the processed objects, 0 if none."""
    terms = []
    if object_masks_list is None:
        B, C, _, _ = segmentation_probs.shape
        if C <= 1:
            return torch.tensor(0.0)  # :63-64
        labels = torch.argmax(segmentation_probs, dim=1)  # :74
        for b in range(B):
            t = _ellipse_object_term(labels[b] == 1, eps)  # :68, :94-98
            if t is not None:
                terms.append(t)
    else:
        for masks in object_masks_list:
            for m in masks:
                t = _ellipse_object_term(m, eps)
                if t is not None:
                    terms.append(t)
    return torch.stack(terms).sum() / len(terms) if terms else torch.tensor(0.0)  # :147, :177


# --------------------------------------------------------------------------------------
# Input / output pipeline.  cv2 and torchvision are absent, so these restate the library calls of
# modules that cannot be imported: PIL (present) does the resize torchvision.transforms.Resize delegates to; the cv2 steps
# follow OpenCV's documented fixed-point implementations.  No reference-generated fixture pins them.
# --------------------------------------------------------------------------------------
def preprocess_image(image_u8: np.ndarray, resize_dim, mean, std, bgr: bool = True) -> torch.Tensor:
    """ImagePreprocessor.preprocess, image_preprocess.py:57-85 with the transforms of :26-31: BGR->RGB (or grey -> 3 channels),
    ToPILImage, Resize (PIL BILINEAR with antialiasing), ToTensor (/255), Normalize."""
    from PIL import Image
    if image_u8.ndim == 2:
        rgb = np.stack([image_u8] * 3, axis=2)          # cv2.COLOR_GRAY2RGB, :79-80
    else:
        rgb = image_u8[:, :, ::-1] if bgr else image_u8  # cv2.COLOR_BGR2RGB, :77-78
    pil = Image.fromarray(np.ascontiguousarray(rgb))
    pil = pil.resize((int(resize_dim[1]), int(resize_dim[0])), Image.BILINEAR)      # transforms.Resize((H, W)) on a PIL image
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float().div(255)  # ToTensor
    m, s_ = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1), torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return (t - m) / s_                                                              # Normalize


def preprocess_mask(mask_u8: np.ndarray, resize_dim, num_classes: int) -> torch.Tensor:
    """preprocess_mask, image_preprocess.py:117-125: cv2.resize(INTER_NEAREST) -- source index min(floor(dst * (1 / (dst/src))),
    src - 1) -- np.clip, long."""
    H, W = int(resize_dim[0]), int(resize_dim[1])
    Hs, Ws = mask_u8.shape
    ify, ifx = 1.0 / (H / Hs), 1.0 / (W / Ws)
    sy = np.minimum(np.floor(np.arange(H) * ify).astype(np.int64), Hs - 1)
    sx = np.minimum(np.floor(np.arange(W) * ifx).astype(np.int64), Ws - 1)
    return torch.from_numpy(np.clip(mask_u8[sy][:, sx].astype(np.int64), 0, num_classes - 1))


def _cv_gray(rgb: np.ndarray) -> np.ndarray:
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return (r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15         # cv2.COLOR_RGB2GRAY on uint8: 15-bit fixed point (OpenCV 3.4 / 4.x RY15, GY15, BY15, gray_shift)


def sobel_edges(rgb_u8: np.ndarray) -> np.ndarray:
    """EdgeDetector.sobel_edges, edge_detection.py:28-44 (ksize 3): grey, Sobel x / y (BORDER_REFLECT_101) in CV_64F, magnitude,
    / max * 255, astype(uint8)."""
    g = np.pad(_cv_gray(rgb_u8), 1, mode="reflect").astype(np.float64)
    gx = (g[:-2, 2:] + 2 * g[1:-1, 2:] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[1:-1, :-2] + g[2:, :-2])
    gy = (g[2:, :-2] + 2 * g[2:, 1:-1] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[:-2, 1:-1] + g[:-2, 2:])
    mag = np.sqrt(gx ** 2 + gy ** 2)
    if np.max(mag) > 0:
        return (mag / np.max(mag) * 255).astype(np.uint8)
    return np.zeros_like(mag, dtype=np.uint8)


def equalize_histogram_rgb(rgb_u8: np.ndarray) -> np.ndarray:
    """HistogramEqualizer.equalize_histogram_rgb, histogram_equalization.py:27-35: cv2 RGB2YUV (14-bit fixed point), equalizeHist on
    Y, YUV2RGB."""
    def descale(v):
        return (v + (1 << 13)) >> 14
    r, g, b = (rgb_u8[..., i].astype(np.int64) for i in range(3))
    Y = descale(r * 4899 + g * 9617 + b * 1868)
    U = np.clip(descale((b - Y) * 8061 + (128 << 14)), 0, 255)
    V = np.clip(descale((r - Y) * 14369 + (128 << 14)), 0, 255)
    hist = np.bincount(Y.reshape(-1), minlength=256)
    i0 = int(np.nonzero(hist)[0][0])
    total = Y.size
    lut = np.arange(256, dtype=np.int64)
    if hist[i0] == total:
        lut[:] = i0
    else:
        scale = np.float32(255.0) / np.float32(total - hist[i0])
        csum = np.cumsum(np.where(np.arange(256) > i0, hist, 0))
        lut = np.where(np.arange(256) > i0, np.clip(np.rint(csum.astype(np.float32) * scale), 0, 255), 0).astype(np.int64)
    Y2 = lut[Y]
    u, v = U - 128, V - 128
    out = np.stack([Y2 + descale(v * 18678), Y2 + descale(u * -6472 + v * -9519), Y2 + descale(u * 33292)], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def patch_mean_u8(img_u8: np.ndarray, patch: int, per_channel: bool = False) -> torch.Tensor:
    """image_to_patches(...).mean(dim=[1,2,3]) / per channel (scripts/graph_refinement.py:97-104; zero padding of :28-33)."""
    a = img_u8[..., None] if img_u8.ndim == 2 else img_u8
    t = torch.from_numpy(a.astype(np.float32)).permute(2, 0, 1)
    patches, _ = image_to_patches(t, patch)
    return patches.mean(dim=[2, 3]) if per_channel else patches.mean(dim=[1, 2, 3]).unsqueeze(-1)


def colorize_labels(labels: np.ndarray, num_classes: int, colors) -> np.ndarray:
    """postprocess_segmentation's colour map, infer_segmentation.py:36-49."""
    vis = np.zeros(labels.shape + (3,), dtype=np.uint8)
    for k in range(num_classes):
        vis[labels == k] = colors[k]
    return vis


# --------------------------------------------------------------------------------------
# DetectionHead (second half): model/fusion_detection/detection_head.py:4-114
# --------------------------------------------------------------------------------------
def detection_head_param_shapes(in_ch, num_classes, fc_hidden_dim=256, input_is_flat=False):
    """state_dict keys of DetectionHead (detection_head.py:31-66)."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    if not input_is_flat:
        c1, c2 = in_ch // 2, in_ch // 4
        out["conv_block.0.weight"], out["conv_block.0.bias"] = (c1, in_ch, 3, 3), (c1,)  # :33
        for k, c in (("2", c1), ("5", c2)):  # :35, :38
            out[f"conv_block.{k}.weight"], out[f"conv_block.{k}.bias"] = (c,), (c,)
            out[f"conv_block.{k}.running_mean"], out[f"conv_block.{k}.running_var"] = (c,), (c,)
        out["conv_block.3.weight"], out["conv_block.3.bias"] = (c2, c1, 3, 3), (c2,)  # :36
        fc_in = c2
    else:
        fc_in = in_ch
    out["fc_layers.0.weight"], out["fc_layers.0.bias"] = (fc_hidden_dim, fc_in), (fc_hidden_dim,)  # :46
    out["fc_layers.3.weight"], out["fc_layers.3.bias"] = (fc_hidden_dim // 2, fc_hidden_dim), (fc_hidden_dim // 2,)  # :49
    out["fc_bbox.weight"], out["fc_bbox.bias"] = (4, fc_hidden_dim // 2), (4,)  # :56
    out["fc_confidence.weight"], out["fc_confidence.bias"] = (1, fc_hidden_dim // 2), (1,)  # :59
    if num_classes > 1:
        out["fc_class_scores.weight"], out["fc_class_scores.bias"] = (num_classes, fc_hidden_dim // 2), (num_classes,)  # :65-66
    return out


def make_detection_head_params(in_ch, num_classes, fc_hidden_dim=256, input_is_flat=False, seed=0):
    params = OrderedDict()
    for name, shape in detection_head_param_shapes(in_ch, num_classes, fc_hidden_dim, input_is_flat).items():
        if name.endswith("running_var"):
            v = formula_uniform(name, shape, 0.5, 1.5, seed)
        elif name.endswith("running_mean"):
            v = formula_uniform(name, shape, -0.2, 0.2, seed)
        elif len(shape) == 1 and ("conv_block.2" in name or "conv_block.5" in name) and name.endswith("weight"):
            v = formula_uniform(name, shape, -1.5, 1.5, seed)   # BatchNorm gamma of both signs: the affine follows the ReLU
        elif len(shape) == 1:
            v = formula_uniform(name, shape, -0.2, 0.2, seed)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = float(np.sqrt(6.0 / fan_in))
            v = formula_uniform(name, shape, -a, a, seed)
        params[name] = torch.from_numpy(v)
    return params


def detection_head_forward(p, f_fused, num_classes, input_is_flat=False, eps=1e-5):
    """DetectionHead.forward in eval mode (dropout = identity, BatchNorm on running statistics), detection_head.py:69-114.
    Note the order Conv -> ReLU -> BatchNorm (:33-38): the affine comes AFTER the ReLU."""
    if not input_is_flat:
        x = F.relu(F.conv2d(f_fused, p["conv_block.0.weight"], p["conv_block.0.bias"], padding=1))
        x = F.batch_norm(x, p["conv_block.2.running_mean"], p["conv_block.2.running_var"], p["conv_block.2.weight"],
                         p["conv_block.2.bias"], False, 0.1, eps)
        x = F.relu(F.conv2d(x, p["conv_block.3.weight"], p["conv_block.3.bias"], padding=1))
        x = F.batch_norm(x, p["conv_block.5.running_mean"], p["conv_block.5.running_var"], p["conv_block.5.weight"],
                         p["conv_block.5.bias"], False, 0.1, eps)
        x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)  # :39, :93
    else:
        x = f_fused
    x = F.relu(x @ p["fc_layers.0.weight"].t() + p["fc_layers.0.bias"])  # :45-52
    x = F.relu(x @ p["fc_layers.3.weight"].t() + p["fc_layers.3.bias"])
    bboxes = torch.sigmoid(x @ p["fc_bbox.weight"].t() + p["fc_bbox.bias"])  # :101
    conf = torch.sigmoid(x @ p["fc_confidence.weight"].t() + p["fc_confidence.bias"])  # :104
    if num_classes > 1:
        return bboxes, conf, x @ p["fc_class_scores.weight"].t() + p["fc_class_scores.bias"]  # :107-111
    return bboxes, conf


# --------------------------------------------------------------------------------------
# Step loops (scripts/train_segmentation.py:117-137, experiments/segmentation_performance.py:119-144)
# --------------------------------------------------------------------------------------
def eval_step(p, x, depth=4):
    """no_grad -> model(x) -> argmax(1): segmentation_performance.py:125-141."""
    with torch.no_grad():
        logits, _, _ = unet_forward(p, x, depth, training=False)
    return logits, torch.argmax(logits, dim=1)


def train_step(p, x, y, depth=4, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, step=1,
               exp_avg=None, exp_avg_sq=None, loss_kind="ce"):
    """zero_grad -> fwd -> CrossEntropyLoss (mean) [+ dice_loss] -> backward -> Adam(lr, wd as L2-in-grad).

    train_segmentation.py:91,96,121-134 with torch.optim.Adam semantics; loss_kind "ce+dice" is the sum the script forms at
    :126-130 (its dice term cannot run there: `F` is never imported).  Returns
    (loss, grads, new_params, new_bn_stats, exp_avg, exp_avg_sq)."""
    names = [k for k, v in p.items() if v.dtype.is_floating_point and "running_" not in k]
    q = OrderedDict((k, (v.clone().requires_grad_(True) if k in names else v.clone())) for k, v in p.items())
    stats = {}
    logits, _, _ = unet_forward(q, x, depth, training=True, new_stats=stats)
    loss = F.cross_entropy(logits, y)
    if loss_kind == "ce+dice":
        loss = loss + dice_loss(logits, y)  # :128-130
    elif loss_kind != "ce":
        raise ValueError(loss_kind)
    grads = torch.autograd.grad(loss, [q[k] for k in names])
    grads = OrderedDict(zip(names, grads))
    new_p = OrderedDict((k, v.detach().clone()) for k, v in q.items())
    exp_avg = exp_avg or {k: torch.zeros_like(p[k]) for k in names}
    exp_avg_sq = exp_avg_sq or {k: torch.zeros_like(p[k]) for k in names}
    b1, b2 = betas
    for k in names:
        g = grads[k] + weight_decay * p[k]
        exp_avg[k] = b1 * exp_avg[k] + (1 - b1) * g
        exp_avg_sq[k] = b2 * exp_avg_sq[k] + (1 - b2) * g * g
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = exp_avg_sq[k].sqrt() / np.sqrt(bc2) + eps
        new_p[k] = p[k] - (lr / bc1) * exp_avg[k] / denom
    for k, v in stats.items():
        new_p[k] = v
    return loss.detach(), grads, new_p, stats, exp_avg, exp_avg_sq
