"""Step loops around the HIP path: replaces the inline loops of
experiments/segmentation_performance.py:119-144 (eval) and scripts/train_end_to_end.py:270-332
(U-Net + patch-graph GAT forward), with the batch sharded over ranks (one process per GPU)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .gat import GATNetwork, _context, _csr_cache, prepared_head_weights, stacked_head_weights
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
        B, _, H, W = x.shape
        X = None
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4:
            # ask the U-Net forward to emit the node features (patch means of decoder_feats[0]) in the same pass over
            # the feature map as the final 1x1 conv (mgu_unet_request_patch_mean)
            p = self.graph.patch_size
            X = torch.empty((B * ((H + p - 1) // p) * ((W + p - 1) // p), self.unet.init_features), device=x.device,
                            dtype=torch.float32)
            ctx = self.unet._context(x.device)
            _lib.check(_lib.lib().mgu_unet_request_patch_mean(ctx.handle, p, X.data_ptr()), ctx.handle)
        try:
            logits, skips, feats = self.unet(x)
        except Exception:
            if X is not None:   # the forward was rejected before it could serve the request: cancel it
                _lib.lib().mgu_unet_request_patch_mean(ctx.handle, 0, None)
            raise
        if X is None:
            X = self.graph.patch_mean_features(feats[0])
        rowptr, col, gp, N, E = self.graph.batched_csr(H, W, B, x.device)
        emb = gat_forward_csr(self.gat, X, rowptr, col, gp)
        return logits, skips, feats, emb


class MinGraphUNetE2E(nn.Module):
    """Stages 1-7 of the reference's end-to-end forward (scripts/train_end_to_end.py:270-453) for a whole batch, on
    deterministic node features: U-Net -> patch-mean node features -> patch GAT -> segment predictor +
    normalized-cut loss (mean over images, :425) -> region stage -> FeatureFusion with the shallowest decoder feature ->
    DetectionHead.  Returns a dict with the tensors the loop produces: logits, node embeddings, loss_partition, soft /
    hard patch assignments, region embeddings, fused features, boxes, confidence (and class scores)."""

    def __init__(self, unet: UNet, patch_gat: GATNetwork, segment_predictor, mincut, region_gat: GATNetwork, detection_head,
                 num_segments: int, patch_size: int = 16):
        super().__init__()
        self.core = MinGraphUNet(unet, patch_gat, patch_size)
        self.segment_predictor, self.mincut, self.region_gat, self.detection_head = segment_predictor, mincut, region_gat, detection_head
        self.num_segments = num_segments

    @torch.no_grad()   # a pipeline of forward values (the pooling / cut / fuse kernels between the modules carry no autograd graph)
    def forward(self, x):
        from .region import region_stage
        B, _, H, W = x.shape
        logits, skips, feats, emb = self.core(x)
        graph = self.core.graph
        nph, npw = graph.grid(H, W)
        K = self.num_segments
        ei_one = graph.edge_index(H, W, x.device)                                   # one image's COO (all images share it)
        if getattr(self.segment_predictor, "use_gnn", False):
            ei_all = graph.edge_index(H, W, x.device, B)
            gp = graph.batched_csr(H, W, B, x.device)[2]                            # per-graph max e (graph_attention.py:86)
            seg_logits = self.segment_predictor.gnn_predictor(emb, ei_all, graph_ptr=gp)
        else:
            seg_logits = self.segment_predictor(emb)
        losses, soft, hard = self.mincut.forward_batched(emb, ei_one, B, K, seg_logits.contiguous())
        region_emb, fused = region_stage(emb, hard, B, K, self.region_gat, nph, npw, H, W, f_u=feats[0])
        det = self.detection_head(fused)
        out = {"logits": logits, "skips": skips, "decoder_feats": feats, "node_embeddings": emb, "loss_partition": losses.mean(),
               "soft_assignments": soft, "hard_labels": hard, "region_embeddings": region_emb, "fused": fused,
               "bboxes": det[0], "confidence": det[1]}
        if len(det) > 2:
            out["class_scores"] = det[2]
        return out


def gat_forward_csr(gat: GATNetwork, X, rowptr, col, graph_ptr):
    """GATNetwork.forward on a prebuilt device CSR (skips the COO->CSR conversion of the COO API)."""
    h = X
    dev = X.device
    ctx = _context(dev)
    G = graph_ptr.numel() - 1 if graph_ptr is not None else 1
    for layer in gat.gat_layers:
        if layer.training and layer.dropout_rate > 0:
            raise RuntimeError("gat_forward_csr is the inference schedule on a prebuilt CSR: call .eval(), or run the layer through "
                               "GATNetwork.forward(X, edge_index) for train-mode dropout (see mgunet.gat)")
        heads = list(layer.heads)
        Fh, H = heads[0].out_features, len(heads)
        W, a = stacked_head_weights(heads, _csr_cache(layer))
        if h.shape[1] % 4 or W.shape[1] != h.shape[1]:
            raise ValueError("node feature width must be a multiple of 4 and match W")
        h = h.contiguous()
        out = torch.empty((h.shape[0], H * Fh if layer.concat else Fh), device=dev, dtype=torch.float32)
        handle = prepared_head_weights(heads, _csr_cache(layer), ctx, dev, col.numel() > 0)
        with torch.cuda.device(dev):
            rc = _lib.lib().mgu_gat_layer_forward_prepared(ctx.handle, handle, h.data_ptr(), h.shape[0], rowptr.data_ptr(),
                                                           col.data_ptr() if col.numel() else None, col.numel(),
                                                           graph_ptr.data_ptr() if graph_ptr is not None else None, G,
                                                           1 if layer.concat else 0, float(layer.alpha), out.data_ptr(),
                                                           _lib.current_stream_ptr(dev))
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


def allreduce_mean_(flat: torch.Tensor, group=None) -> float:
    """Sum-all-reduce `flat` in place over the ranks (RCCL over xGMI when the backend is "nccl") and return
    the factor (1/world) that turns the sum into the mean; 1.0 and no collective for a single process.
    One flat pre-packed buffer = one collective per step (31 MB for UNet(3,2,32,4))."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1:
        return 1.0
    if flat.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal only: gloo has no device path here
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def adam_state_dict(params, exp_avg_flat, exp_avg_sq_flat, step, lr, betas, eps, weight_decay) -> dict:
    """torch.optim.Adam.state_dict() layout from flat moment buffers in named_parameters() order (pure tensor plumbing: runs on any
    device; the CPU tier checks it against a real torch.optim.Adam)."""
    state, off = {}, 0
    for i, p in enumerate(params):
        k = p.numel()
        if step > 0:
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": exp_avg_flat[off:off + k].detach().clone().view_as(p),
                        "exp_avg_sq": exp_avg_sq_flat[off:off + k].detach().clone().view_as(p)}
        off += k
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False, "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


class StepLR:
    """optim.lr_scheduler.StepLR(optimizer, step_size, gamma) for a Trainer (train_segmentation.py:105, stepped once per epoch at
    :143): lr = base_lr * gamma ** (epoch // step_size)."""

    def __init__(self, trainer, step_size: int, gamma: float = 0.1):
        self.trainer, self.step_size, self.gamma = trainer, int(step_size), float(gamma)
        self.base_lr, self.last_epoch = trainer.lr, 0

    def step(self) -> None:
        self.last_epoch += 1
        self.trainer.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def get_last_lr(self):
        return [self.trainer.lr]

    def state_dict(self) -> dict:
        return {"step_size": self.step_size, "gamma": self.gamma, "base_lrs": [self.base_lr], "last_epoch": self.last_epoch}

    def load_state_dict(self, sd: dict) -> None:
        self.step_size, self.gamma, self.base_lr, self.last_epoch = sd["step_size"], sd["gamma"], sd["base_lrs"][0], sd["last_epoch"]
        self.trainer.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))


class Trainer:
    """The train step of scripts/train_segmentation.py:117-137 on the HIP path:
    zero_grad -> logits = model(images) (train-mode BatchNorm) -> CrossEntropyLoss (mean) -> backward ->
    [mean all-reduce of the flat gradient over the data-parallel ranks] -> Adam(lr, weight_decay as L2).
    Parameters are re-homed as views of ONE flat fp32 buffer (named_parameters() order) so the optimizer is
    a single kernel and the gradient exchange a single collective.  Each rank normalises BatchNorm over its
    own shard (DDP semantics)."""

    def __init__(self, model: UNet, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, process_group=None, comm="auto",
                 loss="ce", dice_smooth=1.0, optimizer="adam", momentum=0.9):
        """comm: "auto" (default) -- with more than one rank the gradient is mean-all-reduced by torch.distributed (RCCL when the
        backend is "nccl") after backward: the path exercised with two ranks.  "rccl" -- opt in to libmgunet's own RCCL
        communicator with the bucketed exchange overlapped with backward (mgu_unet_backward_allreduce); tests/test_gpu_rccl.py
        runs it with two ranks on a box that has two GPUs and with one rank elsewhere.  None -- no exchange.
        loss: "ce" -- nn.CrossEntropyLoss() alone (train_segmentation.py:91,127); "ce+dice" -- the sum the script forms at
        :126-130, `criterion_ce(logits, masks) + dice_loss(logits, masks)`: the Dice gradient (through the softmax) is added to
        the cross-entropy gradient in the same dlogits buffer (mgu_dice_loss_backward, accumulate) before the one backward pass.
        ("rccl" also creates the communicator for a single process: the collective then degenerates to a copy.)
        optimizer: "adam" (train_segmentation.py:96, the shipped configs/training.yaml) or "sgd" -- the script's other branch,
        optim.SGD(lr, momentum=sgd_momentum, weight_decay) (:97-98; training.yaml:5-6): one fused kernel on the flat buffers either way."""
        if optimizer not in ("adam", "sgd"):
            raise ValueError(f"unknown optimizer {optimizer!r}: 'adam' or 'sgd'")
        self.optimizer, self.momentum = optimizer, float(momentum)
        params = list(model.named_parameters())
        if not params or not params[0][1].is_cuda:
            raise RuntimeError("move the model to a HIP device before building a Trainer (no CPU fallback)")
        self.model, self.group = model, process_group
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        dev = params[0][1].device
        self.device = dev
        ctx = model._context(dev)
        n = int(_lib.lib().mgu_unet_param_count(ctx.handle))
        if n != sum(p.numel() for _, p in params):
            raise RuntimeError("parameter count mismatch between the module tree and libmgunet")
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for name, p in params:  # named_parameters() order must be the library's flat order
            lo = int(_lib.lib().mgu_unet_param_offset(ctx.handle, name.encode()))
            if lo != off:
                raise RuntimeError(f"flat parameter layout mismatch at {name}: {lo} != {off}")
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        model.mark_parameters_changed()
        model._slots = None
        self.step_count = 0
        if loss not in ("ce", "ce+dice"):
            raise ValueError(f"unknown loss {loss!r}: 'ce' or 'ce+dice'")
        self.loss_kind, self.dice_smooth = loss, float(dice_smooth)
        self._loss = torch.zeros(1, device=dev, dtype=torch.float32)
        self._dice = torch.zeros(1, device=dev, dtype=torch.float32)
        self._rccl = False
        if comm not in ("auto", "rccl", None):
            raise ValueError(f"unknown comm {comm!r}: 'auto', 'rccl' or None")
        self.comm = comm
        if comm == "rccl":
            self.attach_rccl()

    def _dist_world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _dist_backend(self):
        import torch.distributed as dist
        return dist.get_backend(self.group) if dist.is_available() and dist.is_initialized() else None

    def attach_rccl(self) -> None:
        """ncclCommInitRank inside libmgunet for this rank's context.  Rank 0 draws the ncclUniqueId; the launcher's process
        group is used ONLY to hand its 128 bytes to the other ranks."""
        import torch.distributed as dist
        L, ctx = _lib.lib(), self.model._context(self.device)
        world, rank = self._dist_world(), (dist.get_rank(self.group) if self._dist_world() > 1 else 0)
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(L.mgu_comm_get_unique_id(buf), None)
        if world > 1:
            box = [bytes(buf.raw)]
            with torch.cuda.device(self.device):   # an NCCL group moves the pickled object through the CURRENT device
                dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            buf = C.create_string_buffer(box[0], 128)
        with torch.cuda.device(self.device):
            _lib.check(L.mgu_comm_init_rank(ctx.handle, buf, rank, world), ctx.handle)
        self._rccl = True

    def set_lr(self, lr: float) -> None:  # schedulers live on the host (train_segmentation.py:105,143): see StepLR below
        self.lr = lr

    # ---- torch.optim.Adam-compatible optimizer state (train_segmentation.py:154-164 saves optimizer.state_dict()) ------------
    def optimizer_state_dict(self) -> dict:
        """What torch.optim.Adam(model.parameters(), lr, weight_decay=wd).state_dict() would hold after the same steps: per
        parameter {'step', 'exp_avg', 'exp_avg_sq'} (views of the flat moment buffers, cloned) + one param_group.  Loadable into a
        real torch.optim.Adam, and back (load_optimizer_state_dict)."""
        if self.optimizer == "sgd":   # torch.optim.SGD's layout: {'momentum_buffer'} per parameter, one param_group
            params = [p for _, p in self.model.named_parameters()]
            state, off = {}, 0
            for i, p in enumerate(params):
                k = p.numel()
                if self.step_count > 0 and self.momentum != 0:
                    state[i] = {"momentum_buffer": self.exp_avg[off:off + k].view_as(p).clone()}
                off += k
            return {"state": state, "param_groups": [{"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.wd,
                                                      "nesterov": False, "maximize": False, "foreach": None, "differentiable": False,
                                                      "fused": None, "params": list(range(len(params)))}]}
        return adam_state_dict([p for _, p in self.model.named_parameters()], self.exp_avg, self.exp_avg_sq, self.step_count,
                               self.lr, self.betas, self.eps, self.wd)

    def load_optimizer_state_dict(self, sd: dict) -> None:
        params = [p for _, p in self.model.named_parameters()]
        g = sd["param_groups"][0]
        if self.optimizer == "sgd":
            if len(sd["param_groups"]) != 1 or list(g["params"]) != list(range(len(params))):
                raise ValueError("expected the single param_group of optim.SGD(model.parameters(), ...) (train_segmentation.py:98)")
            self.lr, self.momentum, self.wd = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
            off, have = 0, 0
            for i, p in enumerate(params):
                k = p.numel()
                st = sd["state"].get(i)
                if st is not None and st.get("momentum_buffer") is not None:
                    self.exp_avg[off:off + k].copy_(st["momentum_buffer"].reshape(-1))
                    have += 1
                off += k
            if have not in (0, len(params)):
                raise ValueError("a momentum buffer for some parameters only cannot be represented by the fused flat SGD")
            self.step_count = 2 if have else 0     # SGD keeps no step count: any value past the first step continues the buffers
            return
        if len(sd["param_groups"]) != 1 or list(g["params"]) != list(range(len(params))):
            raise ValueError("expected the single param_group of optim.Adam(model.parameters(), ...) (train_segmentation.py:96)")
        self.lr, self.betas, self.eps, self.wd = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        off, steps = 0, set()
        for i, p in enumerate(params):
            k = p.numel()
            st = sd["state"].get(i)
            if st is None:                                    # a parameter Adam has not stepped yet
                self.exp_avg[off:off + k].zero_()
                self.exp_avg_sq[off:off + k].zero_()
            else:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            off += k
        if len(steps) > 1:
            raise ValueError("parameters with different step counts cannot be represented by the fused flat Adam")
        self.step_count = steps.pop() if steps else 0

    def save_checkpoint(self, path: str, epoch: int, loss: float) -> None:
        """The reference's checkpoint dict (train_segmentation.py:158-163): readable by its own loaders
        (infer_segmentation.py:90-95, segmentation_performance.py:86-110) and by torch.optim.Adam."""
        torch.save({"epoch": epoch, "model_state_dict": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                    "optimizer_state_dict": self.optimizer_state_dict(), "loss": loss}, path)

    def load_checkpoint(self, path: str) -> dict:
        """Resume from either layout the reference writes: the checkpoint dict or a bare state_dict (:166-168).  Returns the dict."""
        ck = torch.load(path, map_location=self.device)
        sd = ck["model_state_dict"] if isinstance(ck, dict) and "model_state_dict" in ck else ck
        with torch.no_grad():
            own = dict(self.model.state_dict())
            for k, v in sd.items():
                own[k].copy_(v)                               # in place: parameters stay views of the flat buffer
        self.model.mark_parameters_changed()
        if isinstance(ck, dict) and "optimizer_state_dict" in ck:
            self.load_optimizer_state_dict(ck["optimizer_state_dict"])
        return ck if isinstance(ck, dict) else {"model_state_dict": ck}

    def forward_backward(self, images: torch.Tensor, masks: torch.Tensor, exchange: bool = False) -> torch.Tensor:
        """Fills self.grad with d(mean CE)/d(params) of this rank's shard; returns the loss (device scalar).
        exchange=True (needs attach_rccl): the gradient comes back already averaged over the ranks, the all-reduce having run
        bucket by bucket behind the layers still being differentiated."""
        model, dev = self.model, self.device
        if masks.dtype != torch.int64 or not masks.is_cuda:
            raise TypeError("masks must be an int64 tensor on the HIP device")
        model.train()
        logits, _, _ = model(images)
        B, Cc, H, W = logits.shape
        if tuple(masks.shape) != (B, H, W):
            raise ValueError(f"masks shape {tuple(masks.shape)} does not match logits {(B, H, W)}")
        ctx = model._context(dev)
        L = _lib.lib()
        nhwc = logits.permute(0, 2, 3, 1)
        assert nhwc.is_contiguous()
        masks = masks.contiguous()
        npix = B * H * W
        ldd = (Cc + 3) // 4 * 4
        dlogits = torch.empty((npix, ldd), device=dev, dtype=torch.float32)
        stream = _lib.current_stream_ptr(dev)
        loss = self._loss
        with torch.cuda.device(dev):
            _lib.check(L.mgu_cross_entropy(ctx.handle, nhwc.data_ptr(), masks.data_ptr(), npix, Cc, 1.0 / npix,
                                           dlogits.data_ptr(), self._loss.data_ptr(), stream), ctx.handle)
            if self.loss_kind == "ce+dice":   # loss = loss_ce + loss_dice (:130): d(dice)/d(logits) is ADDED to the CE gradient
                _lib.check(L.mgu_dice_loss_backward(ctx.handle, nhwc.data_ptr(), masks.data_ptr(), B, H * W, Cc, H * W * Cc, 1, Cc,
                                                    self.dice_smooth, 1.0, None, dlogits.data_ptr(), H * W * ldd, 1, ldd, 1,
                                                    self._dice.data_ptr(), stream), ctx.handle)
                loss = self._loss + self._dice
            bwd = L.mgu_unet_backward_allreduce if exchange else L.mgu_unet_backward
            _lib.check(bwd(ctx.handle, dlogits.data_ptr(), self.grad.data_ptr(), stream), ctx.handle)
        return loss

    def check(self) -> None:
        """Synchronise and raise ValueError if a kernel of this trainer met invalid data (a label outside [0, C) other than
        the ignore_index -100) -- the place torch would have raised; train_step itself never blocks the host."""
        ctx = self.model._context(self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgu_sync_check(ctx.handle, _lib.current_stream_ptr(self.device)), ctx.handle)

    def optimizer_step(self, grad_scale: float = 1.0) -> None:
        model, dev = self.model, self.device
        ctx = model._context(dev)
        self.step_count += 1
        with torch.cuda.device(dev):
            if self.optimizer == "sgd":   # the momentum buffer lives in exp_avg (exp_avg_sq is unused by this branch)
                _lib.check(_lib.lib().mgu_sgd_step(ctx.handle, self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                                   self.flat.numel(), self.lr, self.momentum, self.wd, self.step_count, grad_scale,
                                                   _lib.current_stream_ptr(dev)), ctx.handle)
            else:
                _lib.check(_lib.lib().mgu_adam_step(ctx.handle, self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                                    self.exp_avg_sq.data_ptr(), self.flat.numel(), self.lr, self.betas[0],
                                                    self.betas[1], self.eps, self.wd, self.step_count, grad_scale,
                                                    _lib.current_stream_ptr(dev)), ctx.handle)
        model.refresh_packed_weights(dev)   # same tensors, new contents: repack in place (no state_dict round trip per step)

    def train_step(self, images: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        if self._rccl:   # the only data-path collective of the build, issued from inside backward
            loss = self.forward_backward(images, masks, exchange=True)
            self.optimizer_step(1.0)
            return loss
        loss = self.forward_backward(images, masks)
        scale = allreduce_mean_(self.grad, self.group) if self.comm is not None else 1.0   # torch.distributed (RCCL / gloo); 1 rank: no-op
        self.optimizer_step(scale)
        return loss


class FlatAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) (the optimizer of both reference loops, train_end_to_end.py:228) for the
    parameters of ANY modules on the HIP path: they are re-homed as views of one flat fp32 buffer, their .grad as views of a flat
    gradient buffer that autograd accumulates into, and step() is one mgu_adam_step launch.  (`Trainer` does the same for the U-Net,
    whose flat order the library fixes; this is the graph branch's half.)"""

    def __init__(self, modules, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, exclude=()):
        """exclude: parameter-name fragments (as in named_parameters()) left out of the flat buffers -- parameters whose .grad stays
        None in the reference loop: torch.optim.Adam skips those entirely (no step count, no weight decay)."""
        seen, params = set(), []
        for m in modules:
            for name, q in m.named_parameters():
                if id(q) not in seen and q.requires_grad and not any(x in name for x in exclude):
                    seen.add(id(q))
                    params.append(q)
        if not params or not params[0].is_cuda:
            raise RuntimeError("move the modules to a HIP device before building the optimizer (no CPU fallback)")
        self.params, self.device = params, params[0].device
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        n = sum(q.numel() for q in params)
        self.flat = torch.empty(n, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(n, device=self.device, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=self.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=self.device, dtype=torch.float32)
        off = 0
        for q in params:
            k = q.numel()
            self.flat[off:off + k].copy_(q.detach().reshape(-1))
            q.data = self.flat[off:off + k].view_as(q)
            q.grad = self.grad[off:off + k].view_as(q)
            off += k
        self.step_count = 0

    def zero_grad(self) -> None:
        self.grad.zero_()          # the .grad views stay in place (set_to_none would detach them from the flat buffer)

    def step(self, grad_scale: float = 1.0) -> None:
        from .gat import _context
        ctx = _context(self.device)
        self.step_count += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgu_adam_step(ctx.handle, self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                                self.exp_avg_sq.data_ptr(), self.flat.numel(), self.lr, self.betas[0], self.betas[1],
                                                self.eps, self.wd, self.step_count, grad_scale, _lib.current_stream_ptr(self.device)),
                       ctx.handle)
        for q in self.params:      # the kernel wrote the parameters behind torch's back: the packed-weight caches key on _version
            torch.autograd.graph.increment_version(q)


class E2ETrainer:
    """One iteration of scripts/train_end_to_end.py:262-480 on the HIP path, as far as the reference's loss reaches:

        total = L_unet_seg + 0.1 * L_shape (the constant 0, :287) + w_f * L_feature + w_p * L_partition + w_s * L_smooth

    `L_unet_seg` (CrossEntropy of the U-Net logits) is the U-Net's only gradient source -- the script feeds the graph branch torch.randn
    placeholders (:326, :338, :342), not U-Net features -- and goes through `Trainer` (mgu_unet_forward / mgu_cross_entropy /
    mgu_unet_backward).  `L_feature` (FeatureConsistencyLoss on the patch GAT's output, with the batch dimension the script forgets,
    SURVEY appendix A) and `L_partition` (MinCutRefinement) differentiate the patch GAT and the segment predictor through the autograd
    nodes of mgunet.GATNetwork / MinCutRefinement / FeatureConsistencyLoss (mgu_gat_layer_backward, mgu_ncut_backward,
    mgu_feature_consistency_loss_backward).  `L_smooth` is the TV of a constant map (:455) and `L_shape` a constant: their gradient
    is exactly zero and they are not evaluated.
    The whole batch runs as ONE block-diagonal graph (the script loops over the images, :300-437): one patch-GAT launch sequence, one
    segment-predictor sequence and one feature loss for the B images, B normalized-cut nodes (the loss is a per-image quotient, :440).
    Every sub-model of the script's single Adam (:219-229) takes the step: the U-Net in `Trainer`, the patch GAT and the segment
    predictor in `FlatAdam` -- and, when they are handed in, the region GAT, FeatureFusion and DetectionHead too: in the reference
    they hang on the loss through `L_smooth`, whose gradient is zero, so torch gives them zero .grad tensors and Adam still applies
    the L2 term (g = weight_decay * p).
    With more than one rank BOTH flat gradients are averaged over the ranks before the step (the U-Net's by the Trainer's own
    exchange -- torch.distributed or the in-library RCCL buckets -- the graph branch's as one more all-reduce), so every rank holds
    the same parameters after it.  The placeholders are ARGUMENTS here (the script draws them from torch's RNG per image)."""

    def __init__(self, unet_trainer: Trainer, patch_gat: GATNetwork, segment_predictor, mincut, feature_loss, num_segments: int = 2,
                 l_feature_weight: float = 0.1, l_partition_weight: float = 0.5, extra_modules=(),
                 untouched_params=("fc_bbox", "fc_class_scores")):
        """extra_modules: the sub-models the loss does not reach but the script's optimizer holds (region GAT, FeatureFusion,
        DetectionHead, train_end_to_end.py:223-226): stepped with zero gradients, i.e. the weight-decay term alone.
        untouched_params: name fragments of parameters that are NOT on the path to pred_confidence, the one head output the script's
        loss touches (train_end_to_end.py:462) -- DetectionHead.fc_bbox and .fc_class_scores (detection_head.py:57,67): their .grad
        stays None in the reference, so its Adam never steps or decays them, and neither does this one."""
        self.unet = unet_trainer
        self.patch_gat, self.predictor, self.mincut, self.feature_loss = patch_gat, segment_predictor, mincut, feature_loss
        self.K, self.wf, self.wp = num_segments, l_feature_weight, l_partition_weight
        self.extra_modules = [m for m in extra_modules if any(True for _ in m.parameters())]
        self.graph_opt = FlatAdam([patch_gat, segment_predictor, *self.extra_modules], lr=unet_trainer.lr, weight_decay=unet_trainer.wd,
                                  betas=unet_trainer.betas, eps=unet_trainer.eps, exclude=tuple(untouched_params))
        self._batch_graph = None

    def _block_diagonal(self, edge_index: torch.Tensor, Np: int, B: int):
        """(edge_index of the B-image batch, graph_ptr): image b's nodes are [b * Np, (b + 1) * Np).  Cached per (edge_index, B)."""
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, Np, B)
        if self._batch_graph is None or self._batch_graph[0] != key:
            off = (torch.arange(B, device=edge_index.device, dtype=edge_index.dtype) * Np).view(B, 1, 1)
            ei = (edge_index.unsqueeze(0) + off).permute(1, 0, 2).reshape(2, -1).contiguous()
            gp = (torch.arange(B + 1, device=edge_index.device, dtype=torch.int64) * Np).to(torch.int32)
            self._batch_graph = (key, ei, gp, edge_index)
        return self._batch_graph[1], self._batch_graph[2]

    @staticmethod
    def exchange_gradients_(unet_grad: torch.Tensor, graph_grad: torch.Tensor, group=None, unet_exchanged: bool = False):
        """Mean all-reduce of both flat gradients over the ranks; returns the factors (1 / world, or 1.0 where nothing is left to
        scale) for the two Adam steps.  `unet_exchanged`: the U-Net gradient already came back averaged (in-library RCCL exchange)."""
        su = 1.0 if unet_exchanged else allreduce_mean_(unet_grad, group)
        sg = allreduce_mean_(graph_grad, group)
        return su, sg

    def step(self, images, masks, patch_features, f_unet_patches, patch_labels, edge_index) -> dict:
        """images (B,3,H,W), masks (B,H,W) int64; per image b (lists of B tensors, or stacked (B, Np, .) tensors): patch_features[b]
        (Np, D_in) -> patch GAT, f_unet_patches[b] (Np, D) and patch_labels[b] (Np,) for L_feature; edge_index: ONE image's patch
        graph (2, E), shared by the batch.  Returns the script's running-loss entries."""
        B = images.shape[0]
        stack = lambda v: v if isinstance(v, torch.Tensor) else torch.stack(list(v), 0)   # noqa: E731
        X, FU, Y = stack(patch_features), stack(f_unet_patches), stack(patch_labels)
        Np = X.shape[1]
        exchange = self.unet._rccl
        loss_seg = self.unet.forward_backward(images, masks, exchange=exchange)
        self.graph_opt.zero_grad()
        ei_all, gp = self._block_diagonal(edge_index, Np, B)
        h = self.patch_gat(X.reshape(B * Np, -1), ei_all, graph_ptr=gp)                               # :332, all images
        lf = self.feature_loss(FU, h.view(B, Np, -1), Y)                                              # :344 (sum over patches, mean over the batch = :440)
        if getattr(self.predictor, "use_gnn", False):
            seg_logits = self.predictor.gnn_predictor(h, ei_all, graph_ptr=gp)                        # :348-351 (:190)
        else:
            seg_logits = self.predictor(h)
        losses, _soft, _hard = self.mincut.forward_batched(h, edge_index, B, self.K, seg_logits.contiguous())
        lp = losses.mean()                                                                            # :441
        (self.wf * lf + self.wp * lp).backward()                                                      # the graph-branch part of :478
        if self.unet.comm is not None:
            su, sg = self.exchange_gradients_(self.unet.grad, self.graph_opt.grad, self.unet.group, unet_exchanged=exchange)
        else:
            su = sg = 1.0
        self.unet.optimizer_step(su)
        self.graph_opt.step(sg)
        total = loss_seg.detach() + self.wf * lf.detach() + self.wp * lp.detach()
        return {"total": total, "l_unet_seg": loss_seg.detach(), "l_shape": torch.zeros((), device=images.device),
                "l_feature": lf.detach(), "l_partition": lp.detach(), "l_smooth": torch.zeros((), device=images.device)}
