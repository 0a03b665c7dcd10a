"""world_size-2 gloo tests (CPU) of the N>1 logic: batch sharding with no data-path collective for
inference, and the mean all-reduce of the flat gradient for the train step (DDP semantics: per-shard
BatchNorm statistics, gradients averaged).  The oracle plays the per-rank compute here -- the HIP path
needs a GPU -- so this checks exactly the host logic the multi-GPU bench relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mgunet
import mgunet_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, gb = (3, 2, 8, 2), 4
        p = O.make_unet_params(*cfg, seed=31)
        x = torch.from_numpy(O.formula_normal("dist/x", (gb, 3, 32, 32), seed=31))
        y = torch.from_numpy(O.formula_labels("dist/y", (gb, 32, 32), 2, seed=32))
        lo, hi = mgunet.shard_batch(gb, rank, world)
        # inference: independent shards, results gathered only for the check (no data-path collective)
        with torch.no_grad():
            lg = O.unet_forward(p, x[lo:hi], 2)[0]
        parts = [torch.zeros_like(lg) for _ in range(world)]
        dist.all_gather(parts, lg)
        # training: per-shard loss/grad, flat gradient mean all-reduce (the build's one collective)
        loss, grads, _, _, _, _ = O.train_step(p, x[lo:hi], y[lo:hi], 2)
        names = list(grads.keys())
        flat = torch.cat([grads[k].reshape(-1) for k in names])
        scale = mgunet.allreduce_mean_(flat)
        flat = flat * scale
        q.put((rank, torch.cat(parts).numpy(), flat.numpy(), float(loss), scale))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_inference_and_grad_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    cfg, gb = (3, 2, 8, 2), 4
    p = O.make_unet_params(*cfg, seed=31)
    x = torch.from_numpy(O.formula_normal("dist/x", (gb, 3, 32, 32), seed=31))
    y = torch.from_numpy(O.formula_labels("dist/y", (gb, 32, 32), 2, seed=32))
    with torch.no_grad():
        full = O.unet_forward(p, x, 2)[0].numpy()
    for r in res:                                     # eval: sharded == unsharded, every rank sees the same
        assert np.abs(r[1] - full).max() <= 1e-5 and r[4] == 0.5
    assert np.array_equal(res[0][2], res[1][2])       # all ranks hold the same averaged gradient
    # DDP semantics: mean of the per-shard gradients (per-shard BatchNorm statistics)
    gs = []
    for rk in range(world):
        lo, hi = mgunet.shard_batch(gb, rk, world)
        _, g, _, _, _, _ = O.train_step(p, x[lo:hi], y[lo:hi], 2)
        gs.append(torch.cat([g[k].reshape(-1) for k in g]))
    assert np.allclose(res[0][2], ((gs[0] + gs[1]) / 2).numpy(), rtol=1e-5, atol=1e-7)


def test_allreduce_mean_is_identity_without_process_group():
    t = torch.arange(4.0)
    assert mgunet.allreduce_mean_(t) == 1.0 and torch.equal(t, torch.arange(4.0))


def _graph_worker(rank, world, port, q):
    """Graph-branch half of an end-to-end iteration on this rank's shard (oracle compute), then E2ETrainer's gradient exchange."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params, flat = _graph_branch_grads(rank)
        unet_grad = torch.full((8,), float(rank + 1))                # stands for the U-Net's flat gradient
        su, sg = mgunet.E2ETrainer.exchange_gradients_(unet_grad, flat, None)
        new = _adam_step(params, flat * sg)
        q.put((rank, unet_grad.numpy() * su, flat.numpy() * sg, torch.cat([v.reshape(-1) for v in new.values()]).numpy(), su, sg))
    finally:
        dist.destroy_process_group()


def _graph_branch_grads(shard):
    """d(0.1 L_feature + 0.5 L_partition)/d(patch GAT, segment predictor) on one image of shard `shard` (scripts/train_end_to_end.py:332-351)."""
    Dp, K, H = 32, 2, 64
    ei = torch.from_numpy(O.patch_graph_edges(H, H, 16))
    Np = (H // 16) ** 2
    gp = {("gat." + k): torch.nn.Parameter(v.clone()) for k, v in O.make_gat_params(Dp, 16, Dp, 2, 1, seed=61).items()}
    pp = {k: torch.nn.Parameter(v.clone()) for k, v in O.make_segment_predictor_params(Dp, K, 16, True, 2, seed=62).items()}
    x = torch.from_numpy(O.formula_normal(f"distg/{shard}/patch", (Np, Dp), seed=63 + shard)) * 0.4
    fu = torch.from_numpy(O.formula_normal(f"distg/{shard}/funet", (Np, Dp), seed=73 + shard)) * 0.4
    y = torch.from_numpy((O.formula_uniform(f"distg/{shard}/y", (Np,), 0.0, 1.0, 83 + shard) > 0.5).astype(np.int64))
    h = O.gat_network_forward({k[4:]: v for k, v in gp.items()}, x, ei, 2, 1)
    lf = O.feature_consistency_loss(fu[None], h[None], y[None])
    lp, _, _ = O.mincut_forward(h, ei, K, O.segment_predictor_forward(pp, h, ei, True, 2))
    (0.1 * lf + 0.5 * lp).backward()
    params = {**gp, **pp}
    return params, torch.cat([v.grad.reshape(-1) for v in params.values()]).detach().clone()


def _adam_step(params, flat_grad, lr=1e-3, wd=1e-4):
    ps = [torch.nn.Parameter(v.detach().clone()) for v in params.values()]
    off = 0
    for p in ps:
        p.grad = flat_grad[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    torch.optim.Adam(ps, lr=lr, weight_decay=wd).step()
    return {k: p.detach() for k, p in zip(params, ps)}


@pytest.mark.timeout(300)
def test_two_rank_e2e_exchange_keeps_graph_branch_parameters_identical():
    """E2ETrainer averages the graph branch's flat gradient (patch GAT + segment predictor) over the ranks like the U-Net's
    (scripts/train_end_to_end.py:219-229: ONE optimizer over every sub-model): after the step both ranks hold the same parameters,
    the ones a single process gets from the mean of the two shard gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert res[0][4] == res[0][5] == 0.5
    assert np.array_equal(res[0][1], res[1][1]) and np.allclose(res[0][1], 1.5)   # the U-Net gradient: mean of 1 and 2
    assert np.array_equal(res[0][2], res[1][2])                                    # the graph gradient: identical on both ranks ...
    assert np.array_equal(res[0][3], res[1][3])                                    # ... and so are the parameters after Adam
    g0, g1 = _graph_branch_grads(0)[1], _graph_branch_grads(1)[1]
    assert float((g0 - g1).abs().max()) > 1e-6                                    # (the shards really differ)
    assert np.allclose(res[0][2], ((g0 + g1) / 2).numpy(), rtol=1e-6, atol=1e-9)
    want = _adam_step(_graph_branch_grads(0)[0], (g0 + g1) / 2)
    assert np.allclose(res[0][3], torch.cat([v.reshape(-1) for v in want.values()]).numpy(), rtol=0, atol=1e-7)
    # without the exchange the two ranks would have diverged
    lone = [_adam_step(_graph_branch_grads(r)[0], _graph_branch_grads(r)[1]) for r in range(2)]
    assert float((torch.cat([v.reshape(-1) for v in lone[0].values()]) - torch.cat([v.reshape(-1) for v in lone[1].values()])).abs().max()) > 1e-4
