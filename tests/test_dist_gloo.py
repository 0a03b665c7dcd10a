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
