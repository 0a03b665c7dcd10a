"""Rank program of tests/test_gpu_rccl.py's multi-rank cases (started under `python -m torch.distributed.run`, one rank per
GPU; never collected by pytest).  Every rank holds the same initial UNet and its OWN shard of the batch and checks:

  * libmgunet's bucketed exchange issued from inside backward (Trainer(comm="rccl") -> mgu_unet_backward_allreduce: buckets on
    the communicator stream, ncclAvg, the final join) leaves the MEAN over the ranks of the single-rank gradients -- the yardstick
    is `mgu_unet_backward` on the same shard followed by a torch.distributed all-reduce of the flat gradient;
  * loss is the shard's own loss (no collective touches it);
  * after Adam every rank holds bit-identical parameters (the averaged gradient is the same bytes everywhere).

World size 1 degenerates to mean = identity: that is what a one-GPU box can run, and it runs this very file."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "mingraph-unet_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import mgunet  # noqa: E402
import mgunet_oracle as O  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = (3, 2, 8, 2)

    def trainer(comm):
        m = mgunet.UNet(*cfg)
        m.load_state_dict(O.make_unet_params(*cfg, seed=9))
        return mgunet.Trainer(m.to(dev), lr=1e-3, weight_decay=1e-4, comm=comm)

    # each rank its own two images (DDP semantics: per-shard BatchNorm statistics)
    x = torch.from_numpy(O.formula_normal(f"rccl2/x/{rank}", (2, 3, 64, 48), seed=11)).to(dev)
    y = torch.from_numpy(O.formula_labels(f"rccl2/y/{rank}", (2, 64, 48), 2, seed=12)).to(dev)
    plain, ex = trainer(None), trainer("rccl")
    assert ex._rccl and not plain._rccl
    report = {"rank": rank, "world": world, "steps": []}
    for step in range(2):
        assert torch.equal(ex.flat, plain.flat)            # both trainers took the same (bitwise reproducible) steps so far
        l_plain = plain.forward_backward(x, y).clone()
        want = plain.grad.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)        # yardstick: torch's own RCCL all-reduce of the single-rank gradients
        want /= world
        l_ex = ex.forward_backward(x, y, exchange=True).clone()
        torch.cuda.synchronize(dev)
        gmax = float(want.abs().max())
        d = float((ex.grad - want).abs().max())
        assert abs(float(l_plain) - float(l_ex)) <= 1e-6 * abs(float(l_plain)), (float(l_plain), float(l_ex))
        assert d <= 2e-6 * gmax, (step, d, gmax)           # ncclAvg inside RCCL vs SUM then / world: one rounding apart
        if world > 1:                                      # the exchange really mixed the shards
            own = float((ex.grad - plain.grad).abs().max())
            assert own > 1e-3 * gmax, own
        # Adam on the averaged gradient: every rank must end with the same bytes
        plain.grad.copy_(ex.grad)
        plain.optimizer_step(1.0)
        ex.optimizer_step(1.0)
        torch.cuda.synchronize(dev)
        ref = ex.flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, ex.flat), "parameters differ between ranks after the exchanged step"
        assert torch.equal(plain.flat, ex.flat)
        report["steps"].append({"loss": float(l_ex), "max_grad_diff": d, "max_grad": gmax})
    ex.check()
    dist.barrier(device_ids=[local])
    if rank == 0:
        print("RCCL_RANKS_OK " + json.dumps(report), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
