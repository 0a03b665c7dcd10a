"""The gradient exchange on REAL RCCL.  In-process cases use a one-rank communicator: librccl binds inside libmgunet.so,
ncclCommInitRank works from a host-passed id, the collective runs on the caller's stream in order, the bucketed exchange issued
from inside backward leaves the same gradient as backward alone (mean over 1 rank = identity), and a train step through it equals
the step without it.  The multi-rank cases start FRESH child processes under torch.distributed.run (tests/workers/rccl_ranks.py,
one rank per GPU): with two ranks where the box has two GPUs (skipped otherwise), and the same rank program with one rank on
any box.  The two-rank host logic (shard + mean) is also covered on CPU by test_dist_gloo.py.
BASELINE configs[4]; SURVEY 8b mgu_allreduce_grads, 8e."""
import ctypes as C
import os
import socket
import subprocess
import sys

import pytest
import torch

import mgunet
import mgunet_oracle as O
from mgunet import _lib

pytestmark = pytest.mark.gpu


def make_trainer(cuda, comm):
    cfg = (3, 2, 8, 2)
    m = mgunet.UNet(*cfg)
    m.load_state_dict(O.make_unet_params(*cfg, seed=9))
    return mgunet.Trainer(m.to(cuda), lr=1e-3, weight_decay=1e-4, comm=comm)


def test_allreduce_grads_one_rank_stream_order(cuda):
    L = _lib.lib()
    ctx = _lib.Context(0)
    uid = C.create_string_buffer(128)
    _lib.check(L.mgu_comm_get_unique_id(uid), None)
    assert any(uid.raw)
    _lib.check(L.mgu_comm_init_rank(ctx.handle, uid, 0, 1), ctx.handle)
    assert L.mgu_comm_world_size(ctx.handle) == 1 and L.mgu_comm_handle(ctx.handle)
    with pytest.raises(RuntimeError):                     # one communicator per context
        _lib.check(L.mgu_comm_init_rank(ctx.handle, uid, 0, 1), ctx.handle)
    n = 7_766_018                                         # the U-Net's flat gradient
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.zeros(n, device=cuda)
        for k in range(4):                                # producer kernels -> collective -> consumer, all on one stream
            g.add_(1.0)
            _lib.check(L.mgu_allreduce_grads(ctx.handle, g.data_ptr(), n, None, side.cuda_stream), ctx.handle)
            g.mul_(2.0)
        want = 0.0
        for k in range(4):
            want = (want + 1.0) * 2.0
    side.synchronize()
    assert float(g.min()) == want and float(g.max()) == want
    # an explicit ncclComm_t is honoured too
    _lib.check(L.mgu_allreduce_grads(ctx.handle, g.data_ptr(), 1000, L.mgu_comm_handle(ctx.handle), _lib.current_stream_ptr(cuda)), ctx.handle)
    torch.cuda.synchronize()
    _lib.check(L.mgu_comm_destroy(ctx.handle), ctx.handle)
    assert L.mgu_comm_world_size(ctx.handle) == 1 and not L.mgu_comm_handle(ctx.handle)
    with pytest.raises(RuntimeError):
        _lib.check(L.mgu_allreduce_grads(ctx.handle, g.data_ptr(), n, None, _lib.current_stream_ptr(cuda)), ctx.handle)


def test_train_steps_are_bitwise_reproducible(cuda):
    """Two trainers from the same state, three uninterrupted steps each: identical bytes (losses' gradients, parameters,
    BatchNorm running statistics).  Round 2 summed the small layers' weight gradients with float atomics and the per-channel
    statistics in 64 shared slots of double atomics; their summation order moved last bits from run to run, and behind train-mode
    BatchNorm + MaxPool near-ties two identical runs diverged at the second step (4.8e-5, then 7.8e-4).  Now every reduction has
    a fixed order: private partial panels (wgrad_f32.hip), one accumulator row per workgroup (train_kernels.hip)."""
    x = torch.from_numpy(O.formula_normal("rccl/x", (2, 3, 64, 48), seed=1)).to(cuda)
    y = torch.from_numpy(O.formula_labels("rccl/y", (2, 64, 48), 2, seed=2)).to(cuda)
    a, b = make_trainer(cuda, None), make_trainer(cuda, None)
    for step in range(3):
        la, lb = a.train_step(x, y).clone(), b.train_step(x, y).clone()
        torch.cuda.synchronize()
        assert torch.equal(a.grad, b.grad), (step, float((a.grad - b.grad).abs().max()))
        assert torch.equal(a.flat, b.flat) and float(la) == float(lb)
    sa, sb = a.model.state_dict(), b.model.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_backward_with_bucketed_exchange_equals_backward(cuda):
    """Three uninterrupted steps of a plain trainer and of one whose backward issues the bucketed RCCL exchange (one rank: mean =
    identity): the exchange must not change a single byte of the gradient or of the updated parameters."""
    x = torch.from_numpy(O.formula_normal("rccl/x", (2, 3, 64, 48), seed=1)).to(cuda)
    y = torch.from_numpy(O.formula_labels("rccl/y", (2, 64, 48), 2, seed=2)).to(cuda)
    plain, rccl = make_trainer(cuda, None), make_trainer(cuda, "rccl")
    assert rccl._rccl and not plain._rccl
    for step in range(3):
        l0 = plain.train_step(x, y).clone()
        l1 = rccl.train_step(x, y).clone()                 # backward + overlapped buckets + Adam behind the join
        torch.cuda.synchronize()
        d = float((plain.grad - rccl.grad).abs().max())
        print(f"[rccl step {step}] max|grad diff| plain vs rccl {d:.3e}, max|grad| {float(plain.grad.abs().max()):.3e}")
        assert float(l0) == float(l1)
        assert torch.equal(plain.grad, rccl.grad) and torch.equal(plain.flat, rccl.flat), (step, d)
    rccl.check()


def run_rank_program(nranks):
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "workers", "rccl_ranks.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)   # children are fresh processes: never an exec of this one
    assert r.returncode == 0 and "RCCL_RANKS_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_rank_program_one_rank():
    """The multi-rank rank program with world size 1 (what a one-GPU box can hold): exercises the spawn plumbing, the id
    broadcast, the bucketed exchange and the cross-rank assertions of the two-rank case in their degenerate form."""
    out = run_rank_program(1)
    print(out[out.index("RCCL_RANKS_OK"):].strip()[:400])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the box (one rank per GPU)")
def test_two_rank_backward_allreduce():
    """Two ranks, two GPUs: mgu_unet_backward_allreduce (bucket order and ranges, ncclAvg, cross-stream events, the final join)
    against backward + torch.distributed all-reduce on different shards; identical parameters on both ranks after Adam."""
    out = run_rank_program(2)
    print(out[out.index("RCCL_RANKS_OK"):].strip()[:400])
