#!/usr/bin/env python3
"""Headline benchmark: segmented Mpix/s of the full MinGraph-UNet forward (BASELINE.json).

A "step" = one pass of the hot path over one synthetic batch per rank:
    U-Net forward (18 conv3x3 + 4 convT + 1x1) -> per-patch mean of decoder_feats[0] -> block-diagonal
    patch-graph GAT(32 -> 64, 4 heads) over the batch's B*1024 nodes.
Workload at N=1 = BASELINE configs[1]: batch 8 x 3x512x512 fp32 on one MI355X.  For N>1 each rank gets
its own 8 images (weak scaling, = configs[2]'s 8 images/GPU at N=8) and there is NO data-path collective
(images are independent) -- only the timing barrier.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "mingraph-unet_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: bf16 MFMA dense (never the 2:1-sparse figure)


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(batch, H, W, iters):
    """The oracle (CPU restatement of the reference path, same aten/oneDNN kernels the reference's
    modules dispatch to) timed on this box's host cores on a bounded sample of the same workload."""
    import mgunet_oracle as O
    threads = min(host_cores(), int(os.environ.get("MGU_CPU_THREADS", "16")))  # the 1-GPU box's CPU share is 16
    torch.set_num_threads(threads)
    p = O.make_unet_params(3, 2, 32, 4, seed=0)
    gp = O.make_gat_params(32, 128, 64, 4, 1, seed=0)
    ei = torch.from_numpy(O.patch_graph_edges(H, W, 16))
    x = torch.from_numpy(O.formula_normal("bench/cpu/x", (batch, 3, H, W), seed=1))

    def step():
        with torch.no_grad():
            lg, _, ft = O.unet_forward(p, x, 4)
            for b in range(batch):
                O.gat_network_forward(gp, O.patch_mean_features(ft[0][b], 16), ei, 4)
        return lg

    step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(batch * H * W / dt / 1e6, 4), "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{batch} images 3x{H}x{W} fp32 of the same workload, 1 warm-up + {iters} timed iterations "
                      f"({dt:.2f} s/iter), torch {torch.__version__} CPU"}


def max_over_ranks(dt, dist, dev):
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_train(a, world, rank, local_rank, dev, dist):
    """BASELINE configs[4]: training step, batch 32 over 8 GPUs = 4 images/GPU (weak scaling), fwd+bwd+Adam,
    mean all-reduce of the flat fp32 gradient over RCCL (DDP semantics).  Mpix/s = images*H*W per step time."""
    import mgunet
    import mgunet_oracle as O
    B = a.batch if a.batch != 8 else 4
    H = W = a.size
    model = mgunet.UNet(3, 2, 32, 4)
    model.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
    model = model.to(dev)
    tr = mgunet.Trainer(model, lr=1e-3, weight_decay=1e-4)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    x = torch.randn((B, 3, H, W), device=dev, generator=gen)
    y = torch.randint(0, 2, (B, H, W), device=dev, generator=gen)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        loss = tr.train_step(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_step(x, y)
    barrier()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(loss).all())
    dt = max_over_ranks(dt, dist, dev)
    if rank == 0:
        flops = 3.0 * model.flops(B, H, W)  # bwd = dgrad + wgrad ~ 2x fwd
        line = {"metric": "segmented Mpix/sec, MinGraph-UNet train step (fwd + CE + bwd + grad all-reduce + Adam), 512x512",
                "value": round(world * B * H * W * a.steps / dt / 1e6, 3), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic (N(0,1) images, random labels, formula weights)",
                "config": {"workload": f"BASELINE configs[4]: train step, {B} images/GPU x 3x{H}x{W} fp32, UNet(3,2,32,4), "
                                       f"CrossEntropy + Adam(1e-3, wd 1e-4), per-shard BatchNorm (DDP semantics)",
                           "images_per_gpu": B, "global_batch": B * world,
                           "parallelism": f"dp{world}: one flat {tr.flat.numel() * 4 / 1e6:.1f} MB fp32 gradient all-reduce (RCCL) per step"},
                "final_loss": round(float(loss), 6),
                "approx_tflops": round(flops * a.steps / dt / 1e12, 2), "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE config 2: 8)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: BASELINE configs[1]/[2] full forward (the headline); train: configs[4] train step "
                         "(fwd + CE + bwd + RCCL grad all-reduce + Adam), 4 images per GPU unless --batch is given")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: exact-fp32 MFMA (BASELINE configs[1], the headline); bf16: bf16 storage + fp32 accumulate "
                         "(configs[2]'s precision; reported as its own config, never as the fp32 number)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--no-profile-pass", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs `python -m torch.distributed.run --nproc-per-node {a.gpus}` (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # MGU_BENCH_REHEARSAL=1: several ranks share GPU 0 over gloo -- only to rehearse the torchrun plumbing on a
    # one-GPU box (never a measurement; RCCL needs one device per rank)
    rehearsal = os.environ.get("MGU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mgunet
    import mgunet_oracle as O
    from mgunet import _lib

    if a.mode == "train":
        return bench_train(a, world, rank, local_rank, dev, dist)
    B, H, W = a.batch, a.size, a.size
    unet = mgunet.UNet(3, 2, 32, 4, compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    unet.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))     # random-init-scale formula weights
    gat = mgunet.GATNetwork(32, 128, 64, 4, 1)
    gat.load_state_dict(O.make_gat_params(32, 128, 64, 4, 1, seed=0))
    model = mgunet.MinGraphUNet(unet.to(dev).eval(), gat.to(dev).eval(), 16).eval()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    x = torch.randn((B, 3, H, W), device=dev, generator=gen)           # synthetic batch, resident in HBM

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(a.warmup):
            out = model(x)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(a.steps):
            out = model(x)
        ev1.record()
        barrier()
        dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[3]).all())
    dt = max_over_ranks(dt, dist, dev)
    ms_step = dt / a.steps * 1e3
    mpix = world * B * H * W * a.steps / dt / 1e6

    # ---- roofline of the dominant kernel (the fp32 implicit-GEMM conv): HIP events recorded by the
    # library on the launch stream around every conv/convT/1x1/linear GEMM launch of the same steps
    roof = None
    if rank == 0 and not a.no_profile_pass:
        ctx = next(iter(unet._ctx.values()))
        L = _lib.lib()
        conv_ms = tot_ms = 0.0
        launches = 0
        nprof = min(a.steps, 10)
        with torch.no_grad():
            for _ in range(nprof):
                L.mgu_profile_enable(ctx.handle, 1)
                model.unet(x)
                cm, n, tm = C.c_double(), C.c_int(), C.c_double()
                _lib.check(L.mgu_profile_read(ctx.handle, C.byref(cm), C.byref(n), C.byref(tm)), ctx.handle)
                conv_ms += cm.value
                tot_ms += tm.value
                launches += n.value
            L.mgu_profile_enable(ctx.handle, 0)
        flops = unet.flops(B, H, W)
        exe_flops = float(L.mgu_unet_mfma_flops(ctx.handle, B, H, W))   # after the Winograd F(2x2,3x3) reduction
        ach = flops * nprof / (conv_ms * 1e-3) / 1e12
        exe = exe_flops * nprof / (conv_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernels from the committed PMC passes (rocprofv3 cannot run inside
        # this process); only quoted when the workload is the one they were collected on
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tj) and (B, H, W) == (8, 512, 512):
            t = json.load(open(tj))
            traffic, traffic_src = round(t["hbm_bytes_per_launch"]), "profiles/r01_traffic.json (" + t["correction"].split(" (")[0] + ")"
        peak = PEAK_BF16_MFMA_TFLOPS if a.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        if a.dtype == "bf16":
            traffic, traffic_src = None, None   # the committed PMC passes are fp32
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src, "algorithmic_flop_per_launch": round(flops / max(launches // nprof, 1)),
                "note": ("achieved = ALGORITHMIC (direct-convolution, 2*MAC) FLOPs / kernel time; the fp32 3x3 layers run as "
                         "Winograd F(2x2,3x3), which executes 2.25x fewer multiplies, so frac can exceed 1 -- "
                         "executed_* are the FLOPs the matrix cores really issue against the same peak"),
                "executed": round(exe, 2), "executed_frac": round(exe / peak, 4),
                "executed_flop_per_launch": round(exe_flops / max(launches // nprof, 1)),
                "kernel": ("wino3x3_f32_kernel (17 conv3x3) + conv3x3_first_kernel + igemm_kernel (4 ConvTranspose) + "
                           "patch_mean_kernel<float,2> (1x1 head fused with the patch means): the 23 conv launches of a step"
                           if a.dtype == "f32" else
                           "conv3x3_halo_kernel<bf16> + igemm_kernel<bf16>: the 23 conv launches of a step"),
                "launches_per_step": launches // nprof, "kernel_ms_per_step": round(conv_ms / nprof, 4),
                "unet_ms_per_step_with_events": round(tot_ms / nprof, 4), "gflop_per_step": round(flops / 1e9, 2)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(2, H, W, 14)   # ~12 s of host work on the GPU box's 16-core share

    if rank == 0:
        line = {"metric": "segmented Mpix/sec, full MinGraph-UNet forward (U-Net + patch-graph GAT), 512x512 batch",
                "value": round(mpix, 3), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": a.dtype, "data": "synthetic (N(0,1) images resident in HBM, formula random-init-scale weights)",
                "config": {"workload": (f"BASELINE configs[1]: batch {B}/GPU x 3x{H}x{W} fp32 full forward "
                                        if a.dtype == "f32" else
                                        f"BASELINE configs[2] precision: batch {B}/GPU x 3x{H}x{W}, bf16 storage + fp32 accumulate, full forward ")
                                       + "(UNet(3,2,32,4) + patch16 graph GAT(32->64,4 heads)), eval",
                           "images_per_gpu": B, "global_batch": B * world, "parallelism": f"batch-shard x{world}, no collective"},
                "gpu_event_ms_per_step": round(ev0.elapsed_time(ev1) / a.steps, 4),
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
