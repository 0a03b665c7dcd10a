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

The JSON line's `roofline` describes ONE kernel -- the one the step spends most of its time in -- measured live with HIP
events the library records on the launch stream right around every launch (mgu_profile_read_kernels), in a separate pass of
the same steps (events off during the timed region so they cannot perturb `value`):
    achieved = FLOPs the kernel ISSUES on its matrix pipe per launch / its average launch duration,  frac = achieved / peak.
The fp32 3x3 layers run as Winograd F(2x2,3x3) (16 multiplies per 2x2 tile and channel pair instead of 36) with every fp32
operand split exactly into three bf16 pieces (six bf16 MFMA products per fp32 product, fp32 accumulate): `frac` is against
the bf16 MFMA peak that pipe has; the algorithmic (direct-convolution) rate and the fp32-equivalent rate are reported next
to it and never as `frac`.  `gat` is the graph kernel's record (HBM bound), at the headline batch and at 64 graphs.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "mingraph-unet_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: bf16 MFMA dense (never the 2:1-sparse figure)
PEAK_HBM_TBS = 8.0              # same guide: HBM3E spec peak (6.3 TB/s achievable by a streaming copy)
ARITH_3PIECE = "3xbf16 split (fp32 operands split exactly into 3 bf16 pieces, 6 bf16 MFMA products per fp32 product), fp32 accumulate"
PIPE = {0: ("fp32 MFMA (v_mfma_f32_32x32x2_f32)", PEAK_F32_MFMA_TFLOPS), 1: ("bf16 MFMA (v_mfma_f32_32x32x16_bf16)", PEAK_BF16_MFMA_TFLOPS)}


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


def cpu_threads():
    return min(host_cores(), int(os.environ.get("MGU_CPU_THREADS", "16")))  # the 1-GPU box's CPU share is 16


def cpu_baseline(batch, H, W, iters):
    """The oracle (CPU restatement of the reference path, same aten/oneDNN kernels the reference's
    modules dispatch to) timed on this box's host cores on a bounded sample of the same workload."""
    import mgunet_oracle as O
    torch.set_num_threads(cpu_threads())
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


def cpu_baseline_train(H, W, iters):
    """The oracle's train step (train-mode forward, CrossEntropy, torch autograd backward, Adam with L2) on ONE image of the
    same workload on the host cores."""
    import mgunet_oracle as O
    torch.set_num_threads(cpu_threads())
    p = O.make_unet_params(3, 2, 32, 4, seed=0)
    x = torch.from_numpy(O.formula_normal("bench/cpu/tx", (1, 3, H, W), seed=1))
    y = torch.from_numpy(O.formula_labels("bench/cpu/ty", (1, H, W), 2, seed=2))
    O.train_step(p, x, y, 4)
    t0 = time.perf_counter()
    for _ in range(iters):
        O.train_step(p, x, y, 4)
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(H * W / dt / 1e6, 4), "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 image 3x{H}x{W} fp32, the oracle's train step (train-mode forward + CE + autograd backward + Adam), "
                      f"1 warm-up + {iters} timed iterations ({dt:.2f} s/iter), torch {torch.__version__} CPU"}


def max_over_ranks(dt, dist, dev):
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def merge_stats(*lists):
    out = {}
    for lst in lists:
        for k in lst:
            e = out.setdefault(k["name"], {"name": k["name"], "ms": 0.0, "flops_alg": 0.0, "flops_mfma": 0.0, "launches": 0, "pipe": k["pipe"]})
            for f in ("ms", "flops_alg", "flops_mfma", "launches"):
                e[f] += k[f]
    return sorted(out.values(), key=lambda e: -e["ms"])


def roofline_from(stats, nprof, arithmetic, traffic=None):
    """The dominant kernel's record + the per-family table.  stats: merged mgu_profile_read_kernels over nprof steps."""
    mf = [k for k in stats if k["pipe"] >= 0 and k["flops_mfma"] > 0]
    if not mf:
        return None
    dom = mf[0]
    pipe_name, peak = PIPE[dom["pipe"]]
    t = dom["ms"] * 1e-3
    ach = dom["flops_mfma"] / t / 1e12
    fp32_equiv = dom["flops_mfma"] / (6.0 if ("wino" in dom["name"] and dom["pipe"] == 1) else 1.0)
    table = [{"kernel": k["name"], "launches_per_step": round(k["launches"] / nprof, 2), "ms_per_step": round(k["ms"] / nprof, 4),
              "avg_launch_us": round(k["ms"] * 1e3 / k["launches"], 2),
              "issued_tflops": round(k["flops_mfma"] / (k["ms"] * 1e-3) / 1e12, 2) if k["flops_mfma"] else None,
              "algorithmic_tflops": round(k["flops_alg"] / (k["ms"] * 1e-3) / 1e12, 2) if k["flops_alg"] else None,
              "pipe": {0: "f32", 1: "bf16", -1: "valu/hbm"}[k["pipe"]]} for k in stats]
    conv = [k for k in stats if k["flops_alg"] > 0]
    conv_ms = sum(k["ms"] for k in conv)
    # `bound` names the roofline the kernel is priced against (the matrix pipe it issues on); `limiter` says what the evidence shows
    # holds it there: below half of that pipe's peak (and, from the committed PMC passes, below half of the HBM peak too) it is the
    # latency / issue structure of the loop, not a saturated unit (the phase timeline of the wide Winograd kernel)
    frac_issued = ach / peak
    limiter = "mfma" if frac_issued >= 0.5 else "latency/issue (neither the matrix pipe nor HBM above 50 %)"
    return {"bound": "mfma", "limiter": limiter, "kernel": dom["name"], "launches_per_step": round(dom["launches"] / nprof, 2),
            "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "pipe": pipe_name, "arithmetic": arithmetic,
            "issued_flop_per_launch": round(dom["flops_mfma"] / dom["launches"]),
            "algorithmic_flop_per_launch": round(dom["flops_alg"] / dom["launches"]),
            "algorithmic_tflops": round(dom["flops_alg"] / t / 1e12, 2),
            "algorithmic_frac": round(dom["flops_alg"] / t / 1e12 / peak, 4),
            "fp32_equivalent_tflops": round(fp32_equiv / t / 1e12, 2),
            "fp32_equivalent_frac_of_fp32_mfma_peak": round(fp32_equiv / t / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic[0] if traffic else None, "traffic_unit": "HBM bytes per launch of this kernel (PMC FETCH_SIZE x2 + WRITE_SIZE)",
            "traffic_source": traffic[1] if traffic else None,
            "all_conv_kernels": {"ms_per_step": round(conv_ms / nprof, 4), "launches_per_step": round(sum(k["launches"] for k in conv) / nprof, 2),
                                 "algorithmic_tflops": round(sum(k["flops_alg"] for k in conv) / (conv_ms * 1e-3) / 1e12, 2),
                                 "gflop_per_step_algorithmic": round(sum(k["flops_alg"] for k in conv) / nprof / 1e9, 2)},
            "kernels": table,
            "note": ("achieved = FLOPs ISSUED on the kernel's matrix pipe per launch / average launch duration (HIP events on the launch "
                     "stream, instrumented pass); frac = achieved / that pipe's dense peak.  algorithmic_* count the direct-convolution "
                     "2*MAC of the operator (Winograd executes 16/36 of them); fp32_equivalent_* count the Winograd multiplies once "
                     "(not x6 for the three bf16 pieces) against the fp32 MFMA peak -- neither is `frac`.  algorithmic_frac = "
                     "algorithmic_tflops / the peak of the pipe used (SURVEY 8d's definition of MFMA utilisation).")}


def load_traffic(kernel, dtype):
    """HBM bytes per launch of `kernel` from this round's committed PMC passes (rocprofv3 cannot run inside this process)."""
    tj = next((q for q in (os.path.join(ROOT, "profiles", f"{r}_traffic_{dtype}.json") for r in ("r05", "r04", "r03", "r02")) if os.path.exists(q)), None)
    if tj is None:
        return None
    t = json.load(open(tj))
    # a stamped file (round 5 on) names the library build its counters were taken from: a different library -> the bytes describe other
    # kernels, so the field is dropped rather than quoted stale
    if t.get("lib_build_id") is not None:
        import hashlib
        from mgunet import _lib as _l
        cur = hashlib.sha256(open(_l.LIB_PATH, "rb").read()).hexdigest()[:16]
        if cur != t["lib_build_id"]:
            return None
    base = kernel.split("<")[0].split(" ")[0]
    args = kernel.split("<")[1].split(">")[0].replace(" ", "") if "<" in kernel else ""
    if base.startswith("mgu_wino_cp"):     # assembly kernels: the profiler knows them by their symbol, the label's "(asm form of ...<2>)" is prose
        args = ""
    exact = bool(args) and all(ch.isdigit() or ch == "," for ch in args)    # literal template arguments: one instantiation
    n = b = 0
    for k, v in t.get("kernels", {}).items():
        kk = k.replace(" ", "")
        if base in k and (not exact or ("<" + args + ">") in kk or ("<" + args + ",") in kk):   # leading template arguments
            n += v["launches"]
            b += v["launches"] * v["hbm_bytes_per_launch"]
    if not n:
        return None
    return round(b / n), f"profiles/{os.path.basename(tj)} ({t.get('correction', '')})"


def gat_record(dev, L, _lib):
    """The graph layer on synthetic node features, both schedules -- aggregate-first (Fin <= F': the attention-weighted sum is taken
    over the INPUT rows, gat_fused.hip) and the Wh-row gather (gat.hip; forced with MGU_NO_GAT_FUSED=1 in a context of its own so
    that the same layer is measured) -- at three sizes: 8 and 64 graphs of the 512^2 / patch-16 grid (1024 nodes, 3968 edges each;
    64 graphs = the north-star point) with the patch GAT (32 -> 4 x 64), and BASELINE configs[3]'s stress graphs (32 graphs of 2048
    nodes, in-degree exactly 8, random sources; GAT 64 -> 4 x 64)."""
    import mgunet
    import mgunet_oracle as O
    out = {}
    graph = mgunet.PatchGraphConstructor(16)
    heads, Fh = 4, 64
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)

    def patch_case(G):
        rowptr, col, gp, N1, E1 = graph.batched_csr(512, 512, G, dev)
        return f"graphs_{G}", 32, rowptr, col, gp, N1 * G, E1 * G, G

    def c4_case():
        G, N1, deg = 32, 2048, 8
        src = torch.randint(0, N1, (G, N1 * deg), device=dev, generator=gen) + (torch.arange(G, device=dev) * N1)[:, None]
        return ("c4_32x2048_deg8", 64, (torch.arange(G * N1 + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32),
                src.reshape(-1).to(torch.int32), (torch.arange(G + 1, device=dev, dtype=torch.int64) * N1).to(torch.int32), G * N1, G * N1 * deg, G)

    for key, Fin, rowptr, col, gp, N, E, G in (patch_case(8), patch_case(64), c4_case()):
        X = torch.randn((N, Fin), device=dev, generator=gen)
        rec = {"graphs": G, "nodes": N, "edges": E, "Fin": Fin, "heads": heads, "Fout_per_head": Fh}
        # the schedule mgu_gat_prepare PICKS for these layers (Fin <= F': aggregate first) leads; the Wh-row gather is the same layer
        # forced onto the other schedule (MGU_NO_GAT_FUSED=1), for comparison
        for sched, env, product in (("aggregate_first", {}, True), ("wh_row_gather", {"MGU_NO_GAT_FUSED": "1"}, False)):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            ctx = _lib.Context(dev.index or 0)
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            gp_ = O.make_gat_params(Fin, 128, 64, 4, 1, seed=0)
            W = torch.cat([gp_[f"gat_layers.0.heads.{h}.W.weight"] for h in range(heads)], 0).contiguous().to(dev)
            a = torch.cat([gp_[f"gat_layers.0.heads.{h}.a.weight"] for h in range(heads)], 0).contiguous().to(dev)
            hnd = C.c_void_p()
            s = _lib.current_stream_ptr(dev)
            _lib.check(L.mgu_gat_prepare(ctx.handle, W.data_ptr(), a.data_ptr(), heads, Fh, Fin, 1, C.byref(hnd), s), ctx.handle)
            y = torch.empty((N, Fh), device=dev)

            def run():
                _lib.check(L.mgu_gat_layer_forward_prepared(ctx.handle, hnd, X.data_ptr(), N, rowptr.data_ptr(), col.data_ptr(), E,
                                                            gp.data_ptr(), G, 0, 0.2, y.data_ptr(), s), ctx.handle)
            for _ in range(5):
                run()
            torch.cuda.synchronize(dev)
            reps = 20
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(reps):
                run()
            ev1.record()
            torch.cuda.synchronize(dev)
            layer_us = ev0.elapsed_time(ev1) * 1e3 / reps
            L.mgu_profile_enable(ctx.handle, 1)
            for _ in range(reps):
                run()
            ks = _lib.read_kernel_stats(ctx)
            L.mgu_profile_enable(ctx.handle, 0)
            kern = {k["name"]: round(k["ms"] * 1e3 / reps, 2) for k in ks}
            csr = (N + 1 + E) * 4
            if sched == "aggregate_first":
                # gat_stmax: X + CSR read, st written;  gat_fused2: X (the gathered rows: every row once), st, CSR read, out written
                comp = (N * Fin * 4 + csr + N * 2 * heads * 4) + (N * Fin * 4 + csr + N * 2 * heads * 4 + N * Fh * 4)
                dom_name = "gat_fused2_kernel"
                dom_bytes = N * Fin * 4 + csr + N * 2 * heads * 4 + N * Fh * 4
            else:
                # GEMM: X read, Wh + st written;  edge max: st + CSR read;  aggregate: Wh table + st + CSR + node->graph read, out written
                comp = (N * Fin * 4 + N * heads * Fh * 4 + N * 2 * heads * 4) + (N * 2 * heads * 4 + csr + N * 4) + \
                       (N * heads * Fh * 4 + N * 2 * heads * 4 + csr + N * 4 + N * Fh * 4)
                dom_name = "gat_aggregate_kernel"
                dom_bytes = N * heads * Fh * 4 + N * 2 * heads * 4 + csr + N * 4 + N * Fh * 4   # Wh table + s,t + CSR + out
            dom_us = kern.get(dom_name)
            logical = E * (4 + (Fin if sched == "aggregate_first" else heads * Fh) * 4 + 4)
            rec[sched] = {"product_schedule": product, "launches": len(ks), "layer_us": round(layer_us, 2),
                          "layer_compulsory_bytes": comp, "layer_TBps": round(comp / (layer_us * 1e-6) / 1e12, 3),
                          "layer_frac_of_hbm_peak": round(comp / (layer_us * 1e-6) / 1e12 / PEAK_HBM_TBS, 4),
                          "kernel_us": kern, "dominant_kernel": dom_name, "compulsory_bytes": dom_bytes,
                          "achieved_TBps": round(dom_bytes / (dom_us * 1e-6) / 1e12, 3) if dom_us else None,
                          "frac_of_hbm_peak": round(dom_bytes / (dom_us * 1e-6) / 1e12 / PEAK_HBM_TBS, 4) if dom_us else None,
                          "logical_gather_bytes": logical,
                          "logical_gather_TBps": round(logical / (dom_us * 1e-6) / 1e12, 3) if dom_us else None}
            L.mgu_gat_release(ctx.handle, hnd)
            del ctx
        out[key] = rec
    out["note"] = ("product_schedule: the schedule the library selects for the layer (both BASELINE graph layers: aggregate first); "
                   "layer_*: every array each kernel of the layer must touch once / the layer's time back to back (events); "
                   "compulsory_bytes: every array the dominant kernel must touch once (node table or input rows, attention scalars, CSR, "
                   "node->graph ids, output); at these sizes the tables (8-67 MB) stay in the 256 MB Infinity Cache between the producer "
                   "and the gather, so the ceiling is the cache, not HBM -- the HBM fraction the north star asks for is quoted as-is, "
                   "against the 8 TB/s spec peak; logical_gather_*: one source row per edge (what the reference's index / scatter_add "
                   "formulation moves)")
    return out


def bench_train(a, world, rank, local_rank, dev, dist):
    """BASELINE configs[4]: training step, batch 32 over 8 GPUs = 4 images/GPU (weak scaling), fwd+bwd+Adam,
    mean all-reduce of the flat fp32 gradient over RCCL (DDP semantics).  Mpix/s = images*H*W per step time."""
    import mgunet
    import mgunet_oracle as O
    from mgunet import _lib
    B = a.batch if a.batch is not None else 4
    H = W = a.size
    model = mgunet.UNet(3, 2, 32, 4)
    model.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
    model = model.to(dev)
    # the gradient exchange is libmgunet's own RCCL communicator (mgu_unet_backward_allreduce: buckets overlapped with backward);
    # at one rank it is created too, so that the single-GPU number contains the (degenerate) collective calls
    rehearsal = dist is not None and dist.get_backend() != "nccl"
    # one rank: libmgunet's own communicator (degenerate collective, so the single-GPU number contains the calls); several ranks:
    # torch.distributed's RCCL all-reduce after backward unless --comm rccl opts in to the overlapped in-library exchange
    comm = a.comm or ("rccl" if (world == 1 and not rehearsal) else "auto")
    tr = mgunet.Trainer(model, lr=1e-3, weight_decay=1e-4, comm="auto" if rehearsal else comm)
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
    tr.check()
    dt = max_over_ranks(dt, dist, dev)
    roof = cpu = None
    if not a.no_profile_pass:
        # the instrumented pass steps on EVERY rank (a train step holds a collective); only rank 0 records
        ctx = model._context(dev)
        L = _lib.lib()
        nprof = min(a.steps, 5)
        if rank == 0:
            L.mgu_profile_enable(ctx.handle, 1)
        for _ in range(nprof):
            tr.train_step(x, y)
        barrier()
    if rank == 0 and not a.no_profile_pass:
        stats = merge_stats(_lib.read_kernel_stats(ctx))
        L.mgu_profile_enable(ctx.handle, 0)
        dom = next((k for k in stats if k["pipe"] >= 0 and k["flops_mfma"] > 0), None)
        x3w = any(k["name"].startswith("wino_wgrad") and k["pipe"] == 1 for k in stats)
        roof = roofline_from(stats, nprof, "fp32 operands split exactly into 3 bf16 pieces (6 bf16 MFMA products per fp32 product), fp32 "
                             "accumulate, for the forward / data-gradient Winograd convolutions" +
                             (" and the Winograd weight gradients" if x3w else "; exact fp32 MFMA for the weight gradients"),
                             load_traffic(dom["name"], "train") if dom else None)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline_train(H, W, 4)
    if rank == 0:
        flops = 3.0 * model.flops(B, H, W)  # bwd = dgrad + wgrad ~ 2x fwd
        line = {"metric": "segmented Mpix/sec, MinGraph-UNet train step (fwd + CE + bwd + grad all-reduce + Adam), 512x512",
                "value": round(world * B * H * W * a.steps / dt / 1e6, 3), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic (N(0,1) images, random labels, formula weights)",
                "config": {"workload": f"BASELINE configs[4]: train step, {B} images/GPU x 3x{H}x{W} fp32, UNet(3,2,32,4), "
                                       f"CrossEntropy + Adam(1e-3, wd 1e-4), per-shard BatchNorm (DDP semantics)",
                           "images_per_gpu": B, "global_batch": B * world,
                           "parallelism": (f"dp{world}: mean all-reduce of the flat {tr.flat.numel() * 4 / 1e6:.1f} MB fp32 gradient on "
                                           f"libmgunet's own RCCL communicator, in >= 4 MB buckets overlapped with backward"
                                           if tr._rccl else f"dp{world}: torch.distributed all-reduce of the flat gradient after backward "
                                                            f"({dist.get_backend() if dist is not None else 'single process'})")},
                "final_loss": round(float(loss), 6),
                "approx_tflops": round(flops * a.steps / dt / 1e12, 2), "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def self_launch(n):
    """Run this script with n ranks of ONE node as child processes (never exec: see the GPU-box rules) and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: 8 = BASELINE configs[1]/[2]; train mode: 4 = configs[4])")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--workload", choices=["c2", "c4"], default="c2",
                    help="c2 (default): BASELINE configs[1]/[2], batch 8 x 3x512x512 + the 16-pixel patch graph; c4: BASELINE configs[3], "
                         "batch 32 x 3x1024x1024 U-Net forward + the GAT(64->64, 4 heads) on 32 synthetic superpixel graphs of 2048 nodes "
                         "with in-degree 8 (SURVEY 8d C4) -- its own line, never the headline value")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: BASELINE configs[1]/[2] full forward (the headline); train: configs[4] train step "
                         "(fwd + CE + bwd + RCCL grad all-reduce + Adam), 4 images per GPU unless --batch is given")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: fp32 results (BASELINE configs[1], the headline); bf16: bf16 storage + fp32 accumulate "
                         "(configs[2]'s precision; reported as its own config, never as the fp32 number)")
    ap.add_argument("--comm", choices=["auto", "rccl"], default=None,
                    help="train mode, gradient exchange: auto = torch.distributed all-reduce after backward (default for N > 1); "
                         "rccl = libmgunet's bucketed RCCL exchange overlapped with backward (default for N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="length of the one long window reported as `sustained` (0 = off); `value` stays the K timed steps")
    ap.add_argument("--spread-windows", type=int, default=4, help="extra timed windows of --steps steps for the min/median/max record")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this parent has not touched the GPU (nothing above initialises HIP); it starts the N
        # ranks as fresh children under torch.distributed.run and relays their output -- rank 0 prints the one JSON line
        return self_launch(a.gpus)
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} does not match WORLD_SIZE={world}")
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
    from mgunet.gat import _context as gat_context

    if a.mode == "train":
        return bench_train(a, world, rank, local_rank, dev, dist)
    c4 = a.workload == "c4"
    if c4:
        a.size = 1024
    B, H, W = a.batch if a.batch is not None else (32 if c4 else 8), a.size, a.size
    unet = mgunet.UNet(3, 2, 32, 4, compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    unet.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))     # random-init-scale formula weights
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    x = torch.randn((B, 3, H, W), device=dev, generator=gen)           # synthetic batch, resident in HBM
    if not c4:
        gat = mgunet.GATNetwork(32, 128, 64, 4, 1)
        gat.load_state_dict(O.make_gat_params(32, 128, 64, 4, 1, seed=0))
        model = mgunet.MinGraphUNet(unet.to(dev).eval(), gat.to(dev).eval(), 16).eval()
    else:
        # per image a synthetic superpixel-like graph, 2048 nodes, 16384 directed edges, in-degree exactly 8,
        # target-sorted; X (2048, 64); GATNetwork(64, 128, 64, 4, 1) -- Fin 64 <= F' 64: the aggregate-first schedule (gat_stmax +
        # gat_fused2), like the patch GAT
        from mgunet.engine import gat_forward_csr
        gat = mgunet.GATNetwork(64, 128, 64, 4, 1)
        gat.load_state_dict(O.make_gat_params(64, 128, 64, 4, 1, seed=0))
        gat = gat.to(dev).eval()
        unet = unet.to(dev).eval()
        N1, deg = 2048, 8
        src = torch.randint(0, N1, (B, N1 * deg), device=dev, generator=gen) + (torch.arange(B, device=dev) * N1)[:, None]
        g_col = src.reshape(-1).to(torch.int32)
        g_rowptr = (torch.arange(B * N1 + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32)
        g_ptr = (torch.arange(B + 1, device=dev, dtype=torch.int64) * N1).to(torch.int32)
        gX = torch.randn((B * N1, 64), device=dev, generator=gen)

        def model(xx):
            lg, sk, ft = unet(xx)
            return lg, sk, ft, gat_forward_csr(gat, gX, g_rowptr, g_col, g_ptr)

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
        # spread: further windows of the same K steps (local clock; the headline `value` stays the first, barrier-bracketed window)
        windows = [dt / a.steps * 1e3]
        for _ in range(max(0, a.spread_windows)):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(a.steps):
                out = model(x)
            torch.cuda.synchronize(dev)
            windows.append((time.perf_counter() - t1) / a.steps * 1e3)
        # sustained: ONE window of >= --sustained-seconds of back-to-back steps (the clock the chip holds under seconds of load)
        sustained = None
        if a.sustained_seconds > 0:
            nsus = max(a.steps, int(1.08 * a.sustained_seconds * 1e3 / max(min(windows), 1e-3)) + 1)   # >= the asked-for seconds
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(nsus):
                out = model(x)
            torch.cuda.synchronize(dev)
            sdt = time.perf_counter() - t1
            sustained = {"steps": nsus, "seconds": round(sdt, 3), "ms_per_step": round(sdt / nsus * 1e3, 4),
                         "mpix_per_s_per_gpu": round(B * H * W * nsus / sdt / 1e6, 2)}
    assert bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[3]).all())
    dt = max_over_ranks(dt, dist, dev)
    ms_step = dt / a.steps * 1e3
    mpix = world * B * H * W * a.steps / dt / 1e6

    roof = gatrec = timing = None
    if rank == 0 and not a.no_profile_pass:
        uctx = next(iter(unet._ctx.values()))
        gctx = gat_context(dev)
        L = _lib.lib()
        nprof = min(a.steps, 10)
        with torch.no_grad():
            L.mgu_profile_enable(uctx.handle, 1)
            L.mgu_profile_enable(gctx.handle, 1)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(nprof):
                model(x)
            torch.cuda.synchronize(dev)
            inst_ms = (time.perf_counter() - t1) / nprof * 1e3
            ustats, gstats = _lib.read_kernel_stats(uctx), _lib.read_kernel_stats(gctx)
            L.mgu_profile_enable(uctx.handle, 0)
            L.mgu_profile_enable(gctx.handle, 0)
        stats = merge_stats(ustats, gstats)
        dom = next((k for k in stats if k["pipe"] >= 0 and k["flops_mfma"] > 0), None)
        # the label describes the DOMINANT record itself: an fp32 result computed on the bf16 pipe is the three-piece split
        arithmetic = ("bf16 storage, fp32 accumulate" if a.dtype == "bf16" else
                      (ARITH_3PIECE if dom is not None and dom["pipe"] == 1 else "exact fp32 MFMA operands"))
        traffic = load_traffic(dom["name"], a.dtype) if dom and (B, H, W) == (8, 512, 512) and not c4 else None
        roof = roofline_from(stats, nprof, arithmetic, traffic)
        ksum = sum(k["ms"] for k in stats) / nprof
        timing = {"ms_per_step": round(ms_step, 4), "instrumented_ms_per_step": round(inst_ms, 4),
                  "instrumented_kernel_ms_per_step": round(ksum, 4),
                  "kernel_ms_per_step_scaled": round(ksum * min(1.0, ms_step / inst_ms), 4),
                  "note": "event records between launches stretch the instrumented pass; kernel times scale with it (rocprofv3 "
                          "--kernel-trace summaries of the same command are under profiles/)"}
        gk = {k["name"]: round(k["ms"] * 1e3 / nprof, 2) for k in gstats}
        gatrec = {"headline_batch": {"graphs": B, "launches_per_step": round(sum(k["launches"] for k in gstats) / nprof, 2), "kernel_us": gk}}
        try:
            gatrec.update(gat_record(dev, L, _lib))
        except Exception as e:   # the record is diagnostics: never lose the headline line over it
            gatrec["error"] = repr(e)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(2, H, W, 14) if not c4 else cpu_baseline(1, H, W, 4)   # ~12 s of host work on the GPU box's 16-core share

    # The two other single-GPU configurations of BASELINE.json, measured in the same run so that the driver's record carries them too
    # (each also has its own full line: --dtype bf16 / --mode train).  Diagnostics: never lose the headline line over them.
    other = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not c4 and a.dtype == "f32" and (B, H, W) == (8, 512, 512):
        other = {}
        try:
            del out
            ub = mgunet.UNet(3, 2, 32, 4, compute_dtype=torch.bfloat16)
            ub.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
            mb = mgunet.MinGraphUNet(ub.to(dev).eval(), gat, 16).eval()
            with torch.no_grad():
                for _ in range(5):
                    mb(x)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(40):
                    mb(x)
                torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t1) / 40 * 1e3
            rec = {"ms_per_step": round(ms, 4), "mpix_per_s": round(B * H * W / ms / 1e3, 2), "steps": 40, "own_line": "bench.py --dtype bf16"}
            try:   # the dominant kernel's roofline of THIS configuration (instrumented pass of its own, after the timed one)
                bctx = next(iter(ub._ctx.values()))
                L = _lib.lib()
                L.mgu_profile_enable(bctx.handle, 1)
                with torch.no_grad():
                    for _ in range(10):
                        mb(x)
                torch.cuda.synchronize(dev)
                bstats = merge_stats(_lib.read_kernel_stats(bctx))
                L.mgu_profile_enable(bctx.handle, 0)
                r = roofline_from(bstats, 10, "bf16 storage, fp32 accumulate", load_traffic(bstats[0]["name"], "bf16") if bstats else None)
                rec["roofline"] = {k: r[k] for k in ("bound", "limiter", "kernel", "launches_per_step", "avg_launch_us", "achieved", "peak", "unit",
                                                      "frac", "algorithmic_tflops", "algorithmic_frac", "traffic")} if r else None
            except Exception as e:
                rec["roofline_error"] = repr(e)
            other["configs[2] precision (bf16 storage, fp32 accumulate), 8 x 3x512x512 full forward"] = rec
            del mb, ub
        except Exception as e:
            other["bf16_error"] = repr(e)
        try:
            ut = mgunet.UNet(3, 2, 32, 4)
            ut.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
            tr = mgunet.Trainer(ut.to(dev), lr=1e-3, weight_decay=1e-4, comm=None)
            xt = x[:4].contiguous()
            yt = torch.randint(0, 2, (4, H, W), device=dev, generator=gen)
            for _ in range(3):
                tr.train_step(xt, yt)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(10):
                tr.train_step(xt, yt)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t1) / 10 * 1e3
            tr.check()
            rec = {"ms_per_step": round(ms, 4), "mpix_per_s": round(4 * H * W / ms / 1e3, 2), "steps": 10, "own_line": "bench.py --mode train"}
            try:
                tctx = next(iter(ut._ctx.values()))
                L = _lib.lib()
                L.mgu_profile_enable(tctx.handle, 1)
                for _ in range(5):
                    tr.train_step(xt, yt)
                torch.cuda.synchronize(dev)
                tstats = merge_stats(_lib.read_kernel_stats(tctx))
                L.mgu_profile_enable(tctx.handle, 0)
                r = roofline_from(tstats, 5, ARITH_3PIECE, None)
                rec["roofline"] = {k: r[k] for k in ("bound", "limiter", "kernel", "launches_per_step", "avg_launch_us", "achieved", "peak", "unit",
                                                      "frac", "algorithmic_tflops", "algorithmic_frac")} if r else None
            except Exception as e:
                rec["roofline_error"] = repr(e)
            other["configs[4] shard (4 images/GPU train step: fwd + CE + bwd + Adam, no exchange at 1 GPU)"] = rec
        except Exception as e:
            other["train_error"] = repr(e)

    if rank == 0:
        line = {"metric": ("segmented Mpix/sec, full MinGraph-UNet forward (U-Net + patch-graph GAT), 512x512 batch" if not c4 else
                           "segmented Mpix/sec, MinGraph-UNet forward (U-Net + stress-graph GAT), 1024x1024 batch"),
                "value": round(mpix, 3), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": a.dtype, "data": "synthetic (N(0,1) images resident in HBM, formula random-init-scale weights)",
                "config": {"workload": (f"BASELINE configs[3]: batch {B}/GPU x 3x{H}x{W} {a.dtype} U-Net forward + GAT(64->64, 4 heads) on {B} "
                                        f"synthetic graphs of 2048 nodes, in-degree 8 (UNet(3,2,32,4)), eval" if c4 else
                                        (f"BASELINE configs[1]: batch {B}/GPU x 3x{H}x{W} fp32 full forward "
                                         if a.dtype == "f32" else
                                         f"BASELINE configs[2] precision: batch {B}/GPU x 3x{H}x{W}, bf16 storage + fp32 accumulate, full forward ")
                                        + "(UNet(3,2,32,4) + patch16 graph GAT(32->64,4 heads)), eval"),
                           "images_per_gpu": B, "global_batch": B * world, "parallelism": f"batch-shard x{world}, no collective"},
                "gpu_event_ms_per_step": round(ev0.elapsed_time(ev1) / a.steps, 4),
                "spread": {"windows": len(windows), "steps_per_window": a.steps, "min_ms": round(min(windows), 4),
                           "median_ms": round(statistics.median(windows), 4), "max_ms": round(max(windows), 4)},
                "sustained": sustained, "timing": timing, "roofline": roof, "gat": gatrec, "cpu_baseline": cpu, "other_configs": other}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
