#!/usr/bin/env python3
"""Region stage + fusion micro-benchmark: B images 512x512, patch 16 -> 1024 patches, K segments,
64-feature patch embeddings, 32-channel U-Net feature.  Prints wall time per call of the whole stage and of the fuse
kernel alone with its algorithmic bytes (read F_u + write the fused NHWC tensor) and the oracle's CPU time on one image.
Run under rocprofv3 for per-kernel durations."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import mgunet, mgunet_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--segments", type=int, default=2)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, K, H, W, D, Cu = a.batch, a.segments, 512, 512, 64, 32
nph = npw = 32
p = O.make_gat_params(D, 128, D, 4, 1, seed=9)
gat = mgunet.GATNetwork(D, 128, D, 4, num_gat_layers=1)
gat.load_state_dict(p)
gat = gat.to(dev).eval()
feats = torch.randn((B * nph * npw, D), device=dev) * 0.5
hard = torch.randint(0, K, (B * nph * npw,), device=dev)
fu = torch.randn((B, H, W, Cu), device=dev).permute(0, 3, 1, 2)


def timed(fn):
    for _ in range(10):   # the first calls pay for the caching allocator's 805 MB output block
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters


t_stage = timed(lambda: mgunet.region_stage(feats, hard, B, K, gat, nph, npw, H, W, f_u=fu))
emb = mgunet.region_stage(feats, hard, B, K, gat, nph, npw, H, W, f_u=fu)[0]
t_fuse = timed(lambda: mgunet.region_fuse(fu, emb, hard, B, H, W, nph, npw, K))
alg = B * H * W * (Cu + (Cu + D)) * 4
torch.set_num_threads(min(16, os.cpu_count() or 1))
fc, hc, fuc = feats[: nph * npw].cpu(), hard[: nph * npw].cpu(), fu[:1].cpu().contiguous()
t0 = time.perf_counter()
for _ in range(3):
    _, pix = O.region_stage(fc, hc, K, p, 4, nph, npw, H, W)
    O.feature_fusion([fuc], pix.unsqueeze(0))
t_cpu = (time.perf_counter() - t0) / 3
print(f"batch={B} K={K} stage_wall_us={t_stage*1e6:.1f} fuse_wall_us={t_fuse*1e6:.1f} fuse_algorithmic_MB={alg/1e6:.1f} "
      f"fuse_TBps={alg/t_fuse/1e12:.2f} stage_Mpix_per_s={B*H*W/t_stage/1e6:.0f} cpu_oracle_ms_per_image={t_cpu*1e3:.1f} "
      f"cpu_Mpix_per_s={H*W/t_cpu/1e6:.1f} cores={torch.get_num_threads()}")
