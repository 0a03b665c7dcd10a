#!/usr/bin/env python3
"""DetectionHead micro-benchmark: fused features (B, 96, 512, 512) -> boxes + confidence.
Prints wall time per call, the algorithmic FLOPs of its two 3x3 convolutions and the oracle's CPU time on one image.
Run under rocprofv3 for per-kernel durations."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import mgunet, mgunet_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, C, H, W = a.batch, 96, 512, 512
p = O.make_detection_head_params(C, 1, 256, False, seed=13)
m = mgunet.DetectionHead(C, 1)
sd = dict(p)
sd["conv_block.2.num_batches_tracked"] = sd["conv_block.5.num_batches_tracked"] = torch.tensor(0)
m.load_state_dict(sd)
m = m.to(dev).eval()
x = torch.randn((B, H, W, C), device=dev).permute(0, 3, 1, 2)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
flops = 2.0 * B * H * W * 9 * (96 * 48 + 48 * 24)
torch.set_num_threads(min(16, os.cpu_count() or 1))
xc = x[:1].cpu().contiguous()
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(2):
        O.detection_head_forward(p, xc, 1)
tc = (time.perf_counter() - t0) / 2
print(f"batch={B} head_wall_ms={dt*1e3:.3f} conv_GFLOP={flops/1e9:.1f} algorithmic_TFLOPs={flops/dt/1e12:.1f} Mpix_per_s={B*H*W/dt/1e6:.0f} "
      f"cpu_oracle_ms_per_image={tc*1e3:.0f} cpu_Mpix_per_s={H*W/tc/1e6:.2f} cores={torch.get_num_threads()}")
