#!/usr/bin/env python3
"""Stages 1-7 of the reference's end-to-end forward (mgunet.MinGraphUNetE2E: U-Net, patch GAT, segment predictor +
normalized-cut loss, region stage, fusion, detection head) on B x 3 x 512 x 512: wall time per call and the share of the
U-Net + patch-GAT part (the headline metric of bench.py)."""
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
B, H, W, K = a.batch, 512, 512, 2
unet = mgunet.UNet(3, 2, 32, 4); unet.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=1))
pgat = mgunet.GATNetwork(32, 128, 64, 4, 1); pgat.load_state_dict(O.make_gat_params(32, 128, 64, 4, 1, seed=2))
pred = mgunet.PatchSegmentPredictor(64, K, hidden_dim=32, use_gnn=True, num_heads=2)
pred.load_state_dict(O.make_segment_predictor_params(64, K, 32, True, 2, seed=3))
rgat = mgunet.GATNetwork(64, 128, 64, 4, 1); rgat.load_state_dict(O.make_gat_params(64, 128, 64, 4, 1, seed=4))
det = mgunet.DetectionHead(96, 1)
sd = dict(O.make_detection_head_params(96, 1, 256, False, seed=5))
sd["conv_block.2.num_batches_tracked"] = sd["conv_block.5.num_batches_tracked"] = torch.tensor(0)
det.load_state_dict(sd)
model = mgunet.MinGraphUNetE2E(unet, pgat, pred, mgunet.MinCutRefinement(), rgat, det, num_segments=K).to(dev).eval()
x = torch.randn((B, 3, H, W), device=dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters


t_all = timed(lambda: model(x))
t_core = timed(lambda: model.core(x))
out = model(x)
print(f"batch={B} e2e_ms={t_all*1e3:.3f} unet_plus_patch_gat_ms={t_core*1e3:.3f} later_stages_ms={(t_all-t_core)*1e3:.3f} "
      f"e2e_Mpix_per_s={B*H*W/t_all/1e6:.0f} loss_partition={float(out['loss_partition']):.5f} finite={bool(torch.isfinite(out['bboxes']).all())}")
