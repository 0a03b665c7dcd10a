#!/usr/bin/env python3
"""Host time to ENQUEUE one forward / one train step (no synchronisation inside the timed loop) against the GPU time of the step: the
reason the library does not capture its launch sequences in hipGraphs -- the stream never runs dry."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mgunet  # noqa: E402
import mgunet_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
model = mgunet.UNet(3, 2, 32, 4)
model.load_state_dict(O.make_unet_params(3, 2, 32, 4, seed=0))
model = model.to(dev).eval()
for B in (8, 1):
    x = torch.from_numpy(O.formula_normal("bench/x", (B, 3, 512, 512), seed=1)).to(dev)
    with torch.no_grad():
        for _ in range(5):
            model(x)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            model(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"forward B={B}: host enqueue {1e6 * (t1 - t0) / n:7.1f} us per step, GPU {1e6 * (t2 - t0) / n:7.1f} us per step (23 launches)")
tr = mgunet.Trainer(mgunet.UNet(3, 2, 32, 4).to(dev), lr=1e-3, weight_decay=1e-4)
x = torch.from_numpy(O.formula_normal("bench/x", (4, 3, 512, 512), seed=1)).to(dev)
y = torch.randint(0, 2, (4, 512, 512), device=dev)
for _ in range(3):
    tr.train_step(x, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"train step B=4: host enqueue {1e6 * (t1 - t0) / n:7.1f} us per step, GPU {1e6 * (t2 - t0) / n:7.1f} us per step (~250 launches)")
