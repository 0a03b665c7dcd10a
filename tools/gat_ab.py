#!/usr/bin/env python3
"""bench.py's `gat` record alone (both schedules at 8 / 64 patch graphs and on the configs[3] stress graphs), one compact line per case:
layer time by events and the per-kernel times of the instrumented pass.  A/B two kernels with the MGU_* switches of the environment
(e.g. MGU_GAT_FUSED_V=1) or two builds (MGU_LIB_PATH)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402  (puts mingraph-unet_amd/ and oracle/ on sys.path)
from mgunet import _lib  # noqa: E402

dev = torch.device("cuda:0")
rec = bench.gat_record(dev, _lib.lib(), _lib)
for key, r in rec.items():
    if not isinstance(r, dict):
        continue
    for sched in ("aggregate_first", "wh_row_gather"):
        s = r[sched]
        print(f"{key:18s} {sched:16s} layer {s['layer_us']:7.2f} us = {s['layer_TBps']} TB/s  kernels {json.dumps(s['kernel_us'])}  dominant {s['achieved_TBps']} TB/s")
