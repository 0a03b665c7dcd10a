"""Interleaved A/B of the whole C2 forward step under different library environment settings (the tuning flags are read when a
context is created, i.e. per mgunet.UNet): prints ms per step for every setting and round, and whether the outputs are bitwise equal
to the first setting's.
  python tools/ab_env_step.py ROUNDS NAME=VALUE[,NAME=VALUE...] [NAME=VALUE...] ...     ("-" = no override)
  e.g.  python tools/ab_env_step.py 3 - MGU_WINO_ASM_NARROW=0 MGU_WINO_ASM=0
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mingraph-unet_amd"))
import mgunet  # noqa: E402

rounds = int(sys.argv[1])
settings = sys.argv[2:] or ["-"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(8, 3, 512, 512, device=dev)
touched = sorted({kv.split("=")[0] for s in settings if s != "-" for kv in s.split(",")})
first = None
for r in range(rounds):
    for s in settings:
        for k in touched:
            os.environ.pop(k, None)
        if s != "-":
            for kv in s.split(","):
                k, v = kv.split("=")
                os.environ[k] = v
        torch.manual_seed(1)
        unet = mgunet.UNet(3, 2, 32, 4).to(dev).eval()
        for _ in range(3):
            lg, sk, ft = unet(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lg, sk, ft = unet(x)
        e1.record()
        torch.cuda.synchronize()
        outs = [lg.clone()] + [t.clone() for t in ft]
        if first is None:
            first = outs
        same = all(torch.equal(a, b) for a, b in zip(first, outs))
        print(json.dumps({"round": r, "env": s, "ms_per_step": round(e0.elapsed_time(e1) / 20, 4), "bitwise_equal_to_first": same}), flush=True)
        del unet
