"""Debug: whole-UNet forward of a 64-image batch vs its 8-image shard, repeated, per kernel-selection setting."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import mgunet
import mgunet_oracle as O
cuda = torch.device("cuda:0")
cfg = (3, 2, 32, 4)
p = O.make_unet_params(*cfg, seed=0)
gen = torch.Generator(device=cuda); gen.manual_seed(7)
xb = torch.randn((64, 3, 512, 512), device=cuda, generator=gen)
for name, env in (("all asm", {}), ("wide only", {"MGU_WINO_ASM_NARROW": "0"}), ("no asm", {"MGU_WINO_ASM": "0"})):
    for k in ("MGU_WINO_ASM", "MGU_WINO_ASM_NARROW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    unet = mgunet.UNet(*cfg); unet.load_state_dict(p); unet = unet.to(cuda).eval()
    for rep in range(4):
        outs = unet(xb)
        sh = unet(xb[24:32])
        names = ["logits"] + [f"skip{i}" for i in range(len(outs[1]))] + [f"feat{i}" for i in range(len(outs[2]))]
        full = [outs[0]] + list(outs[1]) + list(outs[2])
        part = [sh[0]] + list(sh[1]) + list(sh[2])
        bad = []
        for n, a, b in zip(names, full, part):
            a = a[24:32]
            if not torch.equal(a, b):
                d = (a != b)
                idx = d.nonzero()[0].tolist()
                bad.append(f"{n}: {int(d.sum())} differ, first {idx}, max {float((a - b).abs().max()):.2e}")
        print(name, "rep", rep, "OK" if not bad else bad)
    del unet
