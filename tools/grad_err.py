"""Per-parameter gradient error of the HIP train step against the fp64 oracle (diagnosis)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "mingraph-unet_amd"), os.path.join(ROOT, "tests")]
import torch, mgunet_oracle as O, mgunet
from test_gpu_train import build
cuda = torch.device("cuda:0")
for cfg, shape in [((3, 3, 8, 2), (2, 3, 37, 45)), ((1, 2, 8, 2), (1, 1, 32, 32)), ((3, 2, 16, 3), (2, 3, 48, 40)), ((3, 2, 16, 3), (2, 3, 64, 64))]:
    p = O.make_unet_params(*cfg, seed=21)
    x = torch.from_numpy(O.formula_normal("train/x", shape, seed=21))
    y = torch.from_numpy(O.formula_labels("train/y", (shape[0], shape[2], shape[3]), cfg[1], seed=22))
    l, g, *_ = O.train_step(p, x, y, cfg[3])
    p64 = type(p)((k, v.double() if v.dtype.is_floating_point else v) for k, v in p.items())
    l64, g64, *_ = O.train_step(p64, x.double(), y, cfg[3])
    model = build(cfg, 21, cuda)
    tr = mgunet.Trainer(model, lr=1e-3, weight_decay=1e-4)
    loss = tr.forward_backward(x.to(cuda), y.to(cuda))
    sd = dict(model.named_parameters())
    print("==", cfg, shape, "loss", float(loss), float(l64))
    for k in g:
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            continue
        gg = sd[k].grad.detach().cpu().double()
        rn = float(g64[k].norm())
        print(f"  {k:55s} ours {float((gg - g64[k]).norm())/rn:9.2e}  ref32 {float((g[k].double() - g64[k]).norm())/rn:9.2e}")
