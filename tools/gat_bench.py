#!/usr/bin/env python3
"""GAT message-passing micro-benchmark: block-diagonal batch of B patch graphs (512x512 / patch 16 ->
1024 nodes, 3968 edges each) or the C4 stress graph (2048 nodes, in-degree 8), GAT(Fin -> 64, 4 heads).
Run under rocprofv3 (tools/gpu_gat.sh) to get per-kernel durations and HBM bytes."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mgunet, mgunet_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=64)
ap.add_argument("--kind", choices=["patch", "stress"], default="patch")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.kind == "patch":
    fin, pgc = 32, mgunet.PatchGraphConstructor(16)
    rowptr, col, gp, N, E = pgc.batched_csr(512, 512, a.graphs, dev)
else:
    fin, N, deg = 64, 2048, 8
    u = O.formula_uniform("c4/src", (N * deg,), 0.0, 1.0, 3).astype(np.float64)
    src = np.minimum((u * N).astype(np.int64), N - 1)
    E = N * deg
    rp = np.arange(N * a.graphs + 1, dtype=np.int64) * deg
    cl = np.concatenate([src + b * N for b in range(a.graphs)])
    rowptr, col = torch.from_numpy(rp.astype(np.int32)).to(dev), torch.from_numpy(cl.astype(np.int32)).to(dev)
    gp = torch.from_numpy((np.arange(a.graphs + 1) * N).astype(np.int32)).to(dev)
gat = mgunet.GATNetwork(fin, 128, 64, 4, 1)
gat.load_state_dict(O.make_gat_params(fin, 128, 64, 4, 1, seed=0))
gat = gat.to(dev).eval()
X = torch.randn(N * a.graphs, fin, device=dev)
for _ in range(3):
    y = mgunet.gat_forward_csr(gat, X, rowptr, col, gp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    y = mgunet.gat_forward_csr(gat, X, rowptr, col, gp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
nodes, edges, HF, Fo = N * a.graphs, E * a.graphs, 256, 64
comp = nodes * HF * 4 + (nodes + 1 + edges) * 4 + 2 * nodes * 4 * 4 + nodes * Fo * 4
logical = edges * (4 + HF * 4 + 4)
print(f"graphs={a.graphs} kind={a.kind} nodes={nodes} edges={edges} layer_wall_us={dt*1e6:.1f} "
      f"aggregate_compulsory_MB={comp/1e6:.2f} logical_gather_MB={logical/1e6:.2f} finite={bool(torch.isfinite(y).all())}")
