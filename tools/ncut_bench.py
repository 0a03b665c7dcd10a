#!/usr/bin/env python3
"""MinCut stage micro-benchmark: segment predictor (GAT 64 -> 2, 2 heads) + normalized-cut loss on a
block-diagonal batch of B patch graphs (512x512 / patch 16 -> 1024 nodes, 3968 edges each, 64 features per node).
Prints the wall time per call of the whole stage and of the loss alone, the algorithmic bytes of the loss kernel
(rows gathered: E x D x 4 B + node rows N x D x 4 B + assignments + CSR) and the oracle's CPU time on a bounded sample.
Run under rocprofv3 for the per-kernel durations (tools/README.md)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mingraph-unet_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import mgunet, mgunet_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=64)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--cpu-graphs", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
D, K = 64, 2
one = torch.from_numpy(O.patch_graph_edges(512, 512, 16))
ei = torch.cat([one + 1024 * b for b in range(a.graphs)], dim=1).to(dev)
N, E = 1024 * a.graphs, ei.shape[1]
X = torch.randn((N, D), device=dev) * 0.15
p = O.make_segment_predictor_params(D, K, 32, True, 2, seed=7)
pred = mgunet.PatchSegmentPredictor(D, K, hidden_dim=32, use_gnn=True, num_heads=2)
pred.load_state_dict(p)
pred = pred.to(dev).eval()
mc = mgunet.MinCutRefinement()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters


t_stage = timed(lambda: mc(X, ei, K, pred))
soft = mc(X, ei, K, pred)[1]
t_loss = timed(lambda: mc.normalized_cut_loss(X, ei, soft, K))
alg = E * D * 4 + N * D * 4 + (E + N) * K * 4 + (N + 1 + E) * 4
# CPU oracle (reference algorithm) on a bounded sample
g = min(a.cpu_graphs, a.graphs)
eic = torch.cat([one + 1024 * b for b in range(g)], dim=1)
Xc = X[: 1024 * g].cpu()
torch.set_num_threads(min(16, os.cpu_count() or 1))
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    lg = O.segment_predictor_forward(p, Xc, eic, True, 2)
    O.mincut_forward(Xc, eic, K, lg)
t_cpu = (time.perf_counter() - t0) / reps
print(f"graphs={a.graphs} nodes={N} edges={E} stage_wall_us={t_stage*1e6:.1f} loss_wall_us={t_loss*1e6:.1f} "
      f"loss_algorithmic_MB={alg/1e6:.2f} stage_Mnodes_per_s={N/t_stage/1e6:.1f} "
      f"cpu_oracle_ms_per_{g}_graphs={t_cpu*1e3:.1f} cpu_Mnodes_per_s={1024*g/t_cpu/1e6:.3f} cores={torch.get_num_threads()}")
