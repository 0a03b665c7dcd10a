"""Both GAT layer schedules through the prepared-weights entry points (mgu_gat_prepare / mgu_gat_layer_forward_prepared)
against the oracle's per-head forward (model/gat/graph_attention.py:40-118, 150-160):
  aggregate-first (Fin <= F': gat_stmax_kernel + gat_fused2_kernel, 2 launches) and the Wh-row gather (GEMM + gat_edge_max +
  gat_aggregate, forced with MGU_NO_GAT_FUSED=1 in a context of its own), on block-diagonal patch graphs (the fast path of the
  gather kernel: every row <= 4 in-edges), a ragged random graph (rows with 0..11 in-edges, isolated nodes, a tail that is not
  a multiple of the kernels' row groups) and repeated calls (the two alternating per-graph max arrays)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O
from mgunet import _lib
from mgunet.gat import coo_to_csr_device

pytestmark = pytest.mark.gpu


def make_ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _lib.Context(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def oracle_layer(X, ei_list, W, a, heads, Fh, concat, alpha=0.2):
    outs = []
    for ei, lo, hi in ei_list:          # one graph at a time: the max of e is per graph (graph_attention.py:86)
        hs = [O.gat_head_forward(X[lo:hi], ei, W[h * Fh:(h + 1) * Fh], a[h:h + 1], alpha) for h in range(heads)]
        outs.append(torch.cat(hs, 1) if concat else torch.stack(hs).mean(0))
    return torch.cat(outs, 0)


def run(ctx, cuda, X, rowptr, col, gp, W, a, heads, Fh, concat):
    L = _lib.lib()
    N, Fin = X.shape
    E = col.numel()
    h = C.c_void_p()
    s = _lib.current_stream_ptr(cuda)
    Wd, ad = W.contiguous().to(cuda), a.contiguous().to(cuda)
    _lib.check(L.mgu_gat_prepare(ctx.handle, Wd.data_ptr(), ad.data_ptr(), heads, Fh, Fin, 1 if E else 0, C.byref(h), s), ctx.handle)
    out = torch.full((N, heads * Fh if concat else Fh), float("nan"), device=cuda)
    res = []
    for _ in range(3):                  # consecutive calls alternate between the two max-accumulator arrays
        _lib.check(L.mgu_gat_layer_forward_prepared(ctx.handle, h, X.data_ptr(), N, rowptr.data_ptr(), col.data_ptr() if E else None, E,
                                                    gp.data_ptr() if gp is not None else None, gp.numel() - 1 if gp is not None else 1,
                                                    1 if concat else 0, 0.2, out.data_ptr(), s), ctx.handle)
        res.append(out.clone())
    torch.cuda.synchronize()
    L.mgu_gat_release(ctx.handle, h)
    assert torch.equal(res[0], res[1]) and torch.equal(res[1], res[2])
    return res[0].cpu()


@pytest.mark.parametrize("sched", ["aggregate_first", "wh_row_gather"])
@pytest.mark.parametrize("concat", [0, 1])
@pytest.mark.parametrize("Fin,heads,Fh", [(32, 4, 64), (64, 4, 64), (32, 2, 32)])
def test_batched_patch_graphs(cuda, sched, concat, Fin, heads, Fh):
    G, H, Wd = 3, 80, 112                                   # 5 x 7 patch grids
    ctx = make_ctx({"MGU_NO_GAT_FUSED": "1"} if sched == "wh_row_gather" else {})
    pg = mgunet.PatchGraphConstructor(16)
    rowptr, col, gp, N1, E1 = pg.batched_csr(H, Wd, G, cuda)
    ei = torch.from_numpy(O.patch_graph_edges(H, Wd, 16))
    N = N1 * G
    X = torch.from_numpy(O.formula_normal("gs/x", (N, Fin), seed=Fin + heads))
    W = torch.from_numpy(O.formula_uniform("gs/w", (heads * Fh, Fin), -0.4, 0.4, seed=1))
    a = torch.from_numpy(O.formula_uniform("gs/a", (heads, 2 * Fh), -0.4, 0.4, seed=2))
    ref = oracle_layer(X, [(ei, g * N1, (g + 1) * N1) for g in range(G)], W, a, heads, Fh, concat)
    got = run(ctx, cuda, X.to(cuda), rowptr, col, gp, W, a, heads, Fh, concat)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("sched", ["aggregate_first", "wh_row_gather"])
def test_ragged_random_graph_with_isolated_nodes(cuda, sched):
    ctx = make_ctx({"MGU_NO_GAT_FUSED": "1"} if sched == "wh_row_gather" else {})
    rng = np.random.default_rng(5)
    N, Fin, heads, Fh = 203, 32, 4, 64
    deg = rng.integers(0, 12, size=N)
    deg[::9] = 0                                            # isolated targets: rows must be exactly 0
    tgt = np.repeat(np.arange(N), deg)
    src = rng.integers(0, N, size=tgt.size)
    perm = rng.permutation(tgt.size)                        # COO in arbitrary order
    ei = torch.from_numpy(np.stack([src[perm], tgt[perm]]).astype(np.int64))
    rowptr, col = coo_to_csr_device(ei.to(cuda), N)
    X = torch.from_numpy(O.formula_normal("gs/rx", (N, Fin), seed=3)) * 2.0
    W = torch.from_numpy(O.formula_uniform("gs/rw", (heads * Fh, Fin), -0.5, 0.5, seed=4))
    a = torch.from_numpy(O.formula_uniform("gs/ra", (heads, 2 * Fh), -0.5, 0.5, seed=5))
    ref = oracle_layer(X, [(ei, 0, N)], W, a, heads, Fh, 0)
    got = run(ctx, cuda, X.to(cuda), rowptr, col, None, W, a, heads, Fh, 0)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert float(got[::9].abs().max()) == 0.0


@pytest.mark.parametrize("sched", ["aggregate_first", "wh_row_gather"])
@pytest.mark.parametrize("scale", [0.1, 2.0])
def test_segment_degree_patterns(cuda, sched, scale):
    """Four-row CSR segments with every mix of empty, short and full rows (in-degree 0..8, the range of the gather kernel's
    lane-parallel fast path: 4 x 4 and 2 x 8 batches) and a 9-edge row beside them (per-row path).  An empty row in FRONT of a
    short one once broke the fast path: its lanes were switched off around a cross-lane shuffle that read the ids they hold."""
    ctx = make_ctx({"MGU_NO_GAT_FUSED": "1"} if sched == "wh_row_gather" else {})
    Fin, heads, Fh = 32, 4, 64
    W = torch.from_numpy(O.formula_uniform("gs/rw", (heads * Fh, Fin), -0.5, 0.5, seed=4))
    a = torch.from_numpy(O.formula_uniform("gs/ra", (heads, 2 * Fh), -0.5, 0.5, seed=5))
    pats = [[4, 0, 2, 0], [4, 2, 0, 0], [2, 2, 2, 2], [4, 4, 4, 4], [0, 0, 0, 1], [3, 0, 3, 0], [1, 1, 1, 1], [0, 2, 0, 2], [0, 0, 0, 0],
            [8, 8, 8, 8], [5, 0, 8, 1], [0, 7, 0, 6], [8, 0, 0, 0], [9, 1, 1, 1], [4, 4, 4, 5]]
    deg = np.array([d for p in pats for d in p] + [3, 1])          # a ragged last segment
    N = deg.size
    rng = np.random.default_rng(1)
    tgt = np.repeat(np.arange(N), deg)
    src = rng.integers(0, N, size=tgt.size)
    ei = torch.from_numpy(np.stack([src, tgt]).astype(np.int64))
    rowptr, col = coo_to_csr_device(ei.to(cuda), N)
    X = torch.from_numpy(O.formula_normal("gs/px", (N, Fin), seed=3)) * scale
    for concat in (0, 1):
        ref = oracle_layer(X, [(ei, 0, N)], W, a, heads, Fh, concat)
        got = run(ctx, cuda, X.to(cuda), rowptr, col, None, W, a, heads, Fh, concat)
        err = (got - ref).abs().max(1).values
        assert float(err.max()) <= 1e-4 * max(1.0, float(ref.abs().max())), (concat, err.numpy().round(5).tolist())
        assert float(got[torch.from_numpy(deg == 0)].abs().max()) == 0.0


@pytest.mark.parametrize("N,E", [(203, 1500), (1, 0), (50, 0), (5000, 200000), (7, 3)])
def test_device_coo_to_csr_matches_host_routine_bit_exact(cuda, N, E):
    """mgu_coo_to_csr_device (stable radix sort by target) against the host routine mgu_coo_to_csr / the oracle: index maps are
    compared bit for bit; an id outside [0, N) raises IndexError where the reference's indexing would."""
    rng = np.random.default_rng(N + E)
    ei = np.stack([rng.integers(0, N, size=E), rng.integers(0, N, size=E)]).astype(np.int64)
    rowptr, col = coo_to_csr_device(torch.from_numpy(ei).to(cuda), N)
    orp, ocol, _ = O.coo_to_csr(ei, N)
    assert np.array_equal(rowptr.cpu().numpy(), orp) and np.array_equal(col.cpu().numpy(), ocol)
    if E:
        bad = ei.copy()
        bad[1, E // 2] = N
        with pytest.raises(IndexError):
            coo_to_csr_device(torch.from_numpy(bad).to(cuda), N)
        bad[1, E // 2] = -1
        with pytest.raises(IndexError):
            coo_to_csr_device(torch.from_numpy(bad).to(cuda), N)


@pytest.mark.parametrize("tag,H,W,p", [("g128", 128, 128, 32), ("g130", 130, 140, 32), ("g512", 512, 512, 16), ("g1024", 1024, 1024, 16)])
def test_patch_graph_index_maps_bit_exact_vs_reference_fixtures(cuda, golden, tag, H, W, p):
    """The index maps the GPU path really consumes, against the COO fixtures the reference's own PatchGraphConstructor produced
    (tests/golden/patch_graph.npz; preprocessing/graph_construction/patch_graph_construction.py:49-102): mgu_patch_graph_build's
    COO as handed to the device, the device-side stable COO -> CSR of it, and the block-diagonal batched CSR the forward uses --
    all bit-exact (the CPU tier checks the host builder; this one runs on the GPU box)."""
    ref = golden["patch_graph"][tag]
    pgc = mgunet.PatchGraphConstructor(p)
    nph, npw = O.patch_grid(H, W, p)
    N = nph * npw
    _, ei = pgc.construct_patch_graph(torch.zeros(1, H, W), torch.zeros(N, 4, device=cuda))
    assert ei.is_cuda and ei.dtype == torch.int64 and np.array_equal(ei.cpu().numpy(), ref)
    rowptr, col = coo_to_csr_device(ei, N)
    orp, ocol, _ = O.coo_to_csr(ref, N)
    assert np.array_equal(rowptr.cpu().numpy(), orp) and np.array_equal(col.cpu().numpy(), ocol)
    B = 3
    brp, bcol, gp, n1, e1 = pgc.batched_csr(H, W, B, cuda)
    assert (n1, e1) == (N, ref.shape[1]) and gp.cpu().tolist() == [0, N, 2 * N, 3 * N]
    big = np.concatenate([ref + N * b for b in range(B)], axis=1)
    orp, ocol, _ = O.coo_to_csr(big, B * N)
    assert np.array_equal(brp.cpu().numpy(), orp) and np.array_equal(bcol.cpu().numpy(), ocol)
    assert np.array_equal(pgc.edge_index(H, W, cuda, B).cpu().numpy(), big)


@pytest.mark.parametrize("kind", ["patch", "stress"])
def test_persistent_walk_equals_graph_by_graph(cuda, kind):
    """gat_fused2_kernel is persistent and software-pipelined over node tiles (rows / source ids / source rows of later tiles are
    requested while earlier tiles compute).  At BASELINE sizes a workgroup walks several tiles: 72 patch graphs of the 512^2 grid
    (2304 tiles) and configs[3]'s 32 stress graphs (2048 tiles, Fin = 64, in-degree 8).  A graph's rows must be bit-identical
    whether it runs inside the batch or alone (one tile per workgroup: no pipeline), and match the oracle on one graph."""
    ctx = make_ctx({})
    heads, Fh = 4, 64
    if kind == "patch":
        Fin, G = 32, 72
        rowptr, col, gp, N1, E1 = mgunet.PatchGraphConstructor(16).batched_csr(512, 512, G, cuda)
        ei1 = torch.from_numpy(O.patch_graph_edges(512, 512, 16))
    else:
        Fin, G, N1, deg = 64, 32, 2048, 8
        rng = np.random.default_rng(3)
        src = rng.integers(0, N1, size=(G, N1 * deg))
        col = torch.from_numpy((src + (np.arange(G) * N1)[:, None]).reshape(-1).astype(np.int32)).to(cuda)
        rowptr = torch.from_numpy((np.arange(G * N1 + 1) * deg).astype(np.int32)).to(cuda)
        gp = torch.from_numpy((np.arange(G + 1) * N1).astype(np.int32)).to(cuda)
        E1 = N1 * deg
        ei1 = torch.from_numpy(np.stack([src[G - 1], np.repeat(np.arange(N1), deg)]).astype(np.int64))
    N = N1 * G
    X = torch.from_numpy(O.formula_normal("gs/big", (N, Fin), seed=11))
    W = torch.from_numpy(O.formula_uniform("gs/w", (heads * Fh, Fin), -0.4, 0.4, seed=1))
    a = torch.from_numpy(O.formula_uniform("gs/a", (heads, 2 * Fh), -0.4, 0.4, seed=2))
    Xd = X.to(cuda)
    full = run(ctx, cuda, Xd, rowptr, col, gp, W, a, heads, Fh, 0)
    for g in (0, G // 2, G - 1):
        rp1 = (rowptr[g * N1:(g + 1) * N1 + 1] - rowptr[g * N1]).contiguous()
        c1 = (col[g * E1:(g + 1) * E1] - g * N1).contiguous()
        one = run(ctx, cuda, Xd[g * N1:(g + 1) * N1].contiguous(), rp1, c1, None, W, a, heads, Fh, 0)
        assert torch.equal(one, full[g * N1:(g + 1) * N1]), g
    ref = oracle_layer(X[(G - 1) * N1:], [(ei1, 0, N1)], W, a, heads, Fh, 0)
    assert float((full[(G - 1) * N1:] - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("kind", ["patch", "stress"])
def test_repeated_calls_are_bitwise_identical(cuda, kind):
    """40 consecutive calls of the aggregate-first layer on BASELINE-size batches (64 patch graphs; configs[3]'s 32 stress graphs)
    give the same bytes: no unordered sum, no timing-dependent hand-off (an earlier barrier form produced single wrong rows in
    about every fourth call here)."""
    ctx = make_ctx({})
    heads, Fh = 4, 64
    if kind == "patch":
        Fin, G = 32, 64
        rowptr, col, gp, N1, E1 = mgunet.PatchGraphConstructor(16).batched_csr(512, 512, G, cuda)
    else:
        Fin, G, N1, deg = 64, 32, 2048, 8
        rng = np.random.default_rng(3)
        src = rng.integers(0, N1, size=(G, N1 * deg))
        col = torch.from_numpy((src + (np.arange(G) * N1)[:, None]).reshape(-1).astype(np.int32)).to(cuda)
        rowptr = torch.from_numpy((np.arange(G * N1 + 1) * deg).astype(np.int32)).to(cuda)
        gp = torch.from_numpy((np.arange(G + 1) * N1).astype(np.int32)).to(cuda)
    N = N1 * G
    X = torch.from_numpy(O.formula_normal("gs/rep", (N, Fin), seed=13)).to(cuda)
    W = torch.from_numpy(O.formula_uniform("gs/w", (heads * Fh, Fin), -0.4, 0.4, seed=1)).to(cuda)
    a = torch.from_numpy(O.formula_uniform("gs/a", (heads, 2 * Fh), -0.4, 0.4, seed=2)).to(cuda)
    L = _lib.lib()
    h = C.c_void_p()
    s = _lib.current_stream_ptr(cuda)
    _lib.check(L.mgu_gat_prepare(ctx.handle, W.data_ptr(), a.data_ptr(), heads, Fh, Fin, 1, C.byref(h), s), ctx.handle)
    first = None
    for i in range(40):
        out = torch.full((N, Fh), float("nan"), device=cuda)
        _lib.check(L.mgu_gat_layer_forward_prepared(ctx.handle, h, X.data_ptr(), N, rowptr.data_ptr(), col.data_ptr(), col.numel(), gp.data_ptr(), G,
                                                    0, 0.2, out.data_ptr(), s), ctx.handle)
        if first is None:
            first = out
        else:
            bad = (out != first).any(1).nonzero().flatten()
            assert bad.numel() == 0, (i, bad[:8].tolist())
    torch.cuda.synchronize()
    L.mgu_gat_release(ctx.handle, h)
