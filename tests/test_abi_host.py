"""CPU-side checks: the C-ABI library loads and exports every symbol include/mgunet.h declares, the
host-side index routines are bit-exact, and the Python mirror keeps the reference's API surface.
No compute call is made here (no GPU in this tier)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import mgunet
import mgunet_oracle as O
from mgunet import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mgunet.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mgu_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 15
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), f"libmgunet.so does not export {s}"
        assert s in _lib._PROTOS, f"ctypes prototype missing for {s}"
    assert sorted(_lib._PROTOS) == syms


def test_create_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = _lib.lib().mgu_create(0, C.byref(h))
    assert rc == _lib.MGU_ERR_HIP
    assert b"no CPU fallback" in _lib.lib().mgu_last_error(None)
    with pytest.raises(RuntimeError):
        _lib.Context(0)


@pytest.mark.parametrize("tag,H,W,p", [("g128", 128, 128, 32), ("g130", 130, 140, 32), ("g512", 512, 512, 16),
                                        ("g1024", 1024, 1024, 16), ("g1", 16, 16, 16), ("grow", 16, 80, 16)])
def test_patch_graph_build_bit_exact(golden, tag, H, W, p):
    pgc = mgunet.PatchGraphConstructor(p)
    nph, npw = O.patch_grid(H, W, p)
    feats = torch.zeros(nph * npw, 3)
    f2, ei = pgc.construct_patch_graph(torch.zeros(1, H, W), feats)
    assert f2 is feats
    assert ei.dtype == torch.int64 and tuple(ei.shape) == golden["patch_graph"][tag].shape
    assert np.array_equal(ei.numpy(), golden["patch_graph"][tag])
    coo, rowptr, col, a, b = pgc._maps(H, W)
    orp, ocol, _ = O.coo_to_csr(O.patch_graph_edges(H, W, p), nph * npw)
    assert (a, b) == (nph, npw) and np.array_equal(rowptr, orp) and np.array_equal(col, ocol)


def test_patch_graph_feature_count_mismatch_raises_valueerror():
    with pytest.raises(ValueError, match="does not match expected number of patches"):
        mgunet.PatchGraphConstructor(32).construct_patch_graph(torch.zeros(3, 128, 128), torch.zeros(15, 4))


def test_coo_to_csr_host_random_graph():
    rng = np.random.default_rng(0)
    N, E = 97, 1000
    coo = rng.integers(0, N, size=(2, E)).astype(np.int64)
    rp, cl = np.empty(N + 1, np.int32), np.empty(E, np.int32)
    rc = _lib.lib().mgu_coo_to_csr(coo.ctypes.data_as(C.c_void_p), E, N, rp.ctypes.data_as(C.c_void_p),
                                   cl.ctypes.data_as(C.c_void_p))
    assert rc == 0
    orp, ocol, _ = O.coo_to_csr(coo, N)
    assert np.array_equal(rp, orp) and np.array_equal(cl, ocol)
    bad = coo.copy()
    bad[1, 5] = N
    assert _lib.lib().mgu_coo_to_csr(bad.ctypes.data_as(C.c_void_p), E, N, rp.ctypes.data_as(C.c_void_p),
                                     cl.ctypes.data_as(C.c_void_p)) == _lib.MGU_ERR_INVALID


def test_image_to_patches_matches_oracle():
    img = torch.from_numpy(O.formula_normal("graph/img", (5, 37, 45), seed=2))
    a, (h, w) = mgunet.PatchGraphConstructor(16).image_to_patches(img)
    b, (oh, ow) = O.image_to_patches(img, 16)
    assert (h, w) == (oh, ow) == (3, 3) and torch.equal(a, b)


def test_state_dict_keys_match_reference_names():
    m = mgunet.UNet(3, 2, 32, 4)
    assert list(m.state_dict().keys()) == list(O.unet_param_shapes(3, 2, 32, 4).keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(O.unet_param_shapes(3, 2, 32, 4)[k]), k
    assert sum(p.numel() for p in m.parameters()) == 7766018
    m.load_state_dict(O.make_unet_params(3, 2, 32, 4))  # bare state_dict layout (infer_segmentation.py:92-95)
    g = mgunet.GATNetwork(20, 128, 64, 4, 1)
    shapes, _ = O.gat_param_shapes(20, 128, 64, 4, 1)
    assert list(g.state_dict().keys()) == list(shapes.keys())
    g2 = mgunet.GATNetwork(32, 24, 8, 1, num_gat_layers=2)
    shapes2, _ = O.gat_param_shapes(32, 24, 8, 1, 2)
    assert {k: tuple(v.shape) for k, v in g2.state_dict().items()} == dict(shapes2)


def test_cpu_inputs_are_refused_not_silently_computed():
    m = mgunet.UNet(1, 2, 8, 2).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 32, 32))
    g = mgunet.GATNetwork(8, 8, 8, 2).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(4, 8), torch.zeros(2, 0, dtype=torch.long))
    with pytest.raises(ValueError):
        mgunet.UNet(3, 2, 30, 4)


def test_config_keys_and_builder():
    mc = mgunet.load_config(mgunet.config.DEFAULT_CONFIG_DIR, "model.yaml")
    assert mgunet.get_config_recursively(mc, "unet.init_features") == 32
    assert mgunet.get_config_recursively(mc, "gat.nope", 7) == 7
    u, g, pgc, mc, tc = mgunet.build_from_config()
    assert (u.in_channels, u.num_classes, u.init_features, u.depth) == (3, 2, 32, 4)
    assert pgc.patch_size == 16 and tc["learning_rate"] == 0.001 and tc["weight_decay"] == 0.0001


def test_shard_batch_covers_batch_exactly():
    for gb in (1, 7, 8, 64, 33):
        for ws in (1, 2, 3, 8):
            spans = [mgunet.shard_batch(gb, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_checkpoint_interchange_both_pth_layouts(tmp_path):
    """SURVEY 8f row 4 (checkpoint part): a `.pth` written the way the reference writes it -- a bare state_dict
    (train_segmentation.py:158-168) or {'model_state_dict': ...} (infer_segmentation.py:92-95) -- loads into the
    mirror unchanged, and a checkpoint saved from the mirror has exactly the reference's keys and values."""
    ref_sd = O.make_unet_params(3, 2, 32, 4, seed=3)
    for i, payload in enumerate((ref_sd, {"model_state_dict": ref_sd, "epoch": 7})):
        path = tmp_path / f"ck{i}.pth"
        torch.save(payload, path)
        ck = torch.load(path, map_location="cpu")
        sd = ck["model_state_dict"] if isinstance(ck, dict) and "model_state_dict" in ck else ck   # infer_segmentation.py:92-95
        m = mgunet.UNet(3, 2, 32, 4)
        res = m.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        out = tmp_path / f"out{i}.pth"
        torch.save(m.state_dict(), out)
        back = torch.load(out, map_location="cpu")
        assert list(back.keys()) == list(ref_sd.keys())
        assert all(torch.equal(back[k], ref_sd[k]) for k in ref_sd)
    # the modules of SURVEY 8f rows 1-2 carry the reference's keys too
    pred = mgunet.PatchSegmentPredictor(64, 2, hidden_dim=32, use_gnn=True, num_heads=2)
    assert list(pred.state_dict().keys()) == list(O.segment_predictor_param_shapes(64, 2, 32, True, 2).keys())
    mlp = mgunet.PatchSegmentPredictor(24, 3)
    assert list(mlp.state_dict().keys()) == list(O.segment_predictor_param_shapes(24, 3, None, False, 1).keys())
    det = mgunet.DetectionHead(96, 3)
    want = set(O.detection_head_param_shapes(96, 3).keys()) | {"conv_block.2.num_batches_tracked", "conv_block.5.num_batches_tracked"}
    assert set(det.state_dict().keys()) == want


def test_host_routines_under_asan():
    """an AddressSanitizer (+UBSan) build of the C-ABI's host routines driven over their edge cases
    (exact-size buffers, empty / ragged grids, invalid ids).  CPU only -- GPU sanitizers are not available on the pool."""
    import subprocess
    csrc = os.path.join(ROOT, "mingraph-unet_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan-host"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    exe = os.path.join(ROOT, "mingraph-unet_amd", "lib", "host_abi_check_asan")
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "host_abi_check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_adam_state_dict_layout_loads_into_torch_adam_and_steplr():
    """The optimizer_state_dict a Trainer writes into the reference's checkpoint dict (train_segmentation.py:158-163) must be a valid
    torch.optim.Adam state: load it into a real Adam and take a step; StepLR follows optim.lr_scheduler.StepLR (:105, :143)."""
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    params = list(lin.parameters())
    n = sum(p.numel() for p in params)
    m, v = torch.rand(n) * 1e-2, torch.rand(n) * 1e-4
    sd = mgunet.adam_state_dict(params, m, v, 5, 1e-3, (0.9, 0.999), 1e-8, 1e-4)
    opt = torch.optim.Adam(lin.parameters(), lr=0.5, weight_decay=0.0)
    opt.load_state_dict(sd)
    g = opt.param_groups[0]
    assert g["lr"] == 1e-3 and g["weight_decay"] == 1e-4 and tuple(g["betas"]) == (0.9, 0.999)
    st = opt.state[params[1]]
    off = params[0].numel()
    assert float(st["step"]) == 5.0 and torch.equal(st["exp_avg"].reshape(-1), m[off:off + params[1].numel()])
    for p in params:
        p.grad = torch.ones_like(p)
    opt.step()                                              # the loaded state is usable
    assert float(opt.state[params[0]]["step"]) == 6.0
    assert mgunet.adam_state_dict(params, m, v, 0, 1e-3, (0.9, 0.999), 1e-8, 0.0)["state"] == {}

    class T:                                                # StepLR only needs .lr / .set_lr
        lr = 1e-3
        def set_lr(self, lr):
            self.lr = lr
    t = T()
    sch = mgunet.StepLR(t, step_size=3, gamma=0.1)
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    ref = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=3, gamma=0.1)
    for _ in range(8):
        ref_opt.step()
        ref.step()
        sch.step()
        assert abs(sch.get_last_lr()[0] - ref.get_last_lr()[0]) <= 1e-12
