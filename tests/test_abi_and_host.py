"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol that
include/resunet_hip.h declares; the ctypes table covers the header; the host mirrors keep the reference's
surface (state-dict keys/order, checkpoint layout, list-in/list-out) and fail loudly without a GPU."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import resunet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "resunet_hip.h")
T = torch.from_numpy


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ru_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from brats2019_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        from brats2019_amd import build
        build.build(verbose=False)
    return L.load()


def test_library_exports_every_header_symbol(lib):
    from brats2019_amd import _lib as L
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libresunet_hip.so does not export %s" % n
    assert sorted(L.SIGNATURES.keys()) == names, "ctypes table and header disagree"
    assert lib.ru_version() >= 100
    assert isinstance(lib.ru_last_error(), bytes)


def test_param_layout_matches_reference_state_dict_order(lib):
    from brats2019_amd.engine import ParamLayout
    for cfg in (O.DEFAULT_CFG, dict(depth=3, encoder_layers=[1, 1, 2], decoder_layers=[1, 1, 1], number_of_channels=[8, 16, 32], number_of_outputs=3)):
        lay = ParamLayout(**cfg)
        spec = O.state_dict_spec(**cfg)                       # pinned to the reference by tests/test_oracle_golden.py
        assert [k for k, _ in spec] == list(lay.entries.keys())
        off = 0
        for k, shape in spec:
            assert lay.entries[k][0] == tuple(shape) and lay.entries[k][1] == off
            off += int(np.prod(shape))
        assert lay.total == off
        dead = {k for k, v in lay.entries.items() if v[2]}
        d = cfg["depth"] - 1
        assert dead == {k for k, _ in spec if k.startswith("decoder_convs.%d." % d) or k.startswith("decoder_convs1x1.%d." % d)}
    full = ParamLayout(**O.DEFAULT_CFG)
    assert full.total == 5427955 and sum(int(np.prod(v[0])) for v in full.entries.values() if v[2]) == 918016


def test_unet_module_surface(golden):
    from brats2019_amd import model as M
    g = golden("unet32")
    net = M.UNet(**O.DEFAULT_CFG)
    keys = list(net.state_dict().keys())
    assert keys == [k for k, _ in O.state_dict_spec(**O.DEFAULT_CFG)]
    assert set(g["dead_params"].tolist()) <= set(keys)
    for name in ("UNet", "Residual", "conv", "Trilinear"):
        assert hasattr(M, name)
    # HIP-only: a CPU forward must fail loudly, not fall back
    with pytest.raises(RuntimeError):
        net([torch.zeros(1, 4, 8, 8, 8)])
    with pytest.raises(RuntimeError):
        M.Residual(8, 8, 1)(torch.zeros(1, 8, 4, 4, 4))


def test_bad_configuration_is_rejected(lib):
    from brats2019_amd.engine import ParamLayout
    with pytest.raises(RuntimeError):
        ParamLayout(depth=3, encoder_layers=[1, 1, 1], decoder_layers=[1, 1, 1], number_of_channels=[12, 24, 48], number_of_outputs=3)
    from brats2019_amd.engine import UNetEngine
    eng = UNetEngine(**O.DEFAULT_CFG)
    assert lib.ru_unet_workspace_bytes(eng.h, 1, 20, 16, 16, 0) == 0          # 20 is not divisible by 8
    assert b"divisible" in lib.ru_last_error()
    assert lib.ru_unet_workspace_bytes(eng.h, 1, 16, 16, 16, 1) > lib.ru_unet_workspace_bytes(eng.h, 1, 16, 16, 16, 0) > 0


def test_reference_checkpoint_unpickles_into_host_mirror(golden, tmp_path):
    """tests/golden/ckpt/tiny/tinybest_model.pth was written by the REFERENCE's Trainer._save (train.py:320-324)."""
    import shutil
    from brats2019_amd import train as TR, model as M
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "ckpt", "tiny"), tmp_path / "tiny")
    saved = {k: sys.modules.get(k) for k in ("model", "train", "loss")}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        tr = TR.Trainer(name="tiny", models_root=str(tmp_path), model=None, rewrite=False, connect_tb=False)
        assert tr.resume_training
        tr.load_best()
        # the fixture is what the reference really writes: main.py:61 wraps the net, so the pickle is DataParallel(UNet) and its keys
        # carry the `module.` prefix (export_onnx_group_norm.py:28-32); a wrapper without several devices stays as pickled
        assert isinstance(tr.model, torch.nn.DataParallel) and isinstance(tr.model.module, M.UNet) and isinstance(tr.state, TR.TrainingState)
        assert (tr.state.epoch, tr.state.global_step, tr.state.best_val, tr.state.cuda) == (3, 77, 1.25, False)
        cfg = dict(depth=2, encoder_layers=[1, 1], decoder_layers=[1, 1], number_of_channels=[8, 16], number_of_outputs=3)
        params = O.make_params(23, **cfg)
        sd = tr.model.state_dict()
        assert list(sd.keys()) == ["module." + k for k in params.keys()]
        for k, v in params.items():
            assert np.array_equal(sd["module." + k].numpy(), v), k
        # several devices in the wrapper (the reference under --gpus 8): the module itself is kept -- one process per GPU here
        many = torch.nn.DataParallel(module=M.UNet(**cfg))
        many.device_ids = [0, 1]
        os.makedirs(tmp_path / "many", exist_ok=True)
        torch.save({"state": tr.state, "model": many}, tmp_path / "many" / "manybest_model.pth")
        trm = TR.Trainer(name="many", models_root=str(tmp_path), model=None, rewrite=False, connect_tb=False)
        trm.load_best()
        assert isinstance(trm.model, M.UNet)
        with pytest.raises(RuntimeError, match="one process per GPU"):
            M.UNet(**cfg)._replicate_for_data_parallel()
        # a wrapped mirror takes an un-wrapped checkpoint and the reverse (state-dict route, train.py:331-333)
        plain = M.UNet(**cfg)
        plain.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        torch.save({"state": tr.state, "model": plain}, tmp_path / "many" / "manylast_model.pth")
        trw = TR.Trainer(name="many", models_root=str(tmp_path), model=torch.nn.DataParallel(module=M.UNet(**cfg)), rewrite=False, connect_tb=False)
        trw.load_latest()
        assert all(np.array_equal(trw.model.module.state_dict()[k].numpy(), v) for k, v in params.items())
        # state-dict route into a freshly constructed mirror, incl. the DataParallel `module.` prefix
        net = M.UNet(**cfg)
        tr2 = TR.Trainer(name="tiny", models_root=str(tmp_path), model=net, rewrite=False, connect_tb=False)
        tr2.load_best()
        assert all(np.array_equal(net.state_dict()[k].numpy(), v) for k, v in params.items())
        # our own save -> load round trip keeps the reference file naming (no separator)
        tr2.state.epoch = 9
        tr2._save("last_model")
        assert os.path.exists(tmp_path / "tiny" / "tinylast_model.pth")
        tr3 = TR.Trainer(name="tiny", models_root=str(tmp_path), model=None, connect_tb=False)
        tr3.load_latest()
        assert tr3.state.epoch == 9 and list(tr3.model.state_dict().keys()) == list(params.keys())
    finally:
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)


def test_tiling_helpers_match_reference(golden):
    from brats2019_amd import tiling
    g = golden("tiling")
    grid = tiling.grid_for(g["shape"], g["center"])
    assert grid == list(g["grid"])
    idx = 0
    for i in range(grid[0]):
        for j in range(grid[1]):
            for k in range(grid[2]):
                lo, hi = tiling.get_indices((i, j, k), g["center"], g["border"])
                assert lo == list(g["index_min"][idx]) and hi == list(g["index_max"][idx])
                idx += 1
    # (the tile extract / centre paste themselves are device kernels: tests/test_inference.py::test_tile_gather_scatter_match_reference)


def test_shard_assignment():
    from brats2019_amd.parallel import DataParallelStep as S
    assert [S.shard(32, r, 8) for r in range(8)] == [slice(4 * r, 4 * r + 4) for r in range(8)]
    with pytest.raises(ValueError):
        S.shard(6, 0, 4)


def test_argument_errors_are_reported_not_thrown(lib):
    """every entry returns a negative code + ru_last_error() on a bad argument (SURVEY 8(b) conventions); none of these calls touches a GPU"""
    import ctypes as C
    from brats2019_amd import _lib as L
    from brats2019_amd.engine import ParamLayout
    lay = ParamLayout(**O.DEFAULT_CFG)
    assert lib.ru_unet_set_fusion(lay.handle, 0) == 0 and lib.ru_unet_set_fusion(lay.handle, 3) == 0 and lib.ru_unet_set_fusion(lay.handle, 63) == 0
    assert lib.ru_unet_set_fusion(lay.handle, 64) < 0 and b"ru_unet_set_fusion" in lib.ru_last_error()
    assert lib.ru_unet_set_precision(lay.handle, 7) < 0
    assert lib.ru_unet_workspace_bytes(lay.handle, 1, 12, 16, 16, 0) == 0            # extents must be divisible by 2^(depth-1)
    assert lib.ru_unet_workspace_bytes(lay.handle, 1, 16, 16, 16, 1) > lib.ru_unet_workspace_bytes(lay.handle, 1, 16, 16, 16, 0) > 0
    assert lib.ru_allreduce(None, None, 0, L.PRECISIONS["f32"], None) < 0 and b"ru_allreduce" in lib.ru_last_error()
    h = C.c_void_p()
    assert lib.ru_comm_init(C.byref(h), None, 0, 1) < 0 and lib.ru_comm_init(None, None, 0, 0) < 0
    assert lib.ru_comm_destroy(None) == 0 and lib.ru_comm_rank(None) == -1 and lib.ru_comm_world(None) == 0
    assert lib.ru_criterion_value_device(None, 3, 1.0, 1.0, 0.5, 0.5, None, None) < 0
    # round-3 entries: inference driver kernels, optimizer, family probe (argument checks come before any launch)
    i3 = lambda *v: (C.c_int * len(v))(*v)
    one = C.c_void_p(64)                                  # any non-null pointer: never dereferenced by a failing check
    assert lib.ru_tile_gather(None, None, 1, 1, 8, 8, 8, 1, None, 8, 8, 8, None) < 0 and b"ru_tile_gather" in lib.ru_last_error()
    assert lib.ru_tile_gather(one, one, 1, 1, 8, 8, 8, 1, i3(0, 0, 0), 8, 8, 6, None) < 0 and b"multiple of 4" in lib.ru_last_error()
    assert lib.ru_tile_scatter(one, one, 1, 1, 8, 8, 8, 1, i3(0, 0, 0), 8, 8, 8, i3(2, 2, 2), i3(8, 8, 8), None) < 0 and b"centre block outside" in lib.ru_last_error()
    assert lib.ru_tile_scatter(one, one, 1, 1, 8, 8, 8, 1, i3(-4, 0, 0), 8, 8, 8, i3(2, 2, 2), i3(4, 4, 4), None) < 0 and b"before the volume start" in lib.ru_last_error()
    assert lib.ru_case_bbox(None, None, 4, 8, 8, 8, None) < 0
    assert lib.ru_case_stats(one, one, 4, 8, 8, 8, i3(0, 0, 0), i3(9, 8, 8), one, 1 << 20, None) < 0 and b"inside the volume" in lib.ru_last_error()
    assert lib.ru_case_prepare(one, one, one, 4, 8, 8, 8, i3(0, 0, 0), i3(8, 8, 8), i3(1, 0, 0), i3(8, 16, 16), 4, 0, None) < 0 and b"does not fit" in lib.ru_last_error()
    assert lib.ru_tta_merge_box(one, 9, 0, None, one, one, 3, 8, 8, 8, i3(0, 0, 0), i3(8, 8, 8), None) < 0
    assert lib.ru_cc_reject(one, 8, 8, 8, 0.1, one, 16, None) < 0 and b"workspace too small" in lib.ru_last_error()
    assert lib.ru_cc_workspace_bytes(8, 8, 8) >= 2 * 512 * 4 and lib.ru_case_workspace_bytes(4, 8, 8, 8) > 0
    assert lib.ru_paste_labels(one, one, 8, 8, 8, i3(4, 0, 0), i3(8, 8, 8), None) < 0
    assert lib.ru_adam_step(None, None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) < 0 and b"ru_adam_step" in lib.ru_last_error()
    ms, cnt = (C.c_double * 7)(), (C.c_int * 7)()
    assert lib.ru_unet_probe_read_families(lay.handle, ms, cnt, 6) < 0 and lib.ru_unet_probe_read_families(lay.handle, ms, cnt, 7) == 0 and sum(cnt) == 0


def test_default_precision_rule():
    from brats2019_amd import model as M
    from brats2019_amd import _lib as L
    assert M.default_precision([16, 32, 64, 128]) == "bf16x3" and M.default_precision([8, 16, 32]) == "f32"
    net = M.UNet(**O.DEFAULT_CFG)
    assert net._get_engine().precision == "bf16x3"
    net.set_precision("f32")
    assert net._get_engine().precision == "f32"
    with pytest.raises(ValueError):
        net.set_precision("fp16")
    # gradient precision: three products unless asked otherwise; the setting survives until the engine exists and is applied to it
    assert net._get_engine().grad_precision == "bf16x3"
    lib = L.load()
    assert lib.ru_unet_get_grad_precision(net._get_engine().h) == L.GRAD_PRECISIONS["bf16x3"]
    net.set_grad_precision("bf16")
    assert net._get_engine().grad_precision == "bf16" and lib.ru_unet_get_grad_precision(net._get_engine().h) == L.GRAD_PRECISIONS["bf16"]
    net2 = M.UNet(**O.DEFAULT_CFG).set_grad_precision("bf16")
    assert net2._get_engine().grad_precision == "bf16"
    with pytest.raises(ValueError):
        net.set_grad_precision("f32")
    assert lib.ru_unet_set_grad_precision(net._get_engine().h, 0) < 0 and lib.ru_unet_set_grad_precision(None, 1) < 0


def test_inference_workspace_recycles_block_temporaries():
    """ru_unet_workspace_bytes (host-side dry walk, no GPU): an inference forward rewinds the arena behind every block's output, so its
    workspace is less than a third of the training one at the same shape, grows linearly with the batch, and 8 tiles of 192^3 -- the
    reference's sliding-window tile (train.py:158-174) -- fit in 26 GiB (45 GiB before the rewinds)."""
    from brats2019_amd import _lib as L
    from brats2019_amd.engine import UNetEngine
    lib = L.load()
    eng = UNetEngine(precision="bf16x3")
    ws = lambda n, s, tr: lib.ru_unet_workspace_bytes(eng.h, n, s, s, s, tr)
    assert 0 < ws(4, 128, 0) < ws(4, 128, 1) / 3
    assert abs(ws(8, 128, 0) - 2 * ws(4, 128, 0)) < 0.05 * ws(8, 128, 0)
    assert ws(8, 192, 0) < 26 * 2 ** 30
    assert ws(1, 36, 0) == 0                      # extents must be divisible by 8 (three stride-2 levels): size query says 0


@pytest.mark.skipif(not os.path.isfile("/root/reference/model.py"), reason="needs the reference sources (build container only; the GPU box has none)")
def test_checkpoint_written_here_loads_into_the_reference_unet(tmp_path):
    """The reverse direction of test_reference_checkpoint_unpickles_into_host_mirror: `<name>best_model.pth` written by THIS Trainer._save,
    read the way the reference's Trainer._load reads it (train.py:326-333: torch.load, then `model.load_state_dict(s['model'].state_dict())`)
    into the reference's own model.UNet -- un-wrapped and, as main.py:61 builds it, wrapped in nn.DataParallel."""
    import contextlib
    import importlib.util
    import io
    from brats2019_amd import train as TR, model as M
    cfg = dict(depth=2, encoder_layers=[1, 1], decoder_layers=[1, 1], number_of_channels=[8, 16], number_of_outputs=3)
    params = O.make_params(29, **cfg)
    spec = importlib.util.spec_from_file_location("_reference_model_py", "/root/reference/model.py")
    ref = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(ref)
        mk_ref = lambda: ref.UNet(cfg["depth"], cfg["encoder_layers"], cfg["decoder_layers"], cfg["number_of_channels"], cfg["number_of_outputs"])
        ref_plain, ref_wrapped = mk_ref(), torch.nn.DataParallel(module=mk_ref())
    for tag, wrap in (("plain", False), ("wrapped", True)):
        net = M.UNet(**cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        mine = torch.nn.DataParallel(module=net) if wrap else net
        tr = TR.Trainer(name="rev" + tag, models_root=str(tmp_path), model=mine, rewrite=True, connect_tb=False)
        tr.state.epoch, tr.state.global_step = 5, 123
        tr._save("best_model")
        s = torch.load(tmp_path / ("rev" + tag) / ("rev" + tag + "best_model.pth"), map_location=torch.device("cpu"), weights_only=False)   # train.py:327
        assert (s["state"].epoch, s["state"].global_step) == (5, 123)
        target = ref_wrapped if wrap else ref_plain
        target.load_state_dict(s["model"].state_dict())                                    # train.py:333 (strict: names, order and shapes)
        got = (target.module if wrap else target).state_dict()
        assert list(got.keys()) == list(params.keys())
        for k, v in params.items():
            assert np.array_equal(got[k].numpy(), v), (tag, k)


def test_tiling_copy_and_copy_back_on_host_tensors_match_reference(golden):
    """`tiling.copy` / `copy_back` on HOST tensors (the reference's loader_helper.copy / copy_back take any tensor, train.py:165-171 hands them
    CPU volumes): the slicing path that serves tensors the device kernels do not take (CPU, or a tile width that is not a multiple of 4)
    against the tiles and the pasted volume the REFERENCE produced (tests/golden/tiling.npz)."""
    from brats2019_amd import tiling
    g = golden("tiling")
    data = torch.from_numpy(g["small_data"])
    c, b, tl = (8, 8, 8), (4, 4, 4), (16, 16, 16)
    grid = tiling.grid_for(data.shape[2:], c)
    res = torch.zeros_like(data)
    n = 0
    for i in range(grid[0]):
        for j in range(grid[1]):
            for k in range(grid[2]):
                lo, hi = tiling.get_indices((i, j, k), c, b)
                tile = tiling.copy(data, tl, lo, hi)
                assert np.array_equal(tile.numpy(), g["small_tiles"][n]), (i, j, k)
                tiling.copy_back(res, tile, c, lo, hi, b)
                n += 1
    assert np.array_equal(res.numpy(), g["small_result"])


def test_tiling_host_path_dtype_and_negative_origin_follow_the_reference():
    """round-4 advisor finding: loader_helper.copy returns float32 for ANY volume dtype (loader_helper.py:43) and copy_back clamps a centre
    block that starts before the volume (:86) -- the host path must do both (an integer label volume; index_min + border < 0)."""
    from brats2019_amd import tiling
    lab = torch.arange(2 * 1 * 6 * 6 * 6, dtype=torch.int16).reshape(2, 1, 6, 6, 6)
    tile = tiling.copy(lab, (8, 8, 8), [-2, -2, -2], [6, 6, 6])
    assert tile.dtype == torch.float32 and tuple(tile.shape) == (2, 1, 8, 8, 8)
    assert torch.equal(tile[:, :, 2:, 2:, 2:], lab.float()) and float(tile[:, :, :2].abs().sum()) == 0.0
    # centre block of 4 starting at index_min + border = -3 + 1 = -2: only its last two planes land, at volume planes 0..1
    vol = torch.zeros(1, 1, 6, 6, 6)
    t = torch.arange(6 ** 3, dtype=torch.float32).reshape(1, 1, 6, 6, 6)
    tiling.copy_back(vol, t, (4, 4, 4), [-3, -3, -3], [3, 3, 3], (1, 1, 1))
    want = torch.zeros_like(vol)
    want[:, :, 0:2, 0:2, 0:2] = t[:, :, 3:5, 3:5, 3:5]
    assert torch.equal(vol, want)


def test_unet_parameter_cache_follows_replaced_parameters():
    """round-4 advisor finding: load_state_dict(assign=True) and an overwrite-on-conversion _apply create NEW nn.Parameter objects; the
    cached list the forward hands to autograd must be rebuilt, not keep the objects the module no longer owns."""
    from brats2019_amd import model as M
    net = M.UNet(**O.DEFAULT_CFG)
    before = net._param_list()
    assert all(a is b for a, b in zip(before, net.parameters()))
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()}, assign=True)
    after = net._param_list()
    assert all(a is b for a, b in zip(after, net.parameters())) and any(a is not b for a, b in zip(before, after))
    torch.__future__.set_overwrite_module_params_on_conversion(True)
    try:
        net.double()
    finally:
        torch.__future__.set_overwrite_module_params_on_conversion(False)
    assert all(a is b for a, b in zip(net._param_list(), net.parameters()))
    assert "_flat_checked" not in net.__dict__ and "_flat_ends" not in net.__dict__
    # round-5 advisor finding: every Parameter lives in a SUBMODULE, so re-registering one there never passes through UNet's own hooks -- the cached list is
    # checked against the submodules' parameter dictionaries by identity on every use
    net.float()
    held = net._param_list()
    net.__dict__["_flat_checked"] = True                     # (as after a forward)
    net.conv_output.weight = torch.nn.Parameter(torch.zeros_like(net.conv_output.weight))
    now = net._param_list()
    assert all(a is b for a, b in zip(now, net.parameters())) and any(a is not b for a, b in zip(held, now))
    assert "_flat_checked" not in net.__dict__


def test_bench_roofline_bookkeeping():
    """bench.py's bounds are arithmetic on SURVEY 8(d)'s figures -- pinned so that a change of the layer table shows: algorithmic FLOPs of a
    step = 4 x 890.87 GFLOP, fused-lower-bound bytes 37.9 GB, whole-step algorithmic bound 4.94 ms, GroupNorm achievable-fusion bytes 7.52 GB (8.05 before the head conv took over the last residual pass);
    the committed PMC family table parses into eight rows."""
    import bench
    fb = bench.family_bounds(4, 128, "bf16x3")
    gflop = sum(f["gflop"] for f in fb.values())
    gbytes = sum(f["gbytes"] for k, f in fb.items() if k != "groupnorm")
    assert abs(gflop - 4 * bench.FWDBWD_GFLOP_PER_VOL) < 1.0, gflop
    assert abs(gbytes - 37.91) < 0.05, gbytes
    assert abs(sum(f["bound_ms_algorithmic"] for f in fb.values()) - 4.936) < 0.01
    assert abs(fb["groupnorm"]["achievable_gbytes"] - 7.516) < 0.01 and abs(fb["groupnorm"]["bound_ms_achievable_fusion"] - 0.9395) < 0.001
    assert bench.step_roofline_ms(4, 128, "bf16x3") > sum(f["bound_ms_algorithmic"] for f in fb.values())          # executed products cost more than algorithmic ones
    # the achievable bounds of the line (round 5): + GroupNorm passes no fusion removes (7.52 GB), + the y re-read of the 25 norms' backward sums
    # (one tensor each: 3.76 GB) and the (y, d) reads of the 16-channel level's apply inside its weight gradient (2 tensors x 4 norms: 4.29 GB)
    ex = fb["groupnorm"]["achievable_extra_gbytes"]
    assert abs(ex["bst_y_reread"] - 3.758) < 0.005 and abs(ex["wgrad_l0_fused_apply"] - 4.295) < 0.005
    ab = bench.achievable_bounds(fb)
    assert abs(ab["algorithmic_ms"] - 4.936) < 0.01 and abs(ab["achievable_ms"] - 5.875) < 0.01 and abs(ab["achievable_all_ms"] - 6.882) < 0.01
    assert abs(ab["gbytes_fused_lower_bound"] - 37.91) < 0.05 and abs(ab["gbytes_achievable"] - 45.42) < 0.05 and abs(ab["gbytes_achievable_all"] - 53.48) < 0.05
    assert abs((ab["achievable_ms"] + 4.0 * 16 * 4 * 128 ** 3 / 8e12 * 1e3) / 15.833 - 0.375) < 0.002      # with the last residual pass counted (round 4's dataflow): the round-4 driver line, recomputed by its judge as 0.376
    # round 6: the largest single instantiations (`roofline_top` rows with "instance").  The 16-channel level's weight gradient with the fused GroupNorm-backward
    # apply, at the 451 us per launch of profiles/r05_serial_train_step_kernel_stats.txt: 0.30 of its algorithmic bytes (x + dy once), 0.61 of the 2.19 GB it moves
    ib = bench.instance_bounds(4, 128)
    assert abs(ib["conv16_fwd"]["gflop"] - 4 * 115.96) < 0.1 and abs(ib["conv16_fwd"]["gbytes"] - 4 * 1.0737) < 0.001 and ib["conv_deep_fwd"]["launches"] == 20
    rows = bench.instance_rows({"wgrad16_fused_apply": (3 * 4 * 0.451, 12), "conv16_fwd": (0.0, 0)}, ("wgrad16_fused_apply", "conv16_fwd"), 3, 4, 128, "bf16x3")
    assert len(rows) == 1 and rows[0]["launches_per_step"] == 4 and abs(rows[0]["frac_algorithmic"] - 0.2976) < 0.002
    if rows[0]["moved_gbytes_per_step"] is not None:         # (the committed step-traffic table names the kernel: true for every round's table so far)
        assert 0.45 < rows[0]["frac_moved"] < 0.75 and rows[0]["moved_gbytes_per_step"] > 2 * rows[0]["algorithmic_gbytes_per_step"] * 0.9
    ft = bench.committed_family_table()
    assert ft is not None and len(ft["rows"]) >= 8 and {r["channels"] for r in ft["rows"]} == {16, 32, 64, 128}
    assert all(0 < r["mfma_busy_pct"] < 100 and r["avg_us"] > 0 for r in ft["rows"])
