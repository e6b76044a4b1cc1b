"""The C-ABI library loads and exports exactly what include/restormer_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "restormer_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rst_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from superresolution_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("librestormer_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.rst_version() == _lib.ABI_VERSION
    assert lib.rst_last_error() is not None


def test_struct_layouts_match_header():
    from superresolution_amd import _lib
    text = open(HEADER).read()
    body = re.search(r"typedef struct rst_block_params \{(.*?)\} rst_block_params;", text, re.S).group(1)
    fields = re.findall(r"const float\*\s+(\w+);", body)
    assert tuple(fields) == _lib.PARAM_FIELDS
    gbody = re.search(r"typedef struct rst_block_grads \{(.*?)\} rst_block_grads;", text, re.S).group(1)
    assert tuple(re.findall(r"float\*\s+(\w+);", gbody)) == _lib.PARAM_FIELDS
    assert ctypes.sizeof(_lib.BlockDesc) == 32
    assert ctypes.sizeof(_lib.BlockParams) == 8 * len(_lib.PARAM_FIELDS)


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before anything touches a device."""
    from superresolution_amd import _lib
    lib = _lib.load()
    assert lib.rst_layernorm_fwd(None, None, None, None, 1, 4, 16, None) == 1
    assert b"null" in lib.rst_last_error()
    d = _lib.BlockDesc(1, 48, 8, 8, 5, 60, 1, 0)            # 48 % 5 != 0
    assert lib.rst_block_workspace_bytes(ctypes.byref(d)) == 0
    d = _lib.BlockDesc(2, 48, 16, 24, 1, 60, 1, 0)
    assert lib.rst_block_workspace_bytes(ctypes.byref(d)) > 0
    assert lib.rst_block_saved_bytes(ctypes.byref(d)) > lib.rst_attention_saved_bytes(ctypes.byref(d))


def test_module_surface_matches_reference_contract():
    """Same names/shapes/order as the reference state-dict (oracle.model_param_shapes
    restates model/Restromer.py:251-383 and is pinned against the reference by the fixtures)."""
    from oracle import restormer_oracle as orc
    from superresolution_amd.model import Restromer as R
    for kwargs in (dict(inp_channels=3, out_channels=3, dim=48), dict(), dict(dim=16, bias=True, LayerNorm_type="BiasFree"),
                   dict(dim=8, dual_pixel_task=True, bias=True)):
        net = R.Restormer(**kwargs)
        got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        assert got == orc.model_param_shapes(**kwargs)
        assert [k for k, _ in net.named_parameters()] == [k for k, _ in got]
    for name in ("Restormer", "TransformerBlock", "Attention", "FeedForward", "LayerNorm", "WithBias_LayerNorm",
                 "BiasFree_LayerNorm", "OverlapPatchEmbed", "Downsample", "Upsample", "to_3d", "to_4d"):
        assert hasattr(R, name)


def test_cpu_tensors_fail_loudly():
    import torch
    from superresolution_amd.model.Restromer import TransformerBlock
    blk = TransformerBlock(12, 1, 1.25, False, "WithBias")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.randn(1, 12, 4, 4))


def test_one_stream_is_the_default(monkeypatch):
    """The library and the trainer keep a GPU's work on one stream unless asked: dense v_mfma_f32_16x16x32 waves were
    measured to corrupt a dependent multiply-add chain of a kernel from another stream on MI355X
    (tools/two_stream_repro.hip), so every overlap of two launch sequences is opt-in."""
    import torch.nn as nn
    from superresolution_amd import _lib
    from superresolution_amd.model.Restromer import Restormer
    from superresolution_amd.parallel import DataParallelTrainer
    monkeypatch.delenv("RST_BWD_STREAMS", raising=False)
    monkeypatch.delenv("RST_DP_OVERLAP", raising=False)
    monkeypatch.delenv("RST_TWO_STREAMS", raising=False)
    assert _lib.load().rst_bwd_side_stream_enabled() == 0
    assert Restormer.two_streams is False
    assert DataParallelTrainer(nn.Linear(2, 2), world_size=1).overlap is False
