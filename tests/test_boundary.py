"""The drop-in boundary: the C-ABI library loads, exports exactly what include/mvhmr_unproject.h declares,
validates arguments without touching a GPU, and the product never reaches into oracle/."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from multiviewhmr_amd import _capi

HEADER = os.path.join(ROOT, "include", "mvhmr_unproject.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvhmr_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(_capi.LIB_PATH), "run `python -m multiviewhmr_amd.build` (or __graft_entry__.build())"
    assert os.path.realpath(_capi.LIB_PATH).startswith(os.path.realpath(ROOT))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert sorted(_capi.EXPORTS) == declared, "the ctypes binding and the header disagree"


def test_code_object_targets_gfx950_only():
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in blob


def _desc(**kw):
    d = _capi.Desc()
    d.abi_version = _capi.ABI_VERSION
    d.batch, d.views, d.channels, d.feat_h, d.feat_w = 2, 4, 32, 24, 24
    d.vol_x = d.vol_y = d.vol_z = 8
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_argument_validation_without_a_gpu():
    L = _capi.lib()
    assert L.mvhmr_abi_version() == _capi.ABI_VERSION
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(256)       # never dereferenced: validation fails first
    # unknown method -> the reference's ValueError text
    rc = L.mvhmr_unproject_forward(ctypes.byref(_desc(method=9)), one, one, one, one, null, 0, null)
    assert rc == _capi.ERR_INVALID_ARGUMENT and b"Unknown aggregation_method" in L.mvhmr_last_error()
    with pytest.raises(ValueError):
        _capi.check(rc)
    # null pointers, bad sizes, too many views, wrong ABI
    assert L.mvhmr_unproject_forward(ctypes.byref(_desc()), null, one, one, one, null, 0, null) == _capi.ERR_INVALID_ARGUMENT
    assert L.mvhmr_unproject_forward(ctypes.byref(_desc(channels=0)), one, one, one, one, null, 0, null) == _capi.ERR_INVALID_ARGUMENT
    assert L.mvhmr_unproject_forward(ctypes.byref(_desc(views=17)), one, one, one, one, null, 0, null) == _capi.ERR_UNSUPPORTED
    assert L.mvhmr_unproject_forward(ctypes.byref(_desc(abi_version=99)), one, one, one, one, null, 0, null) == _capi.ERR_INVALID_ARGUMENT
    # planar features need a layout workspace; none given -> workspace error, not a crash
    d = _desc(variant=_capi.VARIANT["gather"])
    need = L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(d))
    assert need >= 2 * 4 * 24 * 24 * 32 * 4
    assert L.mvhmr_unproject_forward(ctypes.byref(d), one, one, one, one, null, 0, null) == _capi.ERR_WORKSPACE
    assert L.mvhmr_unproject_forward(ctypes.byref(d), one, one, one, one, ctypes.c_void_p(8), need, null) == _capi.ERR_WORKSPACE
    with pytest.raises(RuntimeError):
        _capi.check(_capi.ERR_WORKSPACE)
    # channels-last fp32 input: forward needs no scratch at all, backward accumulates in place
    d = _desc(feat_layout=_capi.LAYOUT_BVHWC, variant=_capi.VARIANT["gather"])
    assert L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(d)) == 0
    assert L.mvhmr_unproject_backward_workspace_bytes(ctypes.byref(d)) == 0
    # AUTO on planar input of a shape both variants serve: one converted copy + the device-side gate counter (csrc/gate.h)
    # (from 96 bricks' worth of voxels on; smaller launches go to the gather kernels without asking)
    kw = dict(vol_x=64, vol_y=64, vol_z=32)
    fixed = L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(_desc(variant=_capi.VARIANT["brick"], **kw)))
    gated = L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(_desc(variant=_capi.VARIANT["auto"], **kw)))
    assert fixed > 0 and gated == fixed + 256
    small = L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(_desc(variant=_capi.VARIANT["auto"], vol_x=8, vol_y=8, vol_z=32)))
    assert small == L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(_desc(variant=_capi.VARIANT["gather"], vol_x=8, vol_y=8, vol_z=32)))
    assert L.mvhmr_status_string(_capi.ERR_LAUNCH) == b"kernel launch failed"


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing shipped may import, load or mention it."""
    pkg = os.path.join(ROOT, "multiviewhmr_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "libmvhmr_oracle" not in text, fn
                assert "grid_sample" not in text or fn.endswith((".py", ".h", ".hip")) and "F.grid_sample(" not in text, fn


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.lib()


def test_torch_extension_builds_and_speaks_the_abi():
    """the PyTorch-ROCm C++ extension over the C ABI (csrc_ext/mvhmr_torch_ext.cpp) builds in-tree without a GPU and registers its ops"""
    import torch
    from multiviewhmr_amd import build
    ext = build.build_ext()
    assert os.path.exists(ext)
    torch.ops.load_library(ext)
    assert torch.ops.mvhmr_native.abi_version() == _capi.ABI_VERSION
    for name in ("unprojection", "unprojection_backward"):
        assert hasattr(torch.ops.mvhmr_native, name)
