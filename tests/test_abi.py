"""CPU: the C-ABI library loads, exports every symbol include/invflow.h declares, and its argument
validation / error reporting works without a GPU (no compute call is made)."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT

HEADER = os.path.join(ROOT, "include", "invflow.h")


@pytest.fixture(scope="module")
def H():
    import importlib.util
    spec = importlib.util.spec_from_file_location("invflow_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()  # hipcc cross-compiles gfx950 without a GPU
    import invflow_hip
    invflow_hip.lib()
    return invflow_hip


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ifl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(H):
    syms = declared_symbols()
    assert len(syms) >= 12
    L = ctypes.CDLL(H.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libinvflow_hip.so does not export %s" % s
    # and the Python binding table mirrors the header one to one
    assert sorted(H.SIGNATURES) == syms


def test_version_and_workspace(H):
    L = H.lib()
    assert L.ifl_version() >= 1000
    for op in range(5):
        assert L.ifl_workspace_bytes(op, 128, 64, 32, 32, 3, 3, 0) >= 0
    assert L.ifl_workspace_bytes(H.OP_INVERSE, 128, 64, 32, 32, 3, 3, 0) >= 64 * 64 * 9 * 4
    assert L.ifl_workspace_bytes(H.OP_INVERSE, -1, 64, 32, 32, 3, 3, 0) == 0
    assert L.ifl_conv2d_workspace_bytes(8, 4, 4, 8, 8, 3, 3, 1, 1) >= 4 * 4 * 9 * 4
    assert L.ifl_carry_bytes(64, 3, 3) >= 256 + 2 * 64 * 64 * 9 * 4 and L.ifl_carry_bytes(0, 3, 3) == 0
    assert L.ifl_scan_state_bytes() >= 128 * 64 * 1024  # generations + mailbox of 128 images (caller-owned block)


def test_argument_validation_without_gpu(H):
    L = H.lib()
    # bad shape -> IFL_EINVAL with a message, nothing is launched
    rc = L.ifl_inverse_f32(None, None, None, 2, 0, 5, 5, 3, 3, 0, 0, None, 0, None, None, None)
    assert rc == -1 and b"bad shape" in L.ifl_last_error()
    rc = L.ifl_inverse_f32(None, None, None, 2, 4, 5, 5, 3, 3, 7, 0, None, 0, None, None, None)
    assert rc == -1 and b"unknown order" in L.ifl_last_error()
    # empty batch is a no-op success and clears the error
    assert L.ifl_inverse_f32(None, None, None, 0, 4, 5, 5, 3, 3, 0, 0, None, 0, None, None, None) == 0
    assert L.ifl_last_error() == b""
    assert L.ifl_forward_f32(None, None, None, None, 0, 4, 5, 5, 3, 3, 0, 0, None, 0, None) == 0
    # null tensors with a non-empty batch
    rc = L.ifl_inverse_f32(None, None, None, 1, 4, 5, 5, 3, 3, 0, 0, None, 0, None, None, None)
    assert rc == -1 and b"null tensor" in L.ifl_last_error()
    rc = L.ifl_conv2d_f32(None, None, None, None, 1, 4, 4, 2, 2, 3, 3, 0, 0, None, 0, None)
    assert rc == -1 and b"kernel larger" in L.ifl_last_error()


def test_host_checks_match_reference_wording(H):
    import torch
    x = torch.zeros(2, 4, 5, 5)
    w = torch.zeros(4, 4, 3, 3)
    # CHECK_CUDA / CHECK_CONTIGUOUS of inv_conv_with_bp_general.cpp:15-17
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        H.inverse(x, w)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        H.forward(x, w)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        H.backward(x, x, w)
    import inv_conv_with_bp
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        inv_conv_with_bp.inverse(x, w, torch.zeros_like(x))
    assert {"inverse", "forward", "dy", "dw"} <= set(dir(inv_conv_with_bp))


def test_missing_library_fails_loudly(H, monkeypatch):
    import invflow_hip
    monkeypatch.setattr(invflow_hip, "_lib", None)
    monkeypatch.setattr(invflow_hip, "LIB_PATH", "/nonexistent/libinvflow_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        invflow_hip.lib()
