"""CPU: the C-ABI library builds/loads and exports every symbol include/mllp_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from mllp_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    src = open(os.path.join(ROOT, "include", "mllp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mllp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.LIB_PATH


def test_header_declares_expected_surface():
    names = _declared()
    for must in ("mllp_graph_create_host", "mllp_graph_create_device", "mllp_spmm_csr_f32", "mllp_tconv_fwd",
                 "mllp_tconv_bwd", "mllp_gnn_forward", "mllp_gnn_backward", "mllp_gnn_loss_step", "mllp_adam_step",
                 "mllp_topm_metrics", "mllp_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    L = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(L, name), f"{name} declared in include/mllp_hip.h but not exported"
    assert sorted(_lib._PROTOTYPES) == _declared()      # the ctypes table covers the whole header


def test_version_and_argument_errors_without_gpu(lib_path):
    L = _lib.lib()
    assert L.mllp_abi_version() == _lib.ABI_VERSION
    # null arguments are rejected before any HIP call, with a message
    assert L.mllp_graph_dims(None, None) == -1
    assert b"null" in L.mllp_last_error()
    assert L.mllp_spmm_csr_f32(None, 0, None, None, None) == -1
    assert L.mllp_graph_destroy(None) == 0


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.MllpError):
        _lib.lib()
