"""The C-ABI library builds, loads without a GPU and exports every symbol that
include/convdr_hip.h declares (no compute calls here)."""
import os
import re

from convdr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(convdr_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _declared()
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), "libconvdr_hip.so does not export %s" % name
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert L.convdr_version() >= 100


def test_argument_validation_needs_no_gpu():
    L = _lib.lib()
    assert L.convdr_ip_workspace_bytes(1000, 1_000_000, 768, 100, 4096) > 0
    rc = L.convdr_ip_prepare_block(None, 10, 70, None, None, None, None, None)     # d % 64 != 0 -> rejected before any launch
    assert rc != 0 and b"d % 64" in L.convdr_last_error()


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from convdr_amd.search import FlatIPIndex
    with pytest.raises(_lib.ConvdrError):
        FlatIPIndex(768)
