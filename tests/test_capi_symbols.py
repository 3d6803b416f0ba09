"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/gsdf_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import gs_sdf_amd.capi as capi
    return capi


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(gsdf_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 14
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_binding_covers_every_declared_symbol(built):
    assert set(built.exported_symbols()) == _declared()
    l = built.lib()
    hdr = open(os.path.join(ROOT, "include", "gsdf_hip.h")).read()
    declared = int(re.search(r"#define\s+GSDF_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert l.gsdf_abi_version() == declared == built.ABI_VERSION


def test_stale_library_is_refused_at_load(built, monkeypatch):
    """A library whose gsdf_abi_version() differs from the binding's is refused before any call can reach the device."""
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "ABI_VERSION", built.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        built.lib()


def test_missing_library_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libgsdf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        built.lib()


def test_cpu_tensor_is_rejected(built):
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        built.ptr(torch.zeros(3), name="x")
