"""The C-ABI library loads on a CPU-only box and exports every symbol include/mocap_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mocap_hip.h")).read()
    return sorted(set(re.findall(r"MOCAP_API\s+[\w\s\*]+?\b(mocap_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from mocapv2_amd import _abi
    assert declared_symbols() == sorted(_abi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from mocapv2_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _abi.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.mocap_abi_version() == _abi.ABI_VERSION


def test_no_gpu_is_reported_not_hidden():
    """Without a GPU every compute entry point must fail loudly (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mocapv2_amd import _abi
    lib = _abi.load()
    h = ctypes.c_void_p()
    rc = lib.mocap_ctx_create(0, 64, 64, 1, ctypes.byref(h))
    assert rc == -2 and lib.mocap_last_error()
    from mocapv2_amd.engine import MocapContext
    with pytest.raises(RuntimeError):
        MocapContext(64, 64)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mocapv2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
