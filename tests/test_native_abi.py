"""CPU: the C-ABI library loads and exports every symbol include/lkamd.h declares."""
import ctypes

import pytest


def test_library_exports_every_declared_symbol():
    from lkpy_amd import _native

    lib = _native.load(build_if_missing=True)
    names = _native.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_padding_and_errors_without_gpu():
    from lkpy_amd import _native

    lib = _native.load(build_if_missing=True)
    assert b"gfx950" in lib.lk_version()
    assert [lib.lk_padded_dim(k) for k in (1, 16, 17, 25, 64, 65, 128, 200, 256)] == [
        16, 16, 32, 32, 64, 128, 128, 256, 256]  # fmt: skip
    assert lib.lk_padded_dim(0) == 0 and lib.lk_padded_dim(1025) == 0
    # above 256: multiples of 64 up to 1024 (csrc/als_big.hip)
    assert [lib.lk_padded_dim(k) for k in (257, 320, 321, 512, 1000, 1024)] == [
        320, 320, 384, 512, 1024, 1024]
    # argument validation happens before any device work
    h = ctypes.c_void_p(0)
    rc = lib.lk_als_plan_create(ctypes.byref(h), None, 0, 10, 64, 0)
    assert rc == _native.LK_E_INVALID and b"null" in lib.lk_last_error()
    assert lib.lk_gramian_workspace_bytes(64) > 0 and lib.lk_gramian_workspace_bytes(1300) == 0
    assert lib.lk_gramian_workspace_bytes(300) > 0


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU -- never route to the oracle."""
    import torch

    from lkpy_amd import _native

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.BackendUnavailable):
        _native.require_gpu()
    from lkpy_amd import _device

    with pytest.raises(_native.BackendUnavailable):
        _device.device("cuda")


def test_product_package_never_imports_the_oracle():
    import pathlib
    import re

    pkg = pathlib.Path(__file__).resolve().parent.parent / "lkpy_amd"
    for f in pkg.rglob("*.py"):
        text = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_integration_doc_names_every_entry_point():
    "INTEGRATION.md is the maintainer's map of the C ABI: nothing exported may be missing from it"
    from pathlib import Path

    from lkpy_amd import _native

    text = (Path(__file__).resolve().parent.parent / "INTEGRATION.md").read_text()
    missing = [s for s in _native.declared_symbols() if s not in text]
    assert not missing, missing
