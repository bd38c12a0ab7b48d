"""CPU: libclipk.so loads and exports every symbol include/clipk.h declares (no compute without a GPU),
and the product refuses to run without the HIP extension instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "clipk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clipk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from clip_dplm_amd import _ffi
    lib = _ffi.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/clipk.h but not exported"
        assert n in _ffi.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.clipk_arch() == b"gfx950"
    assert lib.clipk_version() == _ffi.ABI_VERSION
    assert b"unsupported" in lib.clipk_status_string(-2)


def test_no_cpu_fallback():
    import torch
    from clip_dplm_amd import _ffi, ops
    with pytest.raises(_ffi.ClipkError):
        ops.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "clip_dplm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
