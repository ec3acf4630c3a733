"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports exactly the symbols that
include/apla_hip.h declares; the Python binding table covers all of them; no compute call is made (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "apla_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(apla_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from apla_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from apla_amd import build
        build.build(verbose=False)
    handle = _lib.lib()
    decl = declared_symbols()
    assert len(decl) >= 18
    for name in decl:
        assert hasattr(handle, name), f"{name} declared in apla_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == decl
    assert handle.apla_version() >= 100


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of computing something else."""
    import torch
    from apla_amd import ops
    from apla_amd._lib import AplaHipError
    with pytest.raises(AplaHipError):
        ops.gemm_nt(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16))
    with pytest.raises(AplaHipError):
        ops.attn_fwd(torch.zeros(4, 192, dtype=torch.bfloat16), 1, 4, 1, 0.125)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "apla_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
