"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/alfi_hip.h declares,
and the product path fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "alfi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alfi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from alfi_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "libalfi_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "ctypes binding lacks %s" % name
    assert sorted(_lib.SIGNATURES) == names


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "alfi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alfi_amd import hip
    with pytest.raises(hip.AlfiHipError):
        hip.Context(0)
