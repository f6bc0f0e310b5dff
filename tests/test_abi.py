"""CPU: the C-ABI library builds, loads, and exports every symbol include/gank.h declares (and the
ctypes table binds exactly that set).  No compute calls -- there is no GPU here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "gank.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gank_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from gan_lib_tensorflow_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in gank.h but not exported"
    assert sorted(_lib.PROTOTYPES) == syms, set(_lib.PROTOTYPES) ^ set(syms)
    assert lib.gank_version() == 100


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan_lib_tensorflow_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(d, f)


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gan_lib_tensorflow_amd import kernels as K
    x = torch.zeros(8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        K.relu_fwd(x)
